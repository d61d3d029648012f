cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04ab.txt
for e in A=1 LF_BB_EVALS_FIRST=1 LF_I8_WGS=208 LF_I8_WGS=192 LF_I8_WGS=176 LF_I8_WGS=160 "LF_BB_EVALS_FIRST=1 LF_I8_WGS=192" A=1 LF_BB_EVALS_FIRST=1; do env $e timeout 300 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 $e ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/r04ab.txt; done
LF_BB_EVALS_FIRST=1 LF_TIMELINE=1 timeout 300 python bench.py --workload C3 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 >/dev/null | tail -34 | head -20 >> gpurun_out/r04ab.txt
cat gpurun_out/r04ab.txt
