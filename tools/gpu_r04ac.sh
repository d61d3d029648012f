cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r04ac.txt
for e in A=1 A=1; do env $e timeout 300 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 $e ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/r04ac.txt; done
LF_TIMELINE=1 timeout 300 python bench.py --workload C3 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 >/dev/null | tail -12 | head -8 >> gpurun_out/r04ac.txt
cat gpurun_out/r04ac.txt
