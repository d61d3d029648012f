cd $GRAFT_REPO_ROOT
rm -f gpurun_out/c3r5.txt
for e in A=1 LF_FOLD_NO_R5TAB=1 A=1 LF_FOLD_NO_R5TAB=1; do env $e timeout 300 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 $e ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/c3r5.txt; done
cat gpurun_out/c3r5.txt
