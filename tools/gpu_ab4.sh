cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_bb.py -x -q -k "fold_step" 2>&1 | tail -2) > gpurun_out/ab4.txt
for e in A=1 LF_I8_BITS=1 LF_I8_COUPLE_W=0 LF_ZR_POS=3 A=1 LF_I8_BITS=1 LF_I8_COUPLE_W=0 LF_ZR_POS=3; do env $e timeout 300 python bench.py --workload C4 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 $e ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/ab4.txt; done
for e in A=1 A=1; do env $e timeout 300 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 $e ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/ab4.txt; done
cat gpurun_out/ab4.txt
