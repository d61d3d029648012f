cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_bb.py -x -q 2>&1 | tail -3) > gpurun_out/q9.txt
(timeout 900 python -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_scale.py -x -q -k "C3 or B14 or B10" 2>&1 | tail -2) >> gpurun_out/q9.txt
for e in A=1 LF_FOLD_R5_ONE_LANE=1 A=1 LF_FOLD_R5_ONE_LANE=1; do env $e timeout 300 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 $e ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/q9.txt; done
LF_TIMELINE=1 timeout 300 python bench.py --workload C3 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 >/dev/null | grep "bb timeline" | tail -14 | head -5 >> gpurun_out/q9.txt
cat gpurun_out/q9.txt
