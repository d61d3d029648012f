cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py -x -q 2>&1 | tail -3) > gpurun_out/q10.txt
for wl in C4 C4 C2 C4; do timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl ms/step %.3f'%d['ms_per_step'], {k:round(v,2) for k,v in d['phases_ms_per_step'].items()})" >> gpurun_out/q10.txt; done
cat gpurun_out/q10.txt
