cd $GRAFT_REPO_ROOT
for e in A=1 LF_FOLD_NO_SV=1 A=1 LF_FOLD_SV_ROUNDS=2; do env $e timeout 300 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline --no-lfplus 2>gpurun_out/r04q.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 $e ms/step %.3f'%d['ms_per_step'], d['phases_ms_per_step'])"; tail -2 gpurun_out/r04q.err; done
(timeout 1200 python -m pytest tests/test_gpu_parity_scale.py -x -q -k "C3 or B14" 2>&1 | tail -5)
(timeout 1200 python -m pytest tests/test_gpu_bb.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3)
