cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ajtai_i8.py -x -q 2>&1 | tail -4) > gpurun_out/r04h_tests.log
(timeout 900 python -m pytest tests/test_gpu_parity_scale.py -x -q -k "C4 or C2 or T18" 2>&1 | tail -4) >> gpurun_out/r04h_tests.log
run() { echo "$1" >> gpurun_out/r04h_ab.txt; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']['k_ajtai_i8']
print('ms/step %.3f  commit avg %.4f ms  frac8d %.3f' % (d['ms_per_step'], k['avg_ms'], d['roofline']['frac']))" >> gpurun_out/r04h_ab.txt
}
run "default (staged evals, couple 4/4)" A=1
run "one stage" LF_EVALS_ONE_STAGE=1
run "default" A=1
run "one stage" LF_EVALS_ONE_STAGE=1
run "default, no coupling" LF_I8_COUPLE_W=0
LF_TIMELINE=1 python bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 | grep "^\[timeline\]" | tail -34 > gpurun_out/r04h_timeline_c4.txt
for wl in C2; do for e in A=1 LF_EVALS_ONE_STAGE=1; do env $e python bench.py --workload $wl --steps 30 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl $e ms/step %.3f'%d['ms_per_step'])" >> gpurun_out/r04h_ab.txt; done; done
cat gpurun_out/r04h_tests.log gpurun_out/r04h_ab.txt gpurun_out/r04h_timeline_c4.txt
