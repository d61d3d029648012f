cd $GRAFT_REPO_ROOT
run() { echo "$1" >> gpurun_out/r04i_ab.txt; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']['k_ajtai_i8']
print('ms/step %.3f  commit avg %.4f ms  frac8d %.3f' % (d['ms_per_step'], k['avg_ms'], d['roofline']['frac']))
print(d['roofline']['phases'][-1]['timeline_mean_ms'])" >> gpurun_out/r04i_ab.txt
}
run "default (staged evals, couple 4/4, early prepare)" A=1
run "one stage" LF_EVALS_ONE_STAGE=1
run "default" A=1
(timeout 900 python -m pytest tests/test_gpu_parity_scale.py -x -q -k "C4 or C2" 2>&1 | tail -3) >> gpurun_out/r04i_ab.txt
cat gpurun_out/r04i_ab.txt
