cd $GRAFT_REPO_ROOT
run() { echo "$1" >> gpurun_out/r04l_ab.txt; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']['k_ajtai_i8']
print('ms/step %.3f  commit avg %.4f ms' % (d['ms_per_step'], k['avg_ms']))
t=d['roofline']['phases'][-1]['timeline_mean_ms']
print({k:v for k,v in t.items() if v<9 and v>1.5})" >> gpurun_out/r04l_ab.txt
}
run "default (early y L+R)" A=1
run "no early y" LF_NO_EARLY_Y=1
run "ZR=3" LF_ZR_POS=3
run "ZR=2" LF_ZR_POS=2
run "ZR=4" LF_ZR_POS=4
run "default" A=1
run "ZR=3 one stage" LF_ZR_POS=3 LF_EVALS_ONE_STAGE=1
run "WGS=240" LF_I8_WGS=240
run "WGS=256 ZR=3" LF_I8_WGS=256 LF_ZR_POS=3
(timeout 900 python -m pytest tests/test_gpu_parity_scale.py -x -q -k "C4 or C2" 2>&1 | tail -3) >> gpurun_out/r04l_ab.txt
cat gpurun_out/r04l_ab.txt
