cd $GRAFT_REPO_ROOT
run() { echo "$1" >> gpurun_out/r04m_ab.txt; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']['k_ajtai_i8']
print('ms/step %.3f  commit avg %.4f ms' % (d['ms_per_step'], k['avg_ms']))
t=d['roofline']['phases'][-1]['timeline_mean_ms']
print({k:v for k,v in t.items() if v<9 and v>1.5})" >> gpurun_out/r04m_ab.txt
}
run "ZR=2 staged" LF_ZR_POS=2
run "ZR=2 one stage" LF_ZR_POS=2 LF_EVALS_ONE_STAGE=1
run "ZR=0 one stage" LF_EVALS_ONE_STAGE=1
run "ZR=0 staged" A=1
run "ZR=2 staged" LF_ZR_POS=2
run "ZR=2 one stage" LF_ZR_POS=2 LF_EVALS_ONE_STAGE=1
run "ZR=0 one stage" LF_EVALS_ONE_STAGE=1
run "ZR=0 staged" A=1
run "ZR=3 staged" LF_ZR_POS=3
(timeout 900 python -m pytest tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_prover.py tests/test_gpu_lfplus_scale.py -x -q 2>&1 | tail -3) >> gpurun_out/r04m_ab.txt
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 2 2>&1 | grep -v "round " | tail -24 >> gpurun_out/r04m_ab.txt
cat gpurun_out/r04m_ab.txt
