cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_lfplus.py tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_prover.py tests/test_gpu_lfplus_scale.py tests/test_gpu_scale.py -x -q 2>&1 | tail -4) > gpurun_out/r04n.txt
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 2 2>&1 | grep -v "round " | tail -26 >> gpurun_out/r04n.txt
timeout 600 python tools/bench_lfplus.py --nvars 17 20 --k 4 --fresh 3 --rounds 3 2>/dev/null >> gpurun_out/r04n.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('C4 ms/step %.3f frac8d %.3f'%(d['ms_per_step'], d['roofline']['frac']))" >> gpurun_out/r04n.txt; done
cat gpurun_out/r04n.txt
