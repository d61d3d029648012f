cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_lfplus_prover.py tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_scale.py -x -q 2>&1 | tail -3) > gpurun_out/lfp4.txt
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 2 --resident 2>&1 | grep -v " round " | tail -28 >> gpurun_out/lfp4.txt
timeout 600 python tools/bench_lfplus.py --nvars 17 20 --k 4 --fresh 3 --rounds 2 2>&1 | tail -2 >> gpurun_out/lfp4.txt
cat gpurun_out/lfp4.txt
