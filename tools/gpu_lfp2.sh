cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_scale.py tests/test_gpu_lfplus_prover.py -x -q 2>&1 | tail -3) > gpurun_out/lfp2.txt
for nb in 4096 2048; do
echo "== LFPLUS_ROUND_BLOCKS=$nb" >> gpurun_out/lfp2.txt
LFPLUS_ROUND_BLOCKS=$nb LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 2>&1 | grep -v " round " | tail -27 >> gpurun_out/lfp2.txt
done
cat gpurun_out/lfp2.txt
