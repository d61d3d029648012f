cd $GRAFT_REPO_ROOT
make -C oracle -s
(timeout 1800 python tests/tools/make_lfplus_digests.py P20 > gpurun_out/r04b_p20_oracle.log 2>&1; cp tests/golden/lfplus_digests.json gpurun_out/lfplus_digests.json) &
(timeout 900 python -m pytest tests/test_gpu_lfplus.py tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_prover.py -x -q 2>&1 | tail -5) > gpurun_out/r04b_lfplus_tests.log
wait
free -g | head -2 >> gpurun_out/r04b_p20_oracle.log
(timeout 900 python -m pytest tests/test_gpu_lfplus_scale.py -x -q 2>&1 | tail -8) > gpurun_out/r04b_lfplus_scale.log
cat gpurun_out/r04b_p20_oracle.log gpurun_out/r04b_lfplus_tests.log gpurun_out/r04b_lfplus_scale.log
