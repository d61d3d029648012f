cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_lfplus.py tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_prover.py tests/test_gpu_lfplus_scale.py -x -q 2>&1 | tail -15) > gpurun_out/r04c_lfplus_tests.log
(timeout 2400 python -m pytest tests/test_dist_shard_lfplus.py -x -q 2>&1 | tail -30) > gpurun_out/r04c_lfplus_shard.log
cat gpurun_out/r04c_lfplus_tests.log gpurun_out/r04c_lfplus_shard.log
