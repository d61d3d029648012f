cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 > gpurun_out/fulltests.txt
cat gpurun_out/fulltests.txt
