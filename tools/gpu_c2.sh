cd $GRAFT_REPO_ROOT
LF_TIMELINE=1 timeout 300 python bench.py --workload C2 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 >/dev/null | grep "^\[timeline\]" | tail -36 > gpurun_out/c2_tl.txt
cat gpurun_out/c2_tl.txt
