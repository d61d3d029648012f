cd $GRAFT_REPO_ROOT
LF_TIMELINE=1 timeout 300 python bench.py --workload C3 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 >/dev/null | grep "bb timeline" | tail -42 > gpurun_out/tl3.txt
cat gpurun_out/tl3.txt
