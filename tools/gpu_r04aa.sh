cd $GRAFT_REPO_ROOT
LF_TIMELINE=1 timeout 300 python bench.py --workload C3 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus 2>gpurun_out/r04aa_timeline_c3.txt >/dev/null
tail -45 gpurun_out/r04aa_timeline_c3.txt
