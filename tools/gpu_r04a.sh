cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r04a_gputest.log
timeout 300 python bench.py > gpurun_out/r04a_bench_c4.json 2> gpurun_out/r04a_bench_c4.err
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 17 20 --k 4 --fresh 3 --rounds 2 > gpurun_out/r04a_lfplus_e2e.txt 2> gpurun_out/r04a_lfplus_e2e.err
tail -3 gpurun_out/r04a_gputest.log; cat gpurun_out/r04a_bench_c4.json | cut -c1-600; cat gpurun_out/r04a_lfplus_e2e.txt; tail -5 gpurun_out/r04a_lfplus_e2e.err
