cd $GRAFT_REPO_ROOT
LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 2>&1 | grep -v "round " | tail -26 > gpurun_out/r04o.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_lfp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lfp -o p -- python $GRAFT_REPO_ROOT/tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 2 >/dev/null 2>&1
f=$(find /tmp/prof_lfp -name '*kernel_stats.csv' | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04o_lfplus_ks_p20.csv
cd $GRAFT_REPO_ROOT; head -22 gpurun_out/r04o_lfplus_ks_p20.csv | cut -c1-160 >> gpurun_out/r04o.txt
(timeout 600 python -m pytest tests/test_gpu_lfplus_protocol.py tests/test_gpu_lfplus_scale.py -x -q 2>&1 | tail -3) >> gpurun_out/r04o.txt
cat gpurun_out/r04o.txt
