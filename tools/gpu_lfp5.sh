cd $GRAFT_REPO_ROOT
rm -f gpurun_out/lfp5.txt
for nb in 512 1024 2048 4096; do
echo "== LFPLUS_SC_BLOCKS=$nb" >> gpurun_out/lfp5.txt
LFPLUS_SC_BLOCKS=$nb LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 --resident 2>&1 | grep -E "set check: sumcheck rounds|gpu_prove_ms" | tail -2 >> gpurun_out/lfp5.txt
done
cat gpurun_out/lfp5.txt
