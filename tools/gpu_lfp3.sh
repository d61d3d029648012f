cd $GRAFT_REPO_ROOT
rm -f gpurun_out/lfp3.txt
for ch in 1024 2048 4096; do
echo "== LFPLUS_EVAL_CHUNKS=$ch" >> gpurun_out/lfp3.txt
LFPLUS_EVAL_CHUNKS=$ch LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 2>&1 | grep -E "evaluation passes|range check: eval|gpu_prove_ms" | tail -3 >> gpurun_out/lfp3.txt
done
cat gpurun_out/lfp3.txt
