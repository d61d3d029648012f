cd $GRAFT_REPO_ROOT
rm -f gpurun_out/lfp1.txt
for nb in 256 512 1024 2048 4096; do
echo "== LFPLUS_ROUND_BLOCKS=$nb" >> gpurun_out/lfp1.txt
LFPLUS_ROUND_BLOCKS=$nb LFPLUS_TIMELINE=1 timeout 600 python tools/bench_lfplus.py --nvars 20 --k 4 --fresh 3 --rounds 1 2>&1 | grep -E "cm: sumchecker|linearize: tables|gpu_prove_ms|cm round  [0-3]:" | tail -14 >> gpurun_out/lfp1.txt
done
cat gpurun_out/lfp1.txt
