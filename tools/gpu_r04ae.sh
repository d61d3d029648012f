cd $GRAFT_REPO_ROOT
for e in A=1 LF_NO_TAIL=1; do
echo "== $e" >> gpurun_out/r04ae.txt
env $e LF_TIMELINE=1 timeout 300 python bench.py --workload C3 --steps 3 --warmup 2 --no-cpu-baseline --no-lfplus 2>&1 >/dev/null | grep "bb timeline" | tail -36 | head -22 >> gpurun_out/r04ae.txt
done
cat gpurun_out/r04ae.txt
