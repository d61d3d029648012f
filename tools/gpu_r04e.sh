cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_ajtai_i8.py -x -q 2>&1 | tail -4) > gpurun_out/r04e_tests.log
(timeout 900 python -m pytest tests/test_gpu_parity_scale.py -x -q -k "C4 or C2" 2>&1 | tail -4) >> gpurun_out/r04e_tests.log
for w in 0 3 6 12 3 0; do
  echo "COUPLE_W=$w" >> gpurun_out/r04e_ab.txt
  LF_I8_COUPLE_W=$w timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']['k_ajtai_i8']
print('ms/step %.3f  commit avg %.4f ms  launches/step %.1f  frac8d %.3f' % (d['ms_per_step'], k['avg_ms'], k['launches_per_step'], d['roofline']['frac']))" >> gpurun_out/r04e_ab.txt
done
# PMC of the coupled kernel (FETCH only: the write side is the partial tiles)
cd /tmp && export TMPDIR=/tmp
for w in 3 0; do
rm -rf /tmp/pmc_f
LF_I8_COUPLE_W=$w timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
f=$(find /tmp/pmc_f -name '*counter_collection.csv' | head -1)
python - "$f" $w <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/r04e_ab.txt
import csv,sys
v=[float(r["Counter_Value"])*2*1024/1e9 for r in csv.DictReader(open(sys.argv[1])) if "k_ajtai_i8s" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("COUPLE_W=%s PMC fetch GB per launch of k_ajtai_i8s:"%sys.argv[2], [round(x,2) for x in v])
PY
done
cd $GRAFT_REPO_ROOT; cat gpurun_out/r04e_tests.log gpurun_out/r04e_ab.txt
