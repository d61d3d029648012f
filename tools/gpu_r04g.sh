cd $GRAFT_REPO_ROOT
run() { # label env...
  echo "$1" >> gpurun_out/r04g_ab.txt; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-lfplus 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['roofline']['kernels']['k_ajtai_i8']
print('ms/step %.3f  commit avg %.4f ms  frac8d %.3f' % (d['ms_per_step'], k['avg_ms'], d['roofline']['frac']))" >> gpurun_out/r04g_ab.txt
}
run "W=0" LF_I8_COUPLE_W=0
run "W=3 E=2" LF_I8_COUPLE_W=3 LF_I8_COUPLE_E=2
run "W=4 E=4" LF_I8_COUPLE_W=4 LF_I8_COUPLE_E=4
run "W=6 E=4" LF_I8_COUPLE_W=6 LF_I8_COUPLE_E=4
run "W=8 E=8" LF_I8_COUPLE_W=8 LF_I8_COUPLE_E=8
run "W=0" LF_I8_COUPLE_W=0
run "W=4 E=2" LF_I8_COUPLE_W=4 LF_I8_COUPLE_E=2
cd /tmp && export TMPDIR=/tmp
for cfg in "4 4" "8 8" "3 2"; do
set -- $cfg
rm -rf /tmp/pmc_f
LF_I8_COUPLE_W=$1 LF_I8_COUPLE_E=$2 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-lfplus >/dev/null 2>&1
f=$(find /tmp/pmc_f -name '*counter_collection.csv' | head -1)
python - "$f" "$cfg" <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/r04g_ab.txt
import csv,sys
v=[float(r["Counter_Value"])*2*1024/1e9 for r in csv.DictReader(open(sys.argv[1])) if "k_ajtai_i8s" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("W E = %s PMC fetch GB per launch of k_ajtai_i8s:"%sys.argv[2], [round(x,2) for x in v])
PY
done
cd $GRAFT_REPO_ROOT; cat gpurun_out/r04g_ab.txt
