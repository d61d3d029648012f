#!/bin/bash
# (GPU box) HBM traffic of every kernel from rocprofv3 PMC counters, collected as MI355X_MICROARCH.md prescribes: one counter
# per pass, --kernel-trace only (no sys/hip/hsa trace domains).  usage: tools/gpu_pmc.sh <tag> <workload>
tag=$1; wl=${2:-C4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o p -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-lfplus --no-ajtai --no-shard-model --chain 0 >/dev/null 2>&1
  f=$(find /tmp/pmc_$ctr -name '*counter_collection.csv' | head -1)
  cp "$f" $R/gpurun_out/pmc_${tag}_${ctr}.csv
done
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${tag}_FETCH_SIZE.csv $R/gpurun_out/pmc_${tag}_WRITE_SIZE.csv $wl > $R/gpurun_out/${tag}_pmc_$(echo $wl | tr A-Z a-z).json
rm -f $R/gpurun_out/pmc_${tag}_FETCH_SIZE.csv $R/gpurun_out/pmc_${tag}_WRITE_SIZE.csv
