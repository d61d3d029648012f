#!/bin/bash
# (GPU box) issue-level breakdown of every kernel from the SQ counters (one rocprofv3 --pmc pass, --kernel-trace only):
# wave cycles split into VALU-active / other-active / waiting, to show which kernels are VALU-bound.  usage: tools/gpu_sq.sh <tag> <workload>
tag=$1; wl=${2:-C4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq_pass
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d /tmp/sq_pass -o p -- python $R/bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-lfplus --no-ajtai --no-shard-model --chain 0 >/dev/null 2>/tmp/sq_err.txt
f=$(find /tmp/sq_pass -name '*counter_collection.csv' | head -1)
if [ -z "$f" ]; then tail -5 /tmp/sq_err.txt; exit 1; fi
python $R/tools/sq_summary.py "$f" $wl > $R/gpurun_out/${tag}_sq_$(echo $wl | tr A-Z a-z).json
