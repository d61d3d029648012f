#!/bin/bash
# (GPU box, ONE GPU) sharded-mode C4 steps with G = 1, 2, 4 processes sharing the GPU (host transport, gloo): the GPU serialises the
# ranks' kernels, so t(G) ~ G * replicated + sharded; the fit gives the per-step replicated / sharded split of the multi-GPU model.
R=${GRAFT_REPO_ROOT:-$(pwd)}
wl=${1:-C4}
mkdir -p $R/gpurun_out
python $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-lfplus 2>/dev/null > $R/gpurun_out/shard_g1.json
for G in 2 4; do
  LF_FORCE_DEVICE=0 LF_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$G --master-addr 127.0.0.1 --master-port $((29500+G)) \
     $R/bench.py --gpus $G --steps 4 --warmup 2 --workload $wl --parallelism shard 2>/dev/null | grep '^{' > $R/gpurun_out/shard_g$G.json
done
python - <<PY
import json
t = {}
for g in (1, 2, 4):
    try:
        d = json.load(open("$R/gpurun_out/shard_g%d.json" % g)); t[g] = d["ms_per_step"]; ex = d.get("exchanges")
        print("G=%d: %.2f ms/step" % (g, t[g]), ex if ex else "")
    except Exception as e:
        print("G=%d failed: %r" % (g, e))
if 1 in t and 2 in t:
    R_ = t[2] - t[1]; S = t[1] - R_
    print("fit from G=1,2: replicated %.2f ms, sharded %.2f ms -> multi-GPU model t(G) = %.2f + %.2f/G : " % (R_, S, R_, S) + ", ".join("G=%d %.1f ms" % (g, R_ + S / g) for g in (2, 4, 8)))
if 2 in t and 4 in t:
    R_ = (t[4] - t[2]) / 2; S = t[2] - 2 * R_
    print("fit from G=2,4: replicated %.2f ms, sharded %.2f ms -> " % (R_, S) + ", ".join("G=%d %.1f ms" % (g, R_ + S / g) for g in (2, 4, 8)))
PY
