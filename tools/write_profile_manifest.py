#!/usr/bin/env python3
"""Writes <dir>/latest.json: the manifest of ONE profiling round (tag) that bench.py reads for `roofline.traffic`, `roofline.phases[].kernel_ms / hbm_bytes /
valu_floor_ms` and `whole_step_traffic` -- so those columns can never point at a stale tag by hand.

    python tools/write_profile_manifest.py <tag> [dir]        dir defaults to profiles/ (tools/gpu_round.sh calls it on gpurun_out/ on the GPU box;
                                                              copy the <tag>_* files AND latest.json into profiles/ together)

A kind (pmc / stats / sq) that the round did not produce for a workload is simply absent: bench.py then reports null for it instead of an older file."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    d = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    files = {}
    for wl in ("c4", "c3", "c2"):
        for kind, name in (("pmc", f"{tag}_pmc_{wl}.json"), ("stats", f"{tag}_{wl}_kernel_stats.csv"), ("sq", f"{tag}_sq_{wl}.json"),
                           ("bench", f"{tag}_bench_{wl}.json"), ("timeline", f"{tag}_timeline_{wl}.txt"), ("step_trace", f"{tag}_step_trace_{wl}.txt")):
            if os.path.exists(os.path.join(d, name)) and os.path.getsize(os.path.join(d, name)) > 0:
                files.setdefault(kind, {})[wl] = name
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        head = None      # (the GPU box has no .git: the commit is filled in when the files are copied into profiles/)
    out = {"tag": tag, "files": files, "source_commit": head,
           "note": "written by tools/write_profile_manifest.py; every file named here was produced by tools/gpu_round.sh in ONE gpurun call on one code state"}
    json.dump(out, open(os.path.join(d, "latest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
