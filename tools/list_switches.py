#!/usr/bin/env python3
"""Every environment switch the library reads (LF_* main path, LFPLUS_* LatticeFold+ slice, LFP_* its launch shapes), with its default and meaning, as one
Markdown table: python tools/list_switches.py > SWITCHES.md.  The main path parses its switches once per call into Tunables (lf_common.h); the table takes
default and comment from there, and lists every other getenv() with the file and line that reads it.  tests/test_abi_cpu.py checks that SWITCHES.md is current."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "latticefold_amd", "csrc")


# meanings of the switches whose reading site carries no comment of its own
EXTRA = {
    "LFPLUS_CACHE_GB": "cap of the per-device scratch cache destroyed LatticeFold+ contexts leave behind (default min(32 GB, 1/4 of HBM))",
    "LFPLUS_CM_DENSE": "Cm::prove keeps every instance table as ring elements (no compact exponent-byte / scalar tables)",
    "LFPLUS_POSEIDON_SCALAR": "Frog transcript on the scalar sparse permutation (FastPerm) instead of the AVX-512 IFMA lanes",
    "LFPLUS_RING_WEIGHTS": "M_q^T eq(r) as ring elements even when every M_q has constant coefficients",
    "LFPLUS_SC_NO_EARLY": "round 0 of the set check inside the loop over all sets instead of per set behind its tables",
    "LFPLUS_SC_TABLES": "materialise the beta^e / beta^2e tables of the set check (k_sc_tables) instead of running rounds 0-1 from the exponent digits",
    "LFPLUS_TIMELINE": "wall-clock marks of the protocol stages on stderr (the stream is drained at every mark)",
    "LF_DIST_NO_HANDSHAKE": "skip that self-check: one host thread issues every exchange",
    "LF_FOLD_FUSE_MIN": "entries from which fix_variables is fused into the folding round kernel (default 16384)",
    "LF_FOLD_LUT_MIN": "entries (m/4) from which rounds 3-4 run from the 81-entry look-up table (default 2^17; 2^14 in lf_fold_step)",
    "LF_FOLD_NO_LUT": "rounds 3-4 of the folding sumcheck on materialised m/4-entry tables",
    "LF_FOLD_NO_R4TAB": "round 4 without the product-free digit-code tables (mode 6)", "LF_FOLD_NO_R5TAB": "round 5 from materialised tables instead of the planes (mode 7)",
    "LF_FOLD_TAB_MIN": "pairs from which rounds 1-2 run as table look-ups (default 16384)", "LF_FOLD_TAB_R1": "round 1 as a table look-up round (disables the GEMM rounds)",
    "LF_FOLD_UNFUSED": "separate k_fix pass before every folding round", "LF_I8_COUPLE_W": "window (tiles) a commit workgroup may run ahead of its paired plane-group workgroup (default 4; 0 switches the coupling off)", "LF_I8_GUARDED": "commit kernel instantiation with guarded tile loads",
    "LF_I8G_PROF": "in-kernel cycle counters of the general commit kernel k_ajtai_i8g (tools/i8g_prof.py; lf_debug_i8_prof then returns its table)", "LF_I8_PROF": "in-kernel cycle counters of the commit kernel (tools/i8_prof.py)", "LF_LIN_U_EVAL": "u of the linearization from stand-alone evaluations instead of the last fix of the sumcheck tables", "LF_POSEIDON_AVX2": "(BabyBear) AVX2 lanes even when AVX-512 IFMA is present", "LF_POSEIDON_SCALAR": "scalar Poseidon permutation on the host", "LF_THETA_EVAL": "theta from stand-alone evaluations instead of the last fix of the folding tables",
    "LF_TIMELINE": "wall-clock marks of a fold step on stderr (lf_last_timeline carries them without it)", "LF_TRACE": "per-stage kernel error checks with a label",
}


def collect():
    rows, seen = {}, set()
    common = open(os.path.join(SRC, "lf_common.h")).read()
    # Tunables members: "<type> name = default;   // LF_NAME[=1]: comment" (the comment may continue on the following // lines)
    lines = common.splitlines()
    for i, ln in enumerate(lines):
        m = re.search(r"\b(\w+)\s*=\s*([^;,]+);\s*//\s*(LF_[A-Z0-9_]+)(?:=1)?:?\s*(.*)", ln)
        if not m:
            continue
        name, default, env, text = m.group(1), m.group(2).strip(), m.group(3), m.group(4).strip()
        j = i + 1
        while j < len(lines) and re.match(r"\s*//", lines[j]):
            text += " " + re.sub(r"^\s*//\s*", "", lines[j]).strip()
            j += 1
        rows[env] = (default, text, "lf_common.h")
    for fn in sorted(os.listdir(SRC)):
        if not fn.endswith((".cpp", ".hip", ".h", ".cc", ".cuh")):
            continue
        for n, ln in enumerate(open(os.path.join(SRC, fn)).read().splitlines(), 1):
            for env in re.findall(r'getenv\("((?:LF|LFPLUS|LFP)_[A-Z0-9_]+)"\)', ln):
                seen.add(env)
                if env not in rows:
                    c = re.search(r"//\s*(.*)$", ln)
                    rows[env] = ("unset", EXTRA.get(env) or (c.group(1).strip() if c else ""), fn)
    py = {}
    for fn in ("bench.py", os.path.join("latticefold_amd", "plus.py"), os.path.join("latticefold_amd", "api.py"), os.path.join("latticefold_amd", "dist.py")):
        for n, ln in enumerate(open(os.path.join(ROOT, fn)).read().splitlines(), 1):
            for env in re.findall(r'environ(?:\.get)?\(?\[?"((?:LF|LFPLUS)_[A-Z0-9_]+)"', ln):
                py.setdefault(env, fn)
    return rows, py


def render():
    rows, py = collect()
    out = ["# Environment switches", "",
           "Generated by `tools/list_switches.py` (do not edit).  Every switch is a test / diagnostic hook: the defaults are what `bench.py` and the parity tests run,",
           "a switch selects an alternative code path that produces the same words (the parity tests flip most of them).  `unset` = the switch is a flag that is off by default.", "",
           "| switch | default | read in | meaning |", "|---|---|---|---|"]
    for env in sorted(rows):
        default, text, where = rows[env]
        text = text or EXTRA.get(env, "")
        text = text.replace("|", "\\|")
        out.append(f"| `{env}` | `{default}` | `{where}` | {text} |")
    out += ["", "Host-language side (Python mirror / bench):", "", "| switch | read in |", "|---|---|"]
    for env in sorted(py):
        if env not in rows:
            out.append(f"| `{env}` | `{py[env]}` |")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    text = render()
    if "--write" in sys.argv:
        open(os.path.join(ROOT, "SWITCHES.md"), "w").write(text)
    else:
        sys.stdout.write(text)
