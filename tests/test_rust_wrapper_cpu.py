"""bindings/latticefold-hip (the safe Rust layer over the generated -sys crate) cannot be compiled in the build image (no Rust toolchain).  What can be checked
here: it is there as a crate (Cargo.toml, build.rs of the -sys crate linking lfhip), every `sys::` function it calls is a declared and exported symbol, it
implements the reference's three prover traits and the two transcript traits, and -- against the reference sources where they are present (names only, nothing is
copied) -- each `prove` has the reference's parameter NAMES and count (nifs/linearization.rs:26-52, nifs/decomposition/structs.rs:48-64,
nifs/folding/structs.rs:43-70, nifs.rs:48-103), and the methods of AjtaiCommitmentScheme / Witness the path uses exist on HipAjtai / HipWitness."""
import os
import re

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = open(os.path.join(ROOT, "bindings", "latticefold-hip", "src", "lib.rs")).read()
REF = "/root/reference/crates/latticefold/src"


def _params(sig):
    """parameter names of a Rust fn signature (text between the outer parentheses), generics / nested parentheses skipped"""
    depth, cur, out = 0, "", []
    for ch in sig:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [p.split(":")[0].strip().lstrip("&").replace("mut ", "").strip() for p in out if ":" in p]


def _fn_sig(text, owner_pat, name="prove"):
    m = re.search(owner_pat, text, re.S)
    assert m, owner_pat
    body = text[m.end():]
    f = re.search(r"fn\s+" + name + r"\s*(?:<[^>]*>)?\s*\(", body)
    assert f, (owner_pat, name)
    depth, i = 1, f.end()
    while depth:
        depth += {"(": 1, ")": -1}.get(body[i], 0)
        i += 1
    return _params(body[f.end():i - 1])


def test_crate_layout_and_sys_calls_are_declared_and_exported():
    from latticefold_amd import api
    for rel in ("bindings/latticefold-hip/Cargo.toml", "bindings/latticefold-hip-sys/Cargo.toml", "bindings/latticefold-hip-sys/build.rs"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    assert "rustc-link-lib=dylib=lfhip" in open(os.path.join(ROOT, "bindings", "latticefold-hip-sys", "build.rs")).read()
    sysrs = open(os.path.join(ROOT, "bindings", "latticefold-hip-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (lf_[a-z0-9_]+)\(", sysrs))
    used = set(re.findall(r"sys::(lf_[a-z0-9_]+)", SRC)) - {"lf_ctx", "lf_witness", "lf_witness_job", "lf_transcript", "lf_params"}     # (the opaque / plain types)
    assert used and used <= declared, sorted(used - declared)
    lib = api._lib()
    assert all(hasattr(lib, s) for s in used)


def test_wrapper_implements_the_reference_traits():
    for pat in (r"impl<[^{]*>\s*LinearizationProver<NTT, T> for HipLinearizationProver<NTT, T>",
                r"impl<[^{]*>\s*DecompositionProver<NTT, T> for HipDecompositionProver<NTT, T>",
                r"impl<[^{]*>\s*FoldingProver<NTT, T> for HipFoldingProver<NTT, T>",
                r"impl<[^{]*>\s*Transcript<NTT> for HipTranscript<NTT, CS>",
                r"impl<[^{]*>\s*TranscriptWithShortChallenges<NTT> for HipTranscript<NTT, CS>"):
        assert re.search(pat, SRC), pat
    for name in ("new", "absorb", "get_challenge", "squeeze_bytes", "get_short_challenge"):
        assert re.search(r"fn " + name + r"\(", SRC), name
    for name in ("new", "rand", "commit", "commit_ntt", "kappa", "width"):          # AjtaiCommitmentScheme
        assert re.search(r"impl<NTT: SuitableRing> HipAjtai<NTT> \{.*?pub fn " + name + r"\(", SRC, re.S), name
    for name in ("from_w_ccs", "from_f", "from_f_coeff", "commit", "within_bound"):   # Witness
        assert re.search(r"pub fn " + name + r"\(", SRC), name


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources not present on this machine")
def test_prove_signatures_carry_the_reference_parameter_names():
    cases = [("nifs/linearization.rs", r"pub trait LinearizationProver<", r"LinearizationProver<NTT, T> for HipLinearizationProver<NTT, T>\s*\{"),
             ("nifs/decomposition/structs.rs", r"pub trait DecompositionProver<", r"DecompositionProver<NTT, T> for HipDecompositionProver<NTT, T>\s*\{"),
             ("nifs/folding/structs.rs", r"pub trait FoldingProver<", r"FoldingProver<NTT, T> for HipFoldingProver<NTT, T>\s*\{"),
             ("nifs.rs", r"impl<NTT: SuitableRing, P: DecompositionParams, T: TranscriptWithShortChallenges<NTT>>\s*NIFSProver<NTT, P, T>\s*\{",
              r"impl<NTT: SuitableRing, P: DecompositionParams, T: TranscriptWithShortChallenges<NTT>> HipNIFSProver<NTT, P, T>\s*\{")]
    for rel, ref_pat, my_pat in cases:
        want = _fn_sig(open(os.path.join(REF, rel)).read(), ref_pat)
        got = [p.lstrip("_") for p in _fn_sig(SRC, my_pat)]
        assert got == want, (rel, got, want)
    # the transcript traits: every REQUIRED method of the reference's traits is implemented
    tr = open(os.path.join(REF, "transcript.rs")).read()
    required = re.findall(r"\n    fn (\w+)\([^)]*\)[^{;]*;", tr)
    assert set(required) >= {"new", "absorb", "get_challenge", "squeeze_bytes", "get_short_challenge"}
    for name in required:
        assert re.search(r"fn " + name + r"\(", SRC), name
