#!/usr/bin/env python3
"""Extract known-answer-test VECTORS (numbers only) from the reference's own unit tests.

Run in the build container (where /root/reference exists):
    python tests/tools/extract_kats.py
Writes tests/golden/kats.json.  Only numeric inputs / expected outputs are taken; no
reference source text is stored.  Each entry names the reference test it came from.
"""
import json, os, re, sys

REF = os.environ.get("LF_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(__file__), "..", "golden", "kats.json")


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def between(src, start, end=None):
    i = src.index(start)
    j = src.index(end, i) if end else len(src)
    return src[i:j]


def main():
    kats = {}

    # --- crates/cyclotomic-rings/src/rotation.rs:174-775  test_rot_lin_combination
    src = between(read("crates/cyclotomic-rings/src/rotation.rs"), "fn test_rot_lin_combination")
    nums = [int(x) for x in re.findall(r"Fq::from\((\d+)u64\)", src)]
    assert len(nums) == 72 + 216 + 72, len(nums)
    kats["rot_lin_combination"] = {
        "source": "crates/cyclotomic-rings/src/rotation.rs:174-775",
        "rho_coeff": [nums[24 * i:24 * (i + 1)] for i in range(3)],
        # theta[i][j] = NTT-form ring element, 8 slots x 3 coords, slot-major
        "theta": [[nums[72 + 72 * i + 24 * j: 72 + 72 * i + 24 * (j + 1)] for j in range(3)] for i in range(3)],
        "expected": [nums[288 + 24 * i: 288 + 24 * (i + 1)] for i in range(3)],
    }

    # --- crates/cyclotomic-rings/src/rings/goldilocks.rs:78-115
    src = between(read("crates/cyclotomic-rings/src/rings/goldilocks.rs"), "fn test_small_challenge_from_random_bytes")
    bs = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", between(src, "short_challenge_from_random_bytes(&[", "])"))]
    co = [int(x) for x in re.findall(r"BigInt\(\[(\d+)\]\)", src)]
    assert len(bs) == 18 and len(co) == 24
    kats["goldilocks_small_challenge_from_bytes"] = {
        "source": "crates/cyclotomic-rings/src/rings/goldilocks.rs:78-115",
        "bytes": bs, "coeffs": co,
    }

    # --- crates/cyclotomic-rings/src/rings/babybear.rs:77-114
    src = between(read("crates/cyclotomic-rings/src/rings/babybear.rs"), "fn test_small_challenge_from_random_bytes")
    bs = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{2})", between(src, "short_challenge_from_random_bytes(&[", "])"))]
    co = [int(x) for x in re.findall(r"BigInt\(\[(\d+)\]\)", src)]
    assert len(bs) == 18 and len(co) == 24
    kats["babybear_small_challenge_from_bytes"] = {
        "source": "crates/cyclotomic-rings/src/rings/babybear.rs:77-114",
        "bytes": bs, "coeffs": co,
    }

    # --- crates/latticefold/src/transcript/poseidon.rs:85-142
    src = read("crates/latticefold/src/transcript/poseidon.rs")
    big = between(src, "fn test_get_big_challenge", "fn test_get_small_challenge")
    small = between(src, "fn test_get_small_challenge")
    kats["poseidon_big_challenge"] = {
        "source": "crates/latticefold/src/transcript/poseidon.rs:85-103",
        "absorb": [0xFF],
        "expected_fq3": [int(x) for x in re.findall(r"BigInt\(\[(\d+)\]\)", big)],
    }
    kats["poseidon_small_challenge"] = {
        "source": "crates/latticefold/src/transcript/poseidon.rs:105-142",
        "absorb": [0xFF],
        "expected_coeffs": [int(x) for x in re.findall(r"BigInt\(\[(\d+)\]\)", small)],
    }
    assert len(kats["poseidon_big_challenge"]["expected_fq3"]) == 3
    assert len(kats["poseidon_small_challenge"]["expected_coeffs"]) == 24

    # --- Poseidon round constants: spot values + count, to pin the Grain-LFSR regeneration
    # crates/cyclotomic-rings/src/rings/poseidon/goldilocks.rs:7-1425
    src = read("crates/cyclotomic-rings/src/rings/poseidon/goldilocks.rs")
    vals = [int(x, 16) for x in re.findall(r"Fq::from\(0x([0-9a-f]+)_i128\)", src)]
    assert len(vals) == 30 * 24 + 24 * 24
    p = 2**64 - 2**32 + 1
    kats["poseidon_goldilocks_params"] = {
        "source": "crates/cyclotomic-rings/src/rings/poseidon/goldilocks.rs:7-1425",
        "full_rounds": 8, "partial_rounds": 22, "alpha": 7, "rate": 20, "capacity": 4,
        "n_ark": 720, "n_mds": 576,
        "ark_first": vals[:4], "ark_last": vals[716:720],
        "mds_first": vals[720:724], "mds_last": vals[-4:],
        # order-sensitive checksums of the full tables: sum_i (i+1)*v_i mod p
        "ark_checksum": sum((i + 1) * v for i, v in enumerate(vals[:720])) % p,
        "mds_checksum": sum((i + 1) * v for i, v in enumerate(vals[720:])) % p,
    }

    # BabyBear table (rings/poseidon/babybear.rs:7-1425): the same 64-bit literals, embedded with Fq::from(i128)
    # (i.e. reduced mod p_BB = 15*2^27+1) -- pinned by checksums of the reduced values
    srcb = read("crates/cyclotomic-rings/src/rings/poseidon/babybear.rs")
    valsb = [int(x, 16) for x in re.findall(r"Fq::from\(0x([0-9a-f]+)_i128\)", srcb)]
    assert len(valsb) == 30 * 24 + 24 * 24
    pb = 15 * 2**27 + 1
    kats["poseidon_babybear_params"] = {
        "source": "crates/cyclotomic-rings/src/rings/poseidon/babybear.rs:7-1425",
        "full_rounds": 8, "partial_rounds": 22, "alpha": 7, "rate": 20, "capacity": 4,
        "same_literals_as_goldilocks": valsb == vals,
        "ark_first": [v % pb for v in valsb[:4]], "mds_last": [v % pb for v in valsb[-4:]],
        "ark_checksum": sum((i + 1) * (v % pb) for i, v in enumerate(valsb[:720])) % pb,
        "mds_checksum": sum((i + 1) * (v % pb) for i, v in enumerate(valsb[720:])) % pb,
    }

    # --- crates/latticefold/src/arith.rs:455-502  test_get_fhat (inputs are written as code there;
    # restated here as data: coefficient vectors and the expected slot values)
    kats["get_fhat"] = {
        "source": "crates/latticefold/src/arith.rs:455-502",
        "f_coeff": [[1, 2, 3] + [0] * 21, [4, 5, 6] + [1] * 21],
        # fhat[j][i] = 8 base-field slot values (each embedded as (c,0,0))
        "fhat_slots": [
            [[1, 2, 3, 0, 0, 0, 0, 0], [4, 5, 6, 1, 1, 1, 1, 1]],
            [[0] * 8, [1] * 8],
            [[0] * 8, [1] * 8],
        ],
    }

    # --- crates/latticefold/src/commitment/commitment_scheme.rs:122-159 test_commit_ntt
    kats["commit_ntt"] = {
        "source": "crates/latticefold/src/commitment/commitment_scheme.rs:122-159",
        "kappa": 9, "n": 1 << 15,
        "matrix_entry": "diag(i*n + j)", "witness_entry": "diag(2)",
        "expected_formula": "n*(2*i*n + (n-1))",
    }

    # --- decomposition parameter sets used by the reference tests/benches
    kats["decomposition_params"] = {
        "source": "crates/latticefold/src/decomposition_parameters.rs:89-105; benches/config.toml:156,163",
        "GoldilocksDP": {"B": 1 << 15, "L": 5, "b": 2, "K": 15},
        "BabyBearDP": {"B": 1 << 8, "L": 4, "b": 2, "K": 8},
        "C3": {"B": 1 << 16, "L": 2, "b": 2, "K": 16, "kappa": 16, "wit_len": 1 << 17},
        "C2": {"B": 1 << 16, "L": 4, "b": 2, "K": 16, "kappa": 25, "wit_len": 16384},
        "C4": {"B": 1 << 16, "L": 4, "b": 2, "K": 16, "kappa": 26, "wit_len": 1 << 18},
    }

    # --- crates/latticefold-plus/src/utils.rs:118-131 test_tensor_product / test_tensor (plain integers; negative products are written as
    #     products of literals in the source and evaluated here)
    src = read("crates/latticefold-plus/src/utils.rs")
    tp = between(src, "fn test_tensor_product", "fn test_tensor()")
    vecs = [[int(x) for x in re.findall(r"-?\d+", v)] for v in re.findall(r"vec!\[([^\]]*)\]", tp)]
    assert len(vecs) == 3
    tt = between(src, "fn test_tensor()")
    vv = re.findall(r"vec!\[([^\]]*)\]", tt)
    r_in = [int(x) for x in re.findall(r"-?\d+", vv[0])]
    exp_terms = [t.strip() for t in vv[1].split(",")]
    expected = []
    for t in exp_terms:
        prod = 1
        for fct in t.split("*"):
            prod *= int(fct.strip())
        expected.append(prod)
    kats["lfp_tensor"] = {
        "source": "crates/latticefold-plus/src/utils.rs:118-131",
        "tensor_product": {"a": vecs[0], "b": vecs[1], "expected": vecs[2]},
        "tensor": {"r": r_in, "expected": expected},
    }

    # --- Frog ring (the ring latticefold-plus runs on): Poseidon table (rings/poseidon/frog.rs:7-1425) = the same 64-bit literals again, embedded
    #     with Fq::from(i128), i.e. reduced mod p_frog; and the challenge-set decoding KAT rings/frog.rs:66-96 (byte - 128 per coefficient), which
    #     is also the map of latticefold-plus' utils::short_challenge(128, ..) (utils.rs:87-101: u = 2^(128/16) = 256, b % u - u/2)
    pf = 15912092521325583641
    srcf = read("crates/cyclotomic-rings/src/rings/poseidon/frog.rs")
    valsf = [int(x, 16) for x in re.findall(r"Fq::from\(0x([0-9a-f]+)_i128\)", srcf)]
    assert len(valsf) == 30 * 24 + 24 * 24
    cfg = re.search(r"PoseidonConfig::<Fq>::new\(full_rounds, partial_rounds, alpha, mds, ark, (\d+), (\d+)\)", srcf)
    kats["poseidon_frog_params"] = {
        "source": "crates/cyclotomic-rings/src/rings/poseidon/frog.rs:7-1425",
        "modulus": pf,
        "full_rounds": int(re.search(r"full_rounds = (\d+)", srcf).group(1)), "partial_rounds": int(re.search(r"partial_rounds = (\d+)", srcf).group(1)),
        "alpha": int(re.search(r"alpha = (\d+)", srcf).group(1)), "rate": int(cfg.group(1)), "capacity": int(cfg.group(2)),
        "same_literals_as_goldilocks": valsf == vals,
        "literals_at_or_above_modulus": sum(1 for v in valsf if v >= pf),
        "ark_first": [v % pf for v in valsf[:4]], "mds_last": [v % pf for v in valsf[-4:]],
        "ark_checksum": sum((i + 1) * (v % pf) for i, v in enumerate(valsf[:720])) % pf,
        "mds_checksum": sum((i + 1) * (v % pf) for i, v in enumerate(valsf[720:])) % pf,
    }
    srcr = read("crates/cyclotomic-rings/src/rings/frog.rs")
    tst = between(srcr, "fn test_small_challenge_from_random_bytes")
    kats["frog_short_challenge"] = {
        "source": "crates/cyclotomic-rings/src/rings/frog.rs:66-96",
        "bytes": [int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})\b", tst)],
        "expected_coeffs": [int(x) for x in re.findall(r"BigInt\(\[(\d+)\]\)", tst)],
    }
    assert len(kats["frog_short_challenge"]["bytes"]) == 16 and len(kats["frog_short_challenge"]["expected_coeffs"]) == 16

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote", os.path.normpath(OUT), "with", len(kats), "KAT groups")


if __name__ == "__main__":
    sys.exit(main())
