"""The product side of the calibration path (tests/test_calibration_cpu.py has the loader and the oracle): a synthetic probe object with a permuted slot map
and the floor digit rule goes through latticefold_amd.calibrate onto a product context AND onto the oracle; both must then compute the same words -- CRT,
balanced digits on the tie values, a commitment and a complete fold step -- and those words must differ from what the default conventions give."""
import numpy as np
import pytest

import lfo
from latticefold_amd import api
from latticefold_amd.calibrate import load_probe
from latticefold_amd.workload import make_workload
from test_calibration_cpu import CASES, P, synth_probe_text

pytestmark = pytest.mark.gpu


def test_probe_output_switches_product_and_oracle_together(tmp_path):
    nr0, y0 = lfo.get_ring()
    y0 = y0.copy()
    try:
        y = y0[np.random.default_rng(8).permutation(8)]
        text, _ = synth_probe_text(nr0, y, 1)
        lfo.set_ring(nr0, y0)
        lfo.set_digit_mode(0)
        path = tmp_path / "stark_rings_tables.json"
        path.write_text(text)
        wl = make_workload("T10")
        rng = np.random.default_rng(2)
        x = rng.integers(0, P, size=(64, 24), dtype=np.uint64)
        ties = np.zeros((len(CASES), 24), dtype=np.uint64)
        ties[:, 0] = np.array(CASES, dtype=np.uint64)
        ties[:, 7] = np.array(CASES[::-1], dtype=np.uint64)

        def product(cal):
            ctx = api.Context(0)
            try:
                if cal is not None:
                    cal.apply(ctx)
                ctx.load_ccs(wl)
                scheme = api.AjtaiCommitmentScheme(ctx, matrix=wl.ajtai_matrix())
                wit = api.Witness.from_w_ccs(ctx, wl.w_ccs)
                cccs = np.concatenate([wit.commit(scheme), wl.x_ccs])
                tr = api.PoseidonTranscript()
                acc, _ = api.LFLinearizationProver.prove(ctx, cccs, wit, tr)
                lc, w0, proof = api.NIFSProver.prove(ctx, acc, wit, cccs, wit, tr)
                return ctx.crt(x), ctx.decompose(ties, 1 << 16, 4, 0), cccs, lc, proof, w0.f
            finally:
                ctx.close()

        cal = load_probe(str(path))
        got = product(cal)
        base = product(None)
        cal.apply_oracle(lfo)
        assert (got[0] == lfo.crt(x)).all() and (got[1] == lfo.decompose(ties, 1 << 16, 4, 0)).all()
        assert not (base[0] == got[0]).all() and not (base[1] == got[1]).all()          # the calibration really moved the product
        inst, A = lfo.Instance(wl), wl.ajtai_matrix()
        f = inst.witness_from_w_ccs(wl.w_ccs)
        to = lfo.Transcript()
        cccs = got[2]
        assert (cccs[:wl.kappa] == lfo.ajtai_commit(A, wl.kappa, wl.N, lfo.crt(f))).all()      # the commitment under the calibrated CRT
        acc_o, _ = inst.linearize(to, cccs, f)
        lc_o, f0_o, proof_o = inst.fold_step(to, A, acc_o, f, cccs, f)
        assert (got[3] == lc_o).all() and (got[4] == proof_o).all() and (got[5] == f0_o).all()
    finally:
        lfo.set_ring(nr0, y0)
        lfo.set_digit_mode(0)
