"""latticefold_amd -- MI355X-native (gfx950) LatticeFold prover hot path.

* `latticefold_amd.api`      ctypes mirror of the reference interface over the C ABI `include/lfhip.h`
                             (`Context`, `AjtaiCommitmentScheme`, `Witness`, `PoseidonTranscript`, `LFLinearizationProver`,
                             `NIFSProver`, `NIFSVerifier`, `MLSumcheckLin`); needs `latticefold_amd/liblfhip.so`
                             (`python -c "import __graft_entry__ as g; g.build()"`) and, for everything except the transcript
                             and the verifier, a GPU -- there is no CPU fallback.
* `latticefold_amd.workload` deterministic synthetic workloads (SURVEY 8.0 configurations), numpy only
* `latticefold_amd.ccs`      R1CS -> CCS front-end (`CCS::from_r1cs_padded`), numpy only
* `latticefold_amd.dist`     helpers for the opt-in intra-step sharding (column shards, all-gather + modular sum)
"""
__all__ = ["api", "workload", "ccs", "dist"]
