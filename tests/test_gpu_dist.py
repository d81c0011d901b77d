"""GPU tests of the distributed (row-block-cyclic) factorisation: P virtual ranks on ONE GPU with a
loopback transport (device-to-device copies in place of ncclBroadcast / ncclAllGather / ncclAllReduce), so
every index map, GEMM and reduction of gpax_amd/csrc/dist.hip runs for real; plus the RCCL driver at
world size 1.  (The reference has no multi-GPU path: the check is against the single-GPU path and the oracle.)"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
@pytest.mark.parametrize("n,m,d,kind", [(700, 90, 4, 0), (1024, 300, 8, 1)])
def test_loopback_matches_oracle(nranks, n, m, d, kind):
    import gpax_amd
    from gpax_amd.dist import LoopbackGroup
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(n, d, m, seed=n + nranks, kind=kind)
    prm = gpax_amd.make_params(kind, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    grp = LoopbackGroup(nranks, x, xn, y - 1.0)
    try:
        info, logp, mean, var = grp.fit_predict(prm, np.ones(m), 0.0)
    finally:
        grp.close()
    logp_ref, mean_ref, var_ref = orc.fit_predict(kind, x, y, xn, p)
    assert info == 0
    assert abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= 1e-8 * max(1.0, np.abs(mean_ref).max())
    assert np.abs(var - var_ref).max() <= 1e-8


def test_loopback_vector_noise_and_not_pd():
    import gpax_amd
    from gpax_amd.dist import LoopbackGroup
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(500, 3, 40, seed=4)
    noise = np.full(500, 0.01)
    noise[77] = 0.2
    prm = gpax_amd.make_params(0, 1.0, 1.0, None, noise, 1e-6)
    grp = LoopbackGroup(4, x, xn, y - 1.0)
    try:
        info, logp, mean, var = grp.fit_predict(prm, np.ones(40), 0.0)
        k = orc.rbf(x, x, 1.0, 1.0, noise, 1e-6)
        logp_ref = orc.mvn_log_prob(y, np.ones(500), k)[0]
        assert info == 0 and abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
        noise[300] = -5.0
        prm = gpax_amd.make_params(0, 1.0, 1.0, None, noise, 1e-6)
        info, logp, _, _ = grp.fit_predict(prm, np.ones(40), 0.0)
        assert info == 301 and not np.isfinite(logp)
    finally:
        grp.close()


def test_rccl_driver_world_size_1():
    """gpx_dist_fit_predict end to end (no communicator needed at one rank): same answer as the single-GPU path."""
    import gpax_amd
    from gpax_amd.dist import DistributedStep
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(900, 8, 77, seed=9)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    step = DistributedStep(x, xn, y - 1.0, device=0)
    try:
        info, logp, mean, var = step.fit_predict(prm, np.ones(77), 0.0)
    finally:
        step.close()
    logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
    assert info == 0 and abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_loopback_early_diagonal_schedule(nranks):
    """The early-diagonal schedule (owner of block k+1 factorises it from its own panel rows before the all-gather of
    panel k; block column k+1 is then updated without that block row): same results as the oracle, not-PD index included."""
    import gpax_amd
    from gpax_amd.dist import LoopbackGroup
    from oracle import gp_oracle as orc

    n, m, d = 900, 140, 5
    x, y, xn, p = orc.synthetic_problem(n, d, m, seed=31 + nranks)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    grp = LoopbackGroup(nranks, x, xn, y - 1.0, early_diag=True)
    try:
        info, logp, mean, var = grp.fit_predict(prm, np.ones(m), 0.0)
        logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
        assert info == 0 and abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
        assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8
        noise = np.full(n, 0.01)
        noise[400] = -5.0
        info, logp, _, _ = grp.fit_predict(gpax_amd.make_params(0, 1.0, 1.0, None, noise, 1e-6), np.ones(m), 0.0)
        assert info == 401 and not np.isfinite(logp)
    finally:
        grp.close()


@pytest.mark.parametrize("lookahead", ["1", "0", "early"])
def test_rccl_driver_single_rank_communicator(lookahead, monkeypatch):
    """The RCCL calls themselves (ncclCommInitRank, ncclBroadcast of W_k and z, the shrinking ncclAllGather of the
    panel, the two ncclAllReduce of the epilogue) on the driver's streams, through a one-rank communicator: the only
    way to run them on a one-GPU box.  Same answer as the oracle; N not a multiple of nb, M > nb (two border blocks)."""
    import gpax_amd
    from gpax_amd.dist import DistributedStep
    from oracle import gp_oracle as orc

    monkeypatch.setenv("GPX_DIST_LOOKAHEAD", "1" if lookahead == "early" else lookahead)
    monkeypatch.setenv("GPX_DIST_EARLY_DIAG", "1" if lookahead == "early" else "0")
    x, y, xn, p = orc.synthetic_problem(1100, 8, 150, seed=11)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    step = DistributedStep(x, xn, y - 1.0, device=0, force_comm=True)
    try:
        for _ in range(2):   # the second call reuses the communicator, events and buffers
            info, logp, mean, var = step.fit_predict(prm, np.ones(150), 0.0)
    finally:
        step.close()
    logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
    assert info == 0 and abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8
