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


@pytest.mark.parametrize("nranks,n,m,nb,pair_min,quad_min", [(1, 1333, 77, 64, 6, 12), (2, 1333, 77, 64, 6, 12), (3, 1100, 150, 64, 2, 5),
                                                           (8, 2100, 90, 64, 4, 10), (4, 1500, 60, 128, 2, 6)])
def test_loopback_grouped_schedule(nranks, n, m, nb, pair_min, quad_min):
    """The grouped schedule of the RCCL driver (the gathered panels of a group side by side in a K-contiguous group buffer,
    one K = g nb bulk pass per group, block column k+1 brought up to date with the pending panels at every step), played
    on P virtual ranks: groups of 4, 2 and 1 in one factorisation, ragged N, two border blocks."""
    import gpax_amd
    from gpax_amd.dist import LoopbackGroup
    from oracle import gp_oracle as orc

    x, y, xn, p = orc.synthetic_problem(n, 5, m, seed=n + nranks)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    grp = LoopbackGroup(nranks, x, xn, y - 1.0, nb=nb, grouped=True)
    try:
        for r in grp.ranks:
            r.core.set_option("pair_min", pair_min)
            r.core.set_option("quad_min", quad_min)
        info, logp, mean, var = grp.fit_predict(prm, np.ones(m), 0.0)
    finally:
        grp.close()
    logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
    assert info == 0 and abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8


def test_loopback_8_ranks_at_n8192_matches_single_gpu():
    """N = 8192 (64 block columns: groups of 4, 2 and 1 at the default thresholds) on 8 virtual ranks, grouped schedule,
    against the single-GPU path on the same inputs (and both against the solve identity's oracle-free check)."""
    import gpax_amd
    from gpax_amd.dist import LoopbackGroup
    from oracle import gp_oracle as orc

    n, m, d = 8192, 300, 8
    x, y, xn, p = orc.synthetic_problem(n, d, m, seed=5)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    one = gpax_amd.Core(0)
    one.set_train(x)
    one.set_test(xn)
    info1, logp1, mean1, var1 = one.fit_predict(prm, y - 1.0, np.ones(m), 0.0)
    one.close()
    grp = LoopbackGroup(8, x, xn, y - 1.0, grouped=True)
    try:
        info, logp, mean, var = grp.fit_predict(prm, np.ones(m), 0.0)
    finally:
        grp.close()
    assert info == info1 == 0
    assert abs(logp - logp1) <= 1e-10 * abs(logp1)
    assert np.abs(mean - mean1).max() <= 1e-8 and np.abs(var - var1).max() <= 1e-8


@pytest.mark.parametrize("group,p2p", [("1", "0"), ("1", "1"), ("0", "0")])
def test_rccl_driver_grouped_and_p2p_single_rank(group, p2p, monkeypatch):
    """gpx_dist_fit_predict's grouped schedule (default) and its per-peer send/recv transport (GPX_DIST_P2P=1) through the
    one-rank communicator, at a size with groups of 4, 2 and 1 (nb = 64, lowered thresholds)."""
    import gpax_amd
    from gpax_amd.dist import DistributedStep
    from oracle import gp_oracle as orc

    monkeypatch.setenv("GPX_DIST_LOOKAHEAD", "1")
    monkeypatch.setenv("GPX_DIST_EARLY_DIAG", "0")
    monkeypatch.setenv("GPX_DIST_GROUP", group)
    monkeypatch.setenv("GPX_DIST_P2P", p2p)
    x, y, xn, p = orc.synthetic_problem(1333, 8, 150, seed=12)
    prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
    step = DistributedStep(x, xn, y - 1.0, device=0, nb=64, force_comm=True)
    try:
        step.core.set_option("pair_min", 6)
        step.core.set_option("quad_min", 12)
        for _ in range(2):
            info, logp, mean, var = step.fit_predict(prm, np.ones(150), 0.0)
    finally:
        step.close()
    logp_ref, mean_ref, var_ref = orc.fit_predict(0, x, y, xn, p)
    assert info == 0 and abs(logp - logp_ref) <= 1e-10 * abs(logp_ref)
    assert np.abs(mean - mean_ref).max() <= 1e-8 and np.abs(var - var_ref).max() <= 1e-8


def test_two_rank_rccl_parity_when_two_gpus_are_present(tmp_path):
    """Two real ranks over RCCL (one process per GPU) against the single-GPU path, both schedules -- skipped on the
    one-GPU test box: C4 / multi-GPU stays UNVERIFIED on hardware until a node with >= 2 GPUs runs this."""
    import subprocess
    import sys

    import gpax_amd

    try:   # a second device? (asked of the library itself: importing torch here would map its bundled HIP runtime)
        gpax_amd.Core(1).close()
    except gpax_amd.GpxError:
        pytest.skip("needs >= 2 GPUs")
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import gpax_amd
import torch.distributed as dist
from gpax_amd.dist import DistributedStep
from oracle import gp_oracle as orc
dist.init_process_group(backend="gloo")
rank = dist.get_rank()
x, y, xn, p = orc.synthetic_problem(3000, 8, 200, seed=3)
prm = gpax_amd.make_params(0, p["k_scale"], p["k_length"], None, p["k_noise"], p["k_jitter"])
ref = orc.fit_predict(0, x, y, xn, p)
for group in ("1", "0"):
    os.environ["GPX_DIST_GROUP"] = group
    step = DistributedStep(x, xn, y - 1.0, device=rank)
    info, logp, mean, var = step.fit_predict(prm, np.ones(200), 0.0)
    step.close()
    assert info == 0 and abs(logp - ref[0]) <= 1e-10 * abs(ref[0]), (group, logp, ref[0])
    assert np.abs(mean - ref[1]).max() <= 1e-8 and np.abs(var - ref[2]).max() <= 1e-8
dist.barrier()
print("rank", rank, "ok")
"""
    script = tmp_path / "two_rank.py"
    script.write_text(code)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29577", str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
