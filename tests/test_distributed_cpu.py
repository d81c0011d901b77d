"""world_size-2 ``gloo`` tests (CPU) of the multi-GPU sharding logic: hyper-parameter draws are dealt
round-robin over the ranks with no data-path collective, results are gathered in draw order."""

import os
import socket

import numpy as np
import pytest

from gpax_amd import parallel


def test_partition_covers_every_draw_once():
    for n in (0, 1, 7, 8, 100):
        for ws in (1, 2, 3, 8):
            parts = [parallel.partition_draws(n, r, ws) for r in range(ws)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        parallel.partition_draws(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, ws, port, n_items, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        calls = []

        def fn_batch(idx):  # stands in for ONE batched schedule over this rank's draws (gpx_sample_batch)
            calls.append(list(idx))
            mu = np.stack([np.random.default_rng(1000 + i).standard_normal(5) + i for i in idx])
            return mu, np.stack([np.full((2, 3), float(i)) for i in idx])

        mu, y = parallel.map_draws_batched(fn_batch, n_items)
        # one call with all of the rank's draws -- or none at all when the rank owns no draw
        assert calls == ([parallel.partition_draws(n_items, rank, ws)] if parallel.partition_draws(n_items, rank, ws) else [])
        q.put((rank, mu, y))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 2, 1])
def test_map_draws_batched_world_size_2_matches_serial(n_items):
    """``parallel.map_draws_batched`` (what ``GaussianProcess._sample_over_hps`` calls) over two gloo ranks, incl.
    ``n_items = 1``: rank 1 owns nothing, never calls ``fn_batch`` and still returns the full result."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_mu = np.stack([np.random.default_rng(1000 + i).standard_normal(5) + i for i in range(n_items)])
    ref_y = np.stack([np.full((2, 3), float(i)) for i in range(n_items)])
    for _, mu, y in got:  # identical, complete and in draw order on every rank
        assert np.array_equal(mu, ref_mu) and np.array_equal(y, ref_y)


def test_single_process_path_is_identity():
    mu, = parallel.map_draws_batched(lambda idx: (np.array([[i, 2.0 * i] for i in idx]),), 4)
    assert np.array_equal(mu, np.array([[0, 0], [1, 2], [2, 4], [3, 6.0]]))


# ---- the socket control plane (gpax_amd._control): what bench.py and gpax_amd.dist use at WORLD_SIZE > 1

def _cp_worker(rank, ws, port, n_items, q):
    import re
    import sys

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="t")
    import gpax_amd  # noqa: F401  (the library first, as bench.py does)
    from gpax_amd import _control, _lib

    _lib.load()
    cp = _control.activate(_control.ControlPlane.from_env(timeout=60))
    try:
        assert parallel.world() == (rank, ws)
        cp.barrier()
        mx = cp.max(10.0 + rank)
        uid = cp.broadcast(b"\x01" * 128 if rank == 0 else None)
        calls = []

        def fn_batch(idx):
            calls.append(list(idx))
            return (np.stack([np.random.default_rng(1000 + i).standard_normal(5) + i for i in idx]), np.stack([np.full((2, 3), float(i)) for i in idx]))

        mu, y = parallel.map_draws_batched(fn_batch, n_items)
        mine = parallel.partition_draws(n_items, rank, ws)
        assert calls == ([mine] if mine else [])
        hips = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if re.search(r"libamdhip64", ln)})
        q.put((rank, mx, uid, mu, y, hips, "torch" in sys.modules))
        cp.barrier()
    finally:
        cp.close()
        _control.activate(None)


@pytest.mark.parametrize("ws,n_items", [(2, 7), (3, 2), (2, 1)])
def test_socket_control_plane_matches_serial_and_maps_one_hip_runtime(ws, n_items):
    """bench.py / gpax_amd.dist at WORLD_SIZE > 1 run on ``gpax_amd._control`` (plain sockets): barrier, MAX, broadcast of
    the NCCL unique id, and the gather behind ``parallel.map_draws_batched`` (incl. ranks that own no draw) -- and the
    process keeps exactly ONE libamdhip64 mapped and never imports torch (``import torch.distributed`` maps PyTorch's
    bundled HIP + HSA runtimes beside the /opt/rocm ones the library is bound to)."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cp_worker, args=(r, ws, port, n_items, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_mu = np.stack([np.random.default_rng(1000 + i).standard_normal(5) + i for i in range(n_items)])
    ref_y = np.stack([np.full((2, 3), float(i)) for i in range(n_items)])
    for rank, mx, uid, mu, y, hips, has_torch in got:
        assert mx == 10.0 + ws - 1 and uid == b"\x01" * 128
        assert np.array_equal(mu, ref_mu) and np.array_equal(y, ref_y)
        assert len(hips) == 1 and not has_torch, (hips, has_torch)
