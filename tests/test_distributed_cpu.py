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
