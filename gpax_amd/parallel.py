"""Multi-GPU sharding of the path (one process per GPU; control plane = ``gpax_amd._control`` sockets when one is active,
else ``torch.distributed`` if the caller initialised a process group -- e.g. gloo in the CPU tests).

The reference has no distributed code at all (gpax/state.py:64 ``# TODO: multi-gpu support?``).  The
path shards over INDEPENDENT units: gpax's ``vmap`` over hyper-parameter draws
(gpax/gp.py:354-364, 384-390) -- each draw is a complete gram + factor + posterior with no
interaction -- so draws are dealt round-robin over the ranks and there is NO collective on the data
path; the only communication is the final gather of the small per-draw results (mu, sd, samples).
"""

from __future__ import annotations

import numpy as np


def world():
    """(rank, world_size) of the active control plane / process group, (0, 1) without one."""
    from gpax_amd import _control

    cp = _control.active()
    if cp is not None:
        return cp.rank, cp.world
    import sys

    if "torch" not in sys.modules and "torch.distributed" not in sys.modules:
        return 0, 1   # never the one to import PyTorch (a second HIP runtime)
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def partition_draws(n_items, rank, world_size):
    """Indices of the hyper-parameter draws rank ``rank`` owns: round-robin, so that every rank gets
    ceil or floor of n/world draws whatever the order of the draws."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, n_items, world_size))


def gather_draws(local, n_items, group=None):
    """Reassemble per-draw results computed under ``partition_draws``.

    ``local``: list of arrays (one per owned draw, all of one shape) on this rank.  Returns an array of
    shape ``(n_items, *shape)`` in draw order on EVERY rank.  One ``all_gather`` of a padded buffer; on a
    GPU node the backend may be "nccl" (= RCCL over xGMI), on CPU tests it is "gloo"."""
    rank, ws = world()
    if ws == 1:
        return np.stack(local) if local else np.zeros((0,))
    from gpax_amd import _control

    cp = _control.active()
    if cp is not None:
        return _control.gather_arrays(cp, local, n_items, lambda r: partition_draws(n_items, r, ws))
    import torch
    import torch.distributed as dist

    per = (n_items + ws - 1) // ws
    shape = None
    if local:
        shape = tuple(np.asarray(local[0]).shape)
    shapes = [None] * ws
    dist.all_gather_object(shapes, shape, group=group)
    shape = next(s for s in shapes if s is not None)
    buf = torch.zeros((per,) + shape, dtype=torch.float64)
    for j, a in enumerate(local):
        buf[j] = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
    out = [torch.zeros_like(buf) for _ in range(ws)]
    dist.all_gather(out, buf, group=group)
    full = np.zeros((n_items,) + shape)
    for r in range(ws):
        for j, idx in enumerate(partition_draws(n_items, r, ws)):
            full[idx] = out[r][j].numpy()
    return full


def map_draws_batched(fn_batch, n_items, group=None):
    """The vmap over hyper-parameter draws, sharded: ``fn_batch`` takes ALL of a rank's draw indices at once (the HIP path
    runs them as one batched schedule, ``gpx_sample_batch``) and returns a tuple of arrays stacked over those draws;
    returns a tuple of arrays stacked over all ``n_items`` draws in draw order, identical on every rank.  A rank that
    owns no draw (fewer draws than ranks) does not call ``fn_batch`` and learns the number and shapes of the outputs
    from the others."""
    rank, ws = world()
    idx = partition_draws(n_items, rank, ws)
    outs = fn_batch(idx) if idx else None
    n_out = len(outs) if outs is not None else None
    if ws > 1:
        from gpax_amd import _control

        cp = _control.active()
        if cp is not None:
            counts = cp.all_gather(n_out)
        else:
            import torch.distributed as dist

            counts = [None] * ws
            dist.all_gather_object(counts, n_out, group=group)
        n_out = next(c for c in counts if c is not None)
    return tuple(gather_draws(list(outs[k]) if outs is not None else [], n_items, group) for k in range(n_out))
