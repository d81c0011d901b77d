"""Distributed (one process per GPU) exact-GP step: ONE covariance matrix sharded by block rows,
1-D block-cyclic, RCCL broadcast of the panel factor + all-gather of the panel column per step
(include/gpx_dist.h, gpax_amd/csrc/dist.hip; SURVEY.md section 8e.2, BASELINE config C4).

Two drivers over the same compute phases:
  * ``DistributedStep.fit_predict``  -- production: one process per GPU, RCCL inside the library; the
    NCCL unique id travels over the control plane (``gpax_amd._control`` sockets, or ``torch.distributed`` if the
    caller initialised it), nothing else does.
  * ``LoopbackGroup``                -- validation on ONE GPU: P virtual ranks in one process, the
    transport played by device-to-device copies.  Used by tests/test_gpu_dist.py; not a fallback.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np

from gpax_amd import _lib
from gpax_amd._lib import Core, _arr, _ptr


class _Rank:
    """One rank's handle + local layout."""

    def __init__(self, core, rank, nranks, x, x_new, y_minus_m):
        self.core, self.rank, self.nranks = core, rank, nranks
        self.lib, self.h = core._lib, core._h
        x, x_new = _arr(x), _arr(x_new).reshape(-1, np.asarray(x).shape[1])
        self.n, self.d, self.m = x.shape[0], x.shape[1], x_new.shape[0]
        core._check(self.lib.gpx_dist_configure(self.h, rank, nranks, _ptr(x), self.n, self.d, _ptr(x_new), self.m), "gpx_dist_configure")
        r = _arr(y_minus_m).reshape(-1)
        core._check(self.lib.gpx_dist_set_targets(self.h, _ptr(r)), "gpx_dist_set_targets")
        nsteps, nloc, owner, ntest = C.c_int64(), C.c_int64(), C.c_int(), C.c_int64()
        core._check(self.lib.gpx_dist_info(self.h, C.byref(nsteps), C.byref(nloc), C.byref(owner), C.byref(ntest)), "gpx_dist_info")
        self.nsteps, self.r_owner, self.n_test_local = nsteps.value, owner.value, ntest.value

    def buffers(self):
        w, send, recv, z = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        wc, cc, zc = C.c_int64(), C.c_int64(), C.c_int64()
        self.core._check(
            self.lib.gpx_dist_buffers(self.h, C.byref(w), C.byref(wc), C.byref(send), C.byref(recv), C.byref(cc), C.byref(z), C.byref(zc)),
            "gpx_dist_buffers",
        )
        return dict(w=w.value, w_count=wc.value, send=send.value, recv=recv.value, chunk=cc.value, z=z.value, z_count=zc.value)

    def local_results(self, m_new):
        mn = _arr(m_new).reshape(-1) if m_new is not None else None
        nt = max(self.n_test_local, 1)
        logdet, quad, info = C.c_double(), C.c_double(), C.c_int()
        idx = (C.c_int64 * nt)()
        mean, var = np.zeros(nt), np.zeros(nt)
        self.core._check(
            self.lib.gpx_dist_local_results(self.h, _ptr(mn), C.byref(logdet), C.byref(quad), idx, _ptr(mean), _ptr(var), C.byref(info)),
            "gpx_dist_local_results",
        )
        k = self.n_test_local
        return logdet.value, quad.value, np.array(idx[:k], dtype=np.int64), mean[:k], var[:k], info.value


class LoopbackGroup:
    """P virtual ranks on ONE GPU; collectives are device-to-device copies.  Test transport only."""

    def __init__(self, nranks, x, x_new, y_minus_m, device=0, nb=128, early_diag=False, grouped=False):
        """``early_diag``: play the early-diagonal schedule of ``gpx_dist_fit_predict`` (the owner of block k+1 updates
        and factorises its diagonal block from its own panel rows, before the all-gather of panel k).
        ``grouped``: play its grouped schedule (the default of the RCCL driver): the gathered panels of a group go side
        by side into a K-contiguous group buffer and the bulk pass applies them in one K = g nb launch."""
        self.early_diag = early_diag
        self.grouped = grouped and not early_diag
        self.P = nranks
        self.ranks = []
        for r in range(nranks):
            core = Core(device)
            core.set_option("nb", nb)
            self.ranks.append(_Rank(core, r, nranks, x, x_new, y_minus_m))
        self.n, self.m = self.ranks[0].n, self.ranks[0].m

    def close(self):
        for r in self.ranks:
            r.core.close()

    def fit_predict(self, params, m_new=None, new_noise=0.0):
        p, _keep = params
        R = self.ranks
        for r in R:
            r.core._check(r.lib.gpx_dist_build(r.h, C.byref(p), float(new_noise)), "gpx_dist_build")
        bufs = [r.buffers() for r in R]
        last_t, group_end, gbuf = -1, -1, 0
        for k in range(R[0].nsteps):
            owner = k % self.P
            if not (self.early_diag and k >= 1):   # early schedule: block k was factorised during step k-1
                R[owner].core._check(R[owner].lib.gpx_dist_step_diag(R[owner].h, k), "step_diag")
            R[owner].core.synchronize()
            for q in range(self.P):  # "ncclBroadcast" of W_k from the owner
                if q != owner:
                    R[q].core._check(R[q].lib.gpx_dist_copy(R[q].h, bufs[q]["w"], bufs[owner]["w"], bufs[q]["w_count"]), "bcast W")
            for r in R:
                r.core._check(r.lib.gpx_dist_step_panel(r.h, k), "step_panel")
                r.core.synchronize()
            if self.early_diag and k + 1 < R[0].nsteps:   # before the "all-gather": only the owner's own panel rows are needed
                nxt = R[(k + 1) % self.P]
                nxt.core._check(nxt.lib.gpx_dist_step_diag_early(nxt.h, k + 1), "step_diag_early")
                nxt.core.synchronize()
            for q in range(self.P):  # "ncclAllGather": every rank's send chunk into slot r of every recv buffer
                for r in range(self.P):
                    dst = bufs[q]["recv"] + r * bufs[q]["chunk"] * 8
                    R[q].core._check(R[q].lib.gpx_dist_copy(R[q].h, dst, bufs[r]["send"], bufs[q]["chunk"]), "allgather")
            if self.grouped:   # the grouped driver: column k+1 gets every panel since the last bulk pass; the bulk pass at group ends
                nblk = R[0].nsteps
                if k > group_end:
                    group_end = min(nblk - 1, k + R[0].lib.gpx_dist_group_size(R[0].h, nblk - k) - 1)
                kfirst, npan = last_t + 1, k - last_t
                for r in R:
                    r.core._check(r.lib.gpx_dist_step_gather_group(r.h, k, k - kfirst, gbuf), "step_gather_group")
                    r.core._check(r.lib.gpx_dist_step_update_group(r.h, kfirst, npan, k + 1, 1, 0, gbuf), "step_update_group (next)")
                    if k == group_end:   # near block (the next group's columns) and the bulk pass behind it, as the RCCL driver splits them
                        g_next = r.lib.gpx_dist_group_size(r.h, nblk - (k + 1)) if k + 1 < nblk else 0
                        near = max(0, min(g_next, nblk - (k + 2)))
                        if near > 0:
                            r.core._check(r.lib.gpx_dist_step_update_group(r.h, kfirst, npan, k + 2, near, 0, gbuf), "step_update_group (near)")
                        r.core._check(r.lib.gpx_dist_step_update_group(r.h, kfirst, npan, k + 2 + near, -1, 0, gbuf), "step_update_group (bulk)")
                    r.core.synchronize()
                if k == group_end:
                    last_t, gbuf = k, gbuf ^ 1
                continue
            for r in R:  # the look-ahead driver's split of the update: block column k+1 first, then the rest
                r.core._check(r.lib.gpx_dist_step_gather(r.h, k), "step_gather")
                if self.early_diag:
                    r.core._check(r.lib.gpx_dist_step_update_next_col(r.h, k), "step_update_next_col")
                else:
                    r.core._check(r.lib.gpx_dist_step_update_cols(r.h, k, k + 1, 1, 0), "step_update_cols (next)")
                r.core._check(r.lib.gpx_dist_step_update_cols(r.h, k, k + 2, -1, 0), "step_update_cols (rest)")
                r.core.synchronize()
        owner = R[0].r_owner
        R[owner].core._check(R[owner].lib.gpx_dist_extract_z(R[owner].h), "extract_z")
        R[owner].core.synchronize()
        for q in range(self.P):  # "ncclBroadcast" of z
            if q != owner:
                R[q].core._check(R[q].lib.gpx_dist_copy(R[q].h, bufs[q]["z"], bufs[owner]["z"], bufs[q]["z_count"]), "bcast z")
        mean, var = np.zeros(self.m), np.zeros(self.m)
        logdet, info, quad = 0.0, 0, 0.0
        for r in R:  # "ncclAllReduce"
            ld, qd, idx, ml, vl, inf = r.local_results(m_new)
            logdet += ld
            quad = qd
            info = inf if info == 0 else (min(info, inf) if inf > 0 else info)  # first failing pivot over all ranks
            mean[idx] = ml
            var[idx] = vl
        logp = -0.5 * quad - logdet - 0.5 * self.n * math.log(2 * math.pi)
        return info, logp, mean, var


class DistributedStep:
    """Production driver: this process is rank ``rank`` of ``nranks`` (one per GPU); RCCL moves the panel
    factors inside ``gpx_dist_fit_predict``.  The 128-byte NCCL unique id travels over the active control plane:
    ``gpax_amd._control`` (plain sockets; what ``bench.py`` uses: no second HIP runtime in the process) or, if the caller
    initialised it, ``torch.distributed`` (any backend)."""

    def __init__(self, x, x_new, y_minus_m, device=None, nb=128, force_comm=False):
        """``force_comm``: create the RCCL communicator even at one rank, so that every collective of the driver
        runs (as a self-copy) -- how the RCCL calls are exercised on a one-GPU box (tests/test_gpu_dist.py)."""
        from gpax_amd import _control, parallel

        cp = _control.active()
        self.rank, self.nranks = parallel.world()
        self.core = Core(self.rank if device is None else device)
        self.core.set_option("nb", nb)
        self.r = _Rank(self.core, self.rank, self.nranks, x, x_new, y_minus_m)
        if self.nranks > 1 or force_comm:
            uid = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(128)
                self.core._check(self.core._lib.gpx_dist_unique_id(buf), "gpx_dist_unique_id")
                uid = [buf.raw]
            if self.nranks > 1:
                if cp is not None:
                    uid = [cp.broadcast(uid[0])]
                else:
                    import torch.distributed as dist

                    dist.broadcast_object_list(uid, src=0)
            self.core._check(self.core._lib.gpx_dist_comm_init(self.core._h, uid[0], self.rank, self.nranks), "gpx_dist_comm_init")

    def fit_predict(self, params, m_new=None, new_noise=0.0):
        p, _keep = params
        mn = _arr(m_new).reshape(-1) if m_new is not None else None
        logp = C.c_double()
        mean, var = np.empty(max(self.r.m, 1)), np.empty(max(self.r.m, 1))
        info = self.core._check(
            self.core._lib.gpx_dist_fit_predict(self.core._h, C.byref(p), _ptr(mn), float(new_noise), C.byref(logp), _ptr(mean), _ptr(var)),
            "gpx_dist_fit_predict",
        )
        return info, logp.value, mean[: self.r.m], var[: self.r.m]

    def close(self):
        self.core._lib.gpx_dist_comm_destroy(self.core._h)
        self.core.close()
