"""Control plane of multi-process runs without PyTorch: barrier, MAX of a clock, a byte broadcast and an array gather over
plain TCP sockets in a star around rank 0 (``RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT`` from the
environment, as ``torch.distributed.run`` exports them).

Why not ``torch.distributed``: importing it maps PyTorch's bundled ``libamdhip64.so`` / ``libhsa-runtime64.so`` (another
ROCm release) beside the ``/opt/rocm`` runtime ``libgpx_hip.so`` is bound to -- two HIP and two HSA runtimes in one process
(checked with ``/proc/self/maps``).  Nothing here touches the data path: the draws of a hyper-parameter
batch are independent (gpax/gp.py:384-390: a ``vmap``), and the distributed factorisation exchanges its panels over RCCL
inside the library; this carries the NCCL unique id, the barriers around a timed region and the small per-draw results.

The launcher's own rendezvous store listens on ``MASTER_PORT``; the control plane uses the first free port above it
(``GPX_CONTROL_PORT`` overrides), found by rank 0 and probed by the others with a handshake that names the job.
"""

from __future__ import annotations

import os
import pickle
import socket
import struct
import time

import numpy as np

_MAGIC = b"GPXCTL1"


def _send(sock, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv(sock) -> bytes:
    hdr = b""
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        hdr += chunk
    (n,) = struct.unpack("<Q", hdr)
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


class ControlPlane:
    """``ControlPlane(rank, world, addr, port)``; ``from_env()`` reads the launcher's variables.  Every collective is a
    gather to rank 0 followed by a broadcast of the result (world sizes here are <= 8)."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29500, job="", timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self._peers, self._sock = [], None
        if self.world <= 1:
            return
        token = _MAGIC + job.encode()
        ports = [int(os.environ["GPX_CONTROL_PORT"])] if os.environ.get("GPX_CONTROL_PORT") else [port + 1 + i for i in range(32)]
        if self.rank == 0:
            srv, err = None, None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind(("", p))
                    break
                except OSError as e:   # taken (e.g. by another job): the clients skip it too, their handshake fails there
                    srv.close()
                    srv, err = None, e
            if srv is None:
                raise OSError(f"control plane: no free port in {ports[0]}..{ports[-1]}: {err}")
            srv.listen(self.world)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.settimeout(timeout)
                try:
                    hello = _recv(conn)
                except (ConnectionError, socket.timeout):
                    conn.close()
                    continue
                if not hello.startswith(token + b":"):
                    conn.close()
                    continue
                r = int(hello[len(token) + 1:])
                _send(conn, token)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[r] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            deadline = time.monotonic() + timeout
            while self._sock is None:
                for p in ports:
                    try:
                        s = socket.create_connection((addr, p), timeout=2.0)
                        s.settimeout(5.0)
                        _send(s, token + b":" + str(self.rank).encode())
                        if _recv(s) == token:
                            s.settimeout(timeout)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self._sock = s
                            break
                        s.close()
                    except (OSError, ConnectionError):
                        continue
                if self._sock is None:
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"control plane: rank {self.rank} found no rank 0 at {addr}:{ports[0]}..{ports[-1]}")
                    time.sleep(0.05)

    @classmethod
    def from_env(cls, timeout=120.0):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   int(os.environ.get("MASTER_PORT", "29500")), os.environ.get("TORCHELASTIC_RUN_ID", ""), timeout)

    # ---- collectives
    def _reduce_bcast(self, obj, fn):
        """Every rank contributes ``obj``; rank 0 applies ``fn`` to the list (rank order) and everyone gets the result."""
        if self.world <= 1:
            return fn([obj])
        if self.rank == 0:
            items = [obj] + [pickle.loads(_recv(c)) for c in self._peers]
            out = pickle.dumps(fn(items), protocol=pickle.HIGHEST_PROTOCOL)
            for c in self._peers:
                _send(c, out)
            return pickle.loads(out)
        _send(self._sock, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return pickle.loads(_recv(self._sock))

    def barrier(self):
        self._reduce_bcast(None, lambda items: None)

    def max(self, value):
        return float(self._reduce_bcast(float(value), max))

    def sum(self, value):
        return float(self._reduce_bcast(float(value), sum))

    def broadcast(self, obj):
        """``obj`` of rank 0 on every rank (e.g. the 128 bytes of the NCCL unique id)."""
        return self._reduce_bcast(obj if self.rank == 0 else None, lambda items: items[0])

    def all_gather(self, obj):
        """The list of every rank's ``obj`` in rank order, on every rank (NumPy arrays travel as they are)."""
        return self._reduce_bcast(obj, lambda items: items)

    def close(self):
        for c in self._peers:
            c.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None


_active = None


def activate(cp):
    """Make ``cp`` the control plane ``gpax_amd.parallel`` / ``gpax_amd.dist`` use (None: back to torch.distributed, if
    the caller initialised it, or single process)."""
    global _active
    _active = cp
    return cp


def active():
    return _active


def gather_arrays(cp, local, n_items, owner_of):
    """Reassemble per-item arrays computed by their owners: ``local`` = list of arrays of this rank's items (all one
    shape), ``owner_of(rank)`` = the item indices of a rank.  Returns ``(n_items, *shape)`` on every rank."""
    parts = cp.all_gather([np.asarray(a, dtype=np.float64) for a in local])
    shape = next((p[0].shape for p in parts if p), None)
    if shape is None:
        return np.zeros((0,))
    full = np.zeros((n_items,) + shape)
    for r, p in enumerate(parts):
        for j, idx in enumerate(owner_of(r)):
            full[idx] = p[j]
    return full
