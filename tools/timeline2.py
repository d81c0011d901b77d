"""Unperturbed timeline of kernel STARTS inside one fit+predict step, from the diagnostic build
(make -C gpax_amd/csrc timeline -> tools/libgpx_hip_tl.so): every kernel's first thread logs s_memrealtime
(100 MHz) into a device buffer; potrf_diag (one workgroup) also logs its end.  No profiler, no event brackets.

    python tools/timeline2.py [N] [first_step last_step]
"""
import ctypes as C
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gpax_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(root, "tools", "libgpx_hip_tl.so")
import gpax_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, m = 8, 1000
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
core = gpax_amd.Core(0)
lib = _lib.load()
core.set_train(x); core.set_test(xn); core.set_targets(y, np.ones(m))
for kv in filter(None, os.environ.get("CP_OPTS", "").split(",")):   # e.g. CP_OPTS=pair_min=24,bulk_wait=1
    core.set_option(kv.split("=")[0], int(kv.split("=")[1]))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
for _ in range(5):
    core.fit_predict_async(p, 0.0); core.fetch_results()
lib.gpx_timeline_start()
core.fit_predict_async(p, 0.0); core.fetch_results()
buf = (C.c_ulonglong * (4 * 16000))()
lib.gpx_timeline_read.restype = C.c_longlong
cnt = lib.gpx_timeline_read(buf, C.c_longlong(16000))
rec = np.frombuffer(buf, dtype=np.uint64)[: 4 * cnt].reshape(-1, 4).astype(np.int64)
ends_buf = (C.c_ulonglong * 16384)()
lib.gpx_timeline_read_ends.restype = C.c_longlong
lib.gpx_timeline_read_ends(ends_buf, C.c_longlong(16384))
ends = np.frombuffer(ends_buf, dtype=np.uint64).astype(np.int64)
rec = rec[np.argsort(rec[:, 1])]
t0 = rec[0, 1]
us = (rec[:, 1] - t0) / 100.0
nb = 128
names = {1: "potrf start", 2: "potrf end", 5: "potrf resident", 7: "potrf pub"}
rows = []
for (kind0, tstart, a, b), t in zip(rec, us):
    kind, tl_id = kind0 & 255, (kind0 >> 8) - 1
    dur = f"  ran {(ends[tl_id] - tstart) / 100.0:6.1f} us" if (tl_id >= 0 and ends[tl_id] > 0) else ""
    if tl_id >= 0 and ends[8192 + tl_id] > 0:
        dur += f"  pub at {(ends[8192 + tl_id] - t0) / 100.0:8.1f}"
    b = int(b)
    if kind in (1, 2, 5, 7):
        rows.append((t, names[kind], f"k={a // nb}"))
    elif kind == 3:
        N_, beta = b // 1000000, b % 1000000
        what = "mini-panel" if (a == nb and N_ == nb and not beta) else ("mini-update" if (a == nb and N_ == nb) else ("Q panel" if not beta else "Q col-update"))
        rows.append((t, what, f"M={a} N={N_}{dur}"))
    elif kind == 6:
        N_, beta = b // 1000000, b % 1000000
        rows.append((t, "  resident:", ("mini-update" if beta else "mini-panel") + f" M={a} N={N_}"))
    elif kind in (4, 8):
        rows.append((t, "T bulk" if kind == 4 else "Q col64", f"M={a} N={b // 1000000} K={b % 1000000}{dur}"))
print(f"{cnt} records; step spans {us[-1]:.1f} us from first to last logged start")
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 15)
show = False
for t, what, info in rows:
    if what in ("potrf start", "potrf resident"):
        k = int(info[2:])
        show = lo <= k <= hi
    if show:
        print(f"{t:9.1f}  {what:13s} {info}")
# chain accounting: potrf durations and the gaps between potrf end and the next potrf start
ps = [t for t, w, _ in rows if w == "potrf start"]
pe = [t for t, w, _ in rows if w == "potrf end"]
dur = np.array(pe) - np.array(ps)
gap = np.array(ps[1:]) - np.array(pe[:-1])
print(f"potrf: mean {dur.mean():.1f} us (min {dur.min():.1f}, max {dur.max():.1f}); between potrfs: mean {gap.mean():.1f} us (min {gap.min():.1f}, max {gap.max():.1f})")
print("gap between potrfs by step:", " ".join(f"{g:.0f}" for g in gap))
print("potrf duration by step:   ", " ".join(f"{g:.0f}" for g in dur))
