"""Critical path of the chain inside one fit+predict step, from the diagnostic timeline build
(make -C gpax_amd/csrc timeline -> tools/libgpx_hip_tl.so; nothing here is the product library).

For every block step k the band kernel's first phase (S = A_{k+1,k} W_k^T) can start when ALL of these have arrived:
  W     potrf_diag(k) has published W_k                      (in-kernel flag; the time its flag went up, after its last instruction)
  Qpub  the panel stream's column updates of step k-1 have published their first block row (in-kernel counter; since round 3
        only row 2 of the band kernel waits for it)
  bulk  the bulk launch that last wrote block (k+1,k+1) has ended (stream-level event in front of the band launch);
  Bpub  ... or, where that launch publishes its first block column first (round 3), the time that column was complete
  res   the band kernel itself is resident                    (its launch, behind whatever its stream waits for)
The table gives each arrival relative to potrf_diag(k)'s end, names the LAST one, when the band kernel's phases S and U
started, when potrf_diag(k+1) became resident (pres: its first instruction, before its wait) and when it started (next).

    python tools/critical_path.py [N] [out.txt]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gpax_amd import _lib  # noqa: E402

os.environ["GPX_LIB_COMPAT"] = "1"
_lib.LIB_PATH = os.path.join(root, "tools", "libgpx_hip_tl.so")
import gpax_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, m, nb = 8, int(os.environ.get("CP_M", "1000")), 128   # CP_M: test points (the border rows)
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); y = rng.standard_normal(n)
core = gpax_amd.Core(0)
lib = _lib.load()
for kv in filter(None, os.environ.get("CP_OPTS", "").split(",")):   # e.g. CP_OPTS=chain_overlap=32,pair_min=24
    core.set_option(kv.split("=")[0], int(kv.split("=")[1]))
core.set_train(x); core.set_test(xn); core.set_targets(y, np.ones(m))
p = gpax_amd.make_params(0, 1.0, 1.0, None, np.array([0.0067]), 1e-6)
for _ in range(8):
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
rec = rec[np.argsort(rec[:, 1], kind="stable")]
t0 = rec[0, 1]


def us(t):
    return (t - t0) / 100.0


potrf_start, potrf_end, potrf_res, potrf_pub = {}, {}, {}, {}
band = {}      # step -> dict of phase times, filled in time order
gemms = []     # (start, end, pub, kind, M, N, K/beta)
band_seq = []  # (time, what) for 128x128 one-pass records, in time order
for kind0, t, a, b in rec:
    kind, tl = int(kind0 & 255), int(kind0 >> 8) - 1
    if kind == 1: potrf_start[int(a) // nb] = us(t)
    elif kind == 2: potrf_end[int(a) // nb] = us(t)
    elif kind == 5: potrf_res[int(a) // nb] = us(t)
    elif kind == 7: potrf_pub[int(a) // nb] = us(t)
    elif kind in (3, 6):
        N_, beta = int(b) // 1000000, int(b) % 1000000
        if int(a) == nb and N_ == nb:
            band_seq.append((us(t), kind, beta))
        elif kind == 3:
            gemms.append(dict(start=us(t), end=us(ends[tl]) if tl >= 0 and ends[tl] else None, pub=us(ends[8192 + tl]) if tl >= 0 and ends[8192 + tl] else None,
                              role="panel" if not beta else "col", M=int(a), N=N_, K=nb))
    elif kind in (4, 8):
        gemms.append(dict(start=us(t), end=us(ends[tl]) if tl >= 0 and ends[tl] else None, pub=us(ends[8192 + tl]) if tl >= 0 and ends[tl] and ends[8192 + tl] else None,
                          role="bulk" if kind == 4 else "col", M=int(a), N=int(b) // 1000000, K=int(b) % 1000000))

rows_total = max(g["M"] for g in gemms if g["role"] == "panel") + nb
nblk = n // nb
# the band kernel of step k: records between potrf_end[k-1]-ish and potrf_start[k+1]; take, per step, the LAST
# "S started" (kind 3, beta 0) before potrf_start[k+1] and the first "resident" record after potrf_start[k] - 40 us
steps = sorted(k for k in potrf_end if k + 1 in potrf_start)
lines = []
hdr = f"{'k':>3} {'potrf':>6} | arrivals relative to potrf_diag(k) end [us]: {'W':>5} {'Qpub':>7} {'bulk':>7} {'Bpub':>7} {'res':>7} | {'S':>6} {'U':>6} {'pres':>6} {'next':>6} | last to arrive"
lines.append(hdr)
last_count = {}
for k in steps:
    pe, nxt = potrf_end[k], potrf_start[k + 1]
    # S phase start and U phase start of this step's band kernel: kind-3 records in (pe - 1, nxt)
    s_t = [t for t, kd, beta in band_seq if kd == 3 and not beta and pe - 1.0 <= t <= nxt]
    u_t = [t for t, kd, beta in band_seq if kd == 3 and beta and pe - 1.0 <= t <= nxt]
    S = s_t[-1] if s_t else None
    U = u_t[-1] if u_t else None
    # resident: the first kind-6 record of this band kernel = the last kind-6 record with beta == (phase-0 ? 1 : 0) before S
    res_c = [t for t, kd, beta in band_seq if kd == 6 and (S is None or t <= S) and t >= potrf_start[k] - 60.0]
    res = res_c[0] if res_c else None
    # Qpub: the column updates of step k-1 that publish their first block row (round 2: column k from block row k+1,
    # M = rows_total - (k+1) nb; round 3: columns k and k+1 from block row k+2, M = rows_total - (k+2) nb, which feed
    # row 2 of the band kernel -- NOT on the critical path: listed for the slack, counted as a dependency of S only
    # with CP_QPUB_CRITICAL=1, i.e. for the round-2 library)
    qp = [g for g in gemms if g["role"] == "col" and g["N"] == nb and g["M"] in (rows_total - (k + 1) * nb, rows_total - (k + 2) * nb)
          and g["pub"] and g["start"] < nxt]
    qpub = max(g["pub"] for g in qp) if qp else None
    # bulk: the latest bulk launch that started before S and whose first column is <= k+1 (cf = (rows_total - M) / nb)
    bl = [g for g in gemms if g["role"] == "bulk" and g["end"] and (rows_total - g["M"]) // nb <= k + 1 and (S is None or g["start"] < S)]
    bulk = bl[-1]["end"] if bl else None
    bpub = bl[-1]["pub"] if bl and bl[-1].get("pub") else None   # its first block column published (round 3)
    arr = {"W": potrf_pub.get(k, pe), "Qpub": qpub if os.environ.get("CP_QPUB_CRITICAL") else None,
           "bulk": bpub if bpub is not None else bulk, "res": res}
    have = {kk: v for kk, v in arr.items() if v is not None}
    last = max(have, key=lambda kk: have[kk]) if have else "?"
    last_count[last] = last_count.get(last, 0) + 1

    def rel(v):
        return f"{v - pe:7.1f}" if v is not None else "      -"

    lines.append(f"{k:3d} {pe - potrf_start[k]:6.1f} | {'':47s} {potrf_pub.get(k, pe) - pe:5.1f} {rel(qpub)} {rel(bulk)} {rel(bpub)} {rel(res)} | {rel(S)[1:]} {rel(U)[1:]} {rel(potrf_res.get(k + 1))[1:]} {nxt - pe:6.1f} | {last}")

dur = np.array([potrf_end[k] - potrf_start[k] for k in sorted(potrf_end)])
gap = np.array([potrf_start[k + 1] - potrf_end[k] for k in steps])
try:
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = ""
out = [f"# critical path of the chain, N={n} d={d} M={m}, one fit+predict step (timeline build of HEAD {head or os.environ.get('GPX_HEAD', '?')})",
       f"# step: {us(rec[-1, 1]):.1f} us from the first to the last logged kernel start; potrf_diag mean {dur.mean():.1f} us; "
       f"gap mean {gap.mean():.1f} us (k >= 10: {gap[10:].mean():.1f}, max {gap[10:].max():.1f})",
       "# last-arriving dependency, count over steps: " + ", ".join(f"{k}: {v}" for k, v in sorted(last_count.items()))] + lines
# the bulk stream: is it the chain that waits for it, or the other way round?
bulks = sorted((g for g in gemms if g["role"] == "bulk" and g["end"]), key=lambda g: g["start"])
panels = sorted((g for g in gemms if g["role"] == "panel" and g["end"] and g["M"] > nb), key=lambda g: g["start"])
out.append("")
out.append("# bulk stream T: each launch, the idle time of the stream in front of it, and what it started behind: the previous bulk")
out.append("# launch (T is the bottleneck there) or the panel it applies (the chain is).  busy = sum of launches / span.")
out.append(f"{'cf':>3} {'tiles(M,N,K)':>18} {'start':>8} {'ran':>6} {'idle':>6} {'after panel end':>16}")
prev_end = None
busy = 0.0
for g in bulks:
    cf = (rows_total - g["M"]) // nb
    # the panel whose rows start at block row cf - 2 - near... the latest panel that ended before this launch started
    pe_ = [p_["end"] for p_ in panels if p_["end"] <= g["start"] + 0.5]
    after_panel = g["start"] - max(pe_) if pe_ else float("nan")
    idle = g["start"] - prev_end if prev_end is not None else 0.0
    out.append(f"{cf:3d} {g['M']:6d},{g['N']:5d},{g['K']:4d} {g['start']:8.1f} {g['end'] - g['start']:6.1f} {idle:6.1f} {after_panel:16.1f}")
    busy += g["end"] - g["start"]
    prev_end = g["end"]
if bulks:
    out.append(f"# T busy {busy:.1f} us of {bulks[-1]['end'] - bulks[0]['start']:.1f} us between its first start and last end")
# host enqueue times (diagnostic build): when were step k's launches queued, and how long did each call take?
if hasattr(lib, "gpx_timeline_host"):
    hb = (C.c_double * 1024)()
    nh = lib.gpx_timeline_host(hb, 1024)
    ht = np.array(hb[:nh])
    if nh >= 3:
        ch = ht[1::2][: nblk] - ht[0::2][: nblk]          # chain(k) call
        sd = ht[2::2][: nblk] - ht[1::2][: nblk]          # side(k) call
        out.append("")
        out.append(f"# host: the factor loop enqueued {nblk} steps in {ht[-1]:.0f} us (chain(k) calls {ch.sum():.0f} us, side(k) calls {sd.sum():.0f} us)")
        out.append("host chain(k) us: " + " ".join(f"{v:.0f}" for v in ch))
        out.append("host side(k) us:  " + " ".join(f"{v:.0f}" for v in sd))
        out.append("host t(chain k):  " + " ".join(f"{v:.0f}" for v in ht[0::2][: nblk]))
out.append("gap by step:   " + " ".join(f"{g:.0f}" for g in gap))
out.append("potrf by step: " + " ".join(f"{g:.0f}" for g in dur))
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
