"""Size sweep of the hot path on ONE MI355X: N in {1024 .. 32768} for the d=8 RBF
workload plus BASELINE.json's five configs, each as points/s, ms/step, whole-step fp64 rate, and the
per-stage roofline figures of the two kernels that carry a roofline (gram -> HBM, trailing update -> MFMA).

    python tools/sweep.py [--out profiles/r01_sweep.json] [--max-n 32768]

Timing: W warm-up steps, then K steps between device synchronisations (value), then the same K steps with the
bulk-update and gram launches timed by their own dispatch timestamps (profile mode 3).  Whole-step flops = N^3/3 (Cholesky) + N^2 (M+1) (border rows:
k_cross and r ride through the panel solves and updates) -- the algorithmic count, not what the kernels issue.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import gpax_amd  # noqa: E402
from scipy.special import ndtr  # noqa: E402

K_NOISE = 0.006744897501960818


def problem(n, d, m, seed=0, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(lo, hi, (n, d))
    x_new = rng.uniform(lo, hi, (m, d))
    y = np.sin(3.0 * x.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(n)
    y = (y - y.mean()) / (y.std() + 1e-8)
    return x, y, x_new


def piecewise(x, t=1.7, b1=4.5, b2=2.5):
    x0 = x[:, 0]
    return np.where(x0 < t, np.abs(x0) ** b1, np.abs(x0) ** b2 + (t ** b1 - t ** b2)) * 1e-2


def run(core, name, kind, n, d, m, steps, warmup, structured=False):
    if structured:
        x, y, x_new = problem(n, d, m, lo=0.0, hi=3.0)
        m_train, m_new = piecewise(x), piecewise(x_new)
    else:
        x, y, x_new = problem(n, d, m)
        m_train, m_new = np.ones(n), np.ones(m)
    params = gpax_amd.make_params(kind, 1.0, np.array([1.0]), None, np.array([K_NOISE]), 1e-6, True)
    core.set_train(x)
    core.set_test(x_new)
    core.set_targets(y - m_train, m_new)

    def step():
        core.fit_predict_async(params, 0.0)
        info, logp, mean, var = core.fetch_results()
        if structured:  # analytic EI over the candidate batch (gpax/acquisition.py:291-298)
            sd, y_max = np.sqrt(var), float(y.max())
            u = (mean - y_max) / sd
            _ = sd * np.exp(-0.5 * u * u - 0.5 * np.log(2 * np.pi)) + (mean - y_max) * ndtr(u)
        return info, logp

    for _ in range(warmup):
        step()
    core.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        info, logp = step()
    core.synchronize()
    sec = (time.perf_counter() - t0) / steps
    assert info == 0 and np.isfinite(logp), (name, info, logp)

    core.profile_reset()
    core.profile_mode(3)   # only the two roofline kernels, by their own dispatch timestamps
    for _ in range(steps):
        step()
    core.synchronize()
    st = {gpax_amd._lib.STAGE_NAMES[i]: core.profile_get(i) for i in range(8)}
    core.profile_mode(0)

    def rate(s, key, scale):
        return s[key] / (s["ms"] * 1e-3) * scale if s["ms"] > 0 else 0.0

    flops = n ** 3 / 3.0 + float(n) * n * (m + 1)
    row = {
        "config": name, "kind": {0: "rbf", 1: "matern52", 2: "periodic"}[kind], "n": n, "d": d, "m": m,
        "ms_per_step": sec * 1e3, "points_per_s": n / sec, "step_tflops": flops / sec * 1e-12,
        "trailing_tflops": rate(st["trailing_update"], "flops", 1e-12),
        "trailing_frac_of_78.6": rate(st["trailing_update"], "flops", 1e-12) / 78.6,
        "gram_gbs": rate(st["gram"], "bytes", 1e-9),
        "gram_frac_of_8000": rate(st["gram"], "bytes", 1e-9) / 8000.0,
        "stage_ms": {k: v["ms"] / steps for k, v in st.items()},
    }
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sweep.json")
    ap.add_argument("--max-n", type=int, default=32768)
    args = ap.parse_args()
    core = gpax_amd.Core(0)
    rows = []
    plan = [(f"N={n}", gpax_amd.RBF, n, 8, 1000, False) for n in (1024, 2048, 4096, 8192, 16384, 32768)]
    plan += [
        ("C1", gpax_amd.RBF, 256, 1, 100, False),
        ("C2", gpax_amd.RBF, 4096, 8, 1000, False),
        ("C3", gpax_amd.MATERN52, 16384, 16, 1000, False),
        ("C4 on one GPU", gpax_amd.RBF, 32768, 8, 1000, False),
        ("C5", gpax_amd.MATERN52, 8192, 2, 10000, True),
    ]
    done = {}
    for name, kind, n, d, m, structured in plan:
        if n > args.max_n:
            continue
        key = (kind, n, d, m, structured)
        if key in done:
            row = dict(done[key], config=name)
        else:
            steps = 20 if n <= 4096 else (5 if n <= 16384 else 2)
            row = run(core, name, kind, n, d, m, steps, 2 if n <= 16384 else 1, structured)
            done[key] = row
        rows.append(row)
        print(json.dumps({k: v for k, v in row.items() if k != "stage_ms"}), flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump({"device": core.device_name(), "rows": rows}, open(args.out, "w"), indent=1)
    print("| config | kernel | N | d | M | ms/step | points/s | step TFLOP/s | trailing TFLOP/s (frac of 78.6) | gram GB/s (frac of 8000) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['config']} | {r['kind']} | {r['n']} | {r['d']} | {r['m']} | {r['ms_per_step']:.3f} | {r['points_per_s']:.3g} | "
              f"{r['step_tflops']:.1f} | {r['trailing_tflops']:.1f} ({r['trailing_frac_of_78.6']:.2f}) | "
              f"{r['gram_gbs']:.0f} ({r['gram_frac_of_8000']:.2f}) |")
    core.close()


if __name__ == "__main__":
    main()
