"""Bit-for-bit comparison of two builds of the library on the same inputs (schedule refactors must not move a bit).
  python tools/bits_r1.py dump <lib.so|default> <out.npz>     run the cases, save every output
  python tools/bits_r1.py cmp <a.npz> <b.npz>                 array_equal on every saved array
"""
import os, sys, time
import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

CASES = [  # (n, d, m, kind, ard)
    (25, 1, 300, 0, False), (100, 2, 33, 1, False), (256, 1, 100, 0, False), (300, 4, 21, 2, True), (640, 8, 129, 1, False),
    (1100, 3, 40, 0, True), (2000, 8, 100, 0, False), (4096, 8, 1000, 0, False), (6300, 8, 257, 0, False), (8192, 2, 1000, 1, False),
]


def dump(lib, out):
    from gpax_amd import _lib
    if lib != "default":
        os.environ["GPX_LIB_COMPAT"] = "1"
        _lib.LIB_PATH = os.path.abspath(lib)
    import gpax_amd
    c = gpax_amd.Core(0)
    res = {}
    for n, d, m, kind, ard in CASES:
        rng = np.random.default_rng(n + d)
        x = rng.uniform(-1, 1, (n, d)); xn = rng.uniform(-1, 1, (m, d)); r = rng.standard_normal(n)
        ell = rng.uniform(0.7, 1.5, d) if ard else np.array([1.1])
        per = rng.uniform(0.8, 1.6, d) if kind == 2 else None
        p = gpax_amd.make_params(kind, 1.3, ell, per, np.array([0.0067]), 1e-6)
        c.set_train(x); c.set_test(xn)
        t0 = time.perf_counter()
        info, logp, mean, var = c.fit_predict(p, r, np.ones(m), 0.0)
        key = f"n{n}_"
        res[key + "fp"] = np.concatenate([[info, logp], mean, var])
        if n <= 4096:
            info, _, l = c.fit_factor(p, want_l=True)
            res[key + "L"] = l
            info, lp, alpha = c.log_marginal(r, want_alpha=True)
            res[key + "lm"] = np.concatenate([[info, lp], alpha])
            info, mean2, var2, cov = c.posterior(np.ones(m), 0.01, want_cov=(m <= 300))
            res[key + "post"] = np.concatenate([mean2, var2] + ([cov.ravel()] if cov is not None else []))
        info, lp, grads, alpha = c.lml_grad(p, r)
        res[key + "grad"] = np.concatenate([[info, lp], alpha] + [np.ravel(v) for _, v in sorted(grads.items())])
        print(f"n={n} d={d} m={m} kind={kind}: logp={logp:.10g} ({time.perf_counter() - t0:.2f} s)", flush=True)
    np.savez(out, **res)


def cmp(a, b):
    za, zb = np.load(a), np.load(b)
    bad = 0
    for k in za.files:
        same = np.array_equal(za[k], zb[k], equal_nan=True)
        if not same:
            bad += 1
            d = np.abs(za[k] - zb[k])
            print(f"DIFF {k}: max abs {np.nanmax(d):.3e} at {int(np.nanargmax(d))} of {d.size}")
    print("bit-identical" if not bad else f"{bad} arrays differ", f"({len(za.files)} arrays)")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3])
    else:
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
