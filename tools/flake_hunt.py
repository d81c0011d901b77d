"""Runs two GPU API tests in a loop to hunt a rare failure: python tools/flake_hunt.py [lib.so|default] [reps]"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from gpax_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "default":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import test_gpu_gp_api as t
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for i in range(reps):
    for fn in (t.test_acquisition_ei_ucb_and_factor_reuse, t.test_fit_recovers_noise_scale, t.test_fast_predict_matches_oracle_with_transforms,
               t.test_get_mvn_posterior_matches_golden):
        try:
            fn()
        except AssertionError:
            bad += 1
            print(f"rep {i}: {fn.__name__} FAILED"); traceback.print_exc(limit=2)
print(f"{os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else 'default'}: {bad} failures in {reps} repetitions")
