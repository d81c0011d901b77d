#!/bin/bash
set -e
mkdir -p gpurun_out
python - <<'PY'
import ctypes
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so.7")
lo, hi = ctypes.c_int(), ctypes.c_int()
print("rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "lo", lo.value, "hi", hi.value)
PY
for n in 1024 4096; do python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; python tools/ab_lib.py tools/libgpx_hip_old.so $n; python tools/ab_lib.py default $n; done
GPX_LIB_COMPAT=1 python3 tools/timeline2.py 4096 12 13 > gpurun_out/tl4096.txt 2>&1
tail -3 gpurun_out/tl4096.txt
