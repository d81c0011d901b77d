#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/c3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c3/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c3/pytest.log
./tools/peaks > gpurun_out/c3/peaks.txt 2>&1; tail -8 gpurun_out/c3/peaks.txt
timeout -k 10 600 python bench.py > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/c3/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c3/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','hip_runtime')})
print('roofline', {k:d['roofline'][k] for k in ('achieved','frac','bound','avg_launch_us','ceilings')})
print('configs', d['configs'])
print('batched', d['batched'])
PY
GPX_LIB_COMPAT=1 python tools/timeline2.py 4096 0 31 > gpurun_out/c3/timeline_4096.txt 2>&1; tail -4 gpurun_out/c3/timeline_4096.txt
