#!/bin/bash
# dataflow potrf_diag in the library: tests, bench, size sweep
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/call11_pytest.log 2>&1 || { tail -30 gpurun_out/call11_pytest.log; exit 1; }
tail -3 gpurun_out/call11_pytest.log
python bench.py > gpurun_out/call11_bench.json 2> gpurun_out/call11_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/call11_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'])
print([ (s['n'], s['ms_per_step']) for s in d.get('sweep',[])])
print(d.get('configs'))
print(d.get('batched'))
PY
