#!/bin/bash
# library check after a kernel change: potrf_diag stand-alone, GPU tests, bench line, step timeline
set -e
mkdir -p gpurun_out
timeout -k 5 60 ./tools/potrf_stamps > gpurun_out/stamps_flow.txt 2>&1
grep "whole kernel\|max|" gpurun_out/stamps_flow.txt
python -m pytest tests -m gpu -x -q > gpurun_out/call11_pytest.log 2>&1 || { tail -30 gpurun_out/call11_pytest.log; exit 1; }
tail -3 gpurun_out/call11_pytest.log
python bench.py > gpurun_out/call11_bench.json 2> gpurun_out/call11_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/call11_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'])
print([ (s['n'], round(s['ms_per_step'],3)) for s in d.get('sweep',[])])
print([(c['config'], round(c['ms_per_step'],3)) for c in d.get('configs',[])])
print([(b['B'], b['n'], round(b['ms_per_batch'],3)) for b in d.get('batched',[])])
PY
GPX_LIB_COMPAT=1 python3 tools/timeline2.py 4096 12 13 > gpurun_out/tl4096.txt 2>&1
tail -3 gpurun_out/tl4096.txt
