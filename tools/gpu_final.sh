#!/bin/bash
# end-of-round check: the whole GPU test suite, then the judged evidence (tools/refresh_profiles.sh)
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1 || { tail -30 gpurun_out/final_pytest.log; exit 1; }
tail -3 gpurun_out/final_pytest.log
bash tools/refresh_profiles.sh > gpurun_out/final_refresh.log 2>&1
tail -3 gpurun_out/final_refresh.log
