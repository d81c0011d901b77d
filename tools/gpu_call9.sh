#!/bin/bash
# round-2 validation of "in-kernel hand-overs for batches of up to four draws"
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/call9_pytest.log 2>&1 || { tail -30 gpurun_out/call9_pytest.log; exit 1; }
tail -3 gpurun_out/call9_pytest.log
python tools/batch_sweep.py 4096 1,2,3,4,8 > gpurun_out/call9_batch.log 2>&1 && cat gpurun_out/call9_batch.log
python tools/batch_sweep.py 1024 1,2,4,8 >> gpurun_out/call9_batch.log 2>&1 && tail -8 gpurun_out/call9_batch.log
python tools/diag_modes.py 4096 1000 > gpurun_out/call9_modes.log 2>&1 && tail -12 gpurun_out/call9_modes.log
