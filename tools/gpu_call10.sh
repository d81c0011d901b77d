#!/bin/bash
# potrf_diag stand-alone: residuals, per-stage stamps, the streaming-solve probe (MFMA / VALU contention on one SIMD)
set -e
mkdir -p gpurun_out
timeout -k 5 60 ./tools/potrf_stamps > gpurun_out/stamps_flow.txt 2>&1
cat gpurun_out/stamps_flow.txt
