#!/bin/bash
# dataflow potrf_diag: stand-alone check (residuals + stamps), old kernel beside it
set -e
mkdir -p gpurun_out
timeout -k 5 60 ./tools/potrf_stamps_old > gpurun_out/stamps_old.txt 2>&1
timeout -k 5 60 ./tools/potrf_stamps > gpurun_out/stamps_flow.txt 2>&1
cat gpurun_out/stamps_flow.txt
