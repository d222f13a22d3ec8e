#!/bin/bash
# What bounds the GEMM loop: the same shapes with parts of the kernel switched off (results are wrong on purpose).
# Needs the experiment build: gemm.hip compiled with -DRS_GEMM_EXPERIMENT into lib/librs_hip_exp.so.
cd "$(dirname "$0")/.."
export RS_HIP_LIB=$PWD/recommendsystem_amd/lib/librs_hip_exp.so
for d in "$@"; do
    echo "RS_GEMM_DBG=$d  (1: no DMA in k loop, 2: no stores, 4: no MFMA, 8: no barrier, 16: no LDS reads)"
    RS_GEMM_DBG=$d timeout -k 10 120 python tools/gemm_tiles.py 2>&1 | grep -E "^(NT|NN|TN)" | cut -c1-80
done
