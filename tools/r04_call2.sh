#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
timeout 600 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "fused_pcg_tail" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | grep "assert\|Error\|passed\|failed\|where" | head -30
