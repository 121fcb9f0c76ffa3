#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
CVD_LIB_VARIANT=tailprof timeout 300 python tools/tail_profile.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $OUT/tail_profile.log
