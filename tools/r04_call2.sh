#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
timeout 300 python tests/golden/reference_py/make_reprojection_golden.py dump gpurun_out/reproj_dump.npz 2>&1 | tail -2
