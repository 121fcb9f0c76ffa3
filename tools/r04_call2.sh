#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
tools/coop_probe.bin 2>&1 | tee $OUT/coop_probe.log
