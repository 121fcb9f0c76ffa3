#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
timeout 400 python tools/shard_sim.py 1 2 4 8 2>&1 | grep "^world\|Error\|error" | tee $OUT/shard_sim.log
timeout 300 python tools/shard_sim.py 8 --replicated 2>&1 | grep "^world\|Error\|error" | tee -a $OUT/shard_sim.log
