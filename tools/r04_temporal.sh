#!/bin/bash
# scratch call (GPU): the temporally coarse level in the pair-sharded mode
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/temporal; mkdir -p $OUT; cd $R
exec < /dev/null
timeout 1500 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_huber.py tests/test_reference_reprojection.py -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/tests.log
timeout 300 python tools/forced_dist_check.py > $OUT/forced.log 2>&1
timeout 400 python tools/shard_sim.py 1 2 8 2>/dev/null | grep "^world" > $OUT/shard_sim.log
tail -15 $OUT/tests.log; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $OUT/forced.log | tail -8; cat $OUT/shard_sim.log
