#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
timeout 300 python tools/tail_profile.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|development variant" | tee $OUT/tail_profile.log
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>> $OUT/bench.err | python tools/bench_line.py
timeout 600 python -m pytest -q -m gpu -x tests/test_gpu_parity.py -k "fused_pcg_tail" tests/test_abi.py 2>&1 | tail -2
