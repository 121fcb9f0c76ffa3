#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -q -m gpu -x tests/test_gpu_dense_mode.py -k real_resolution --durations=3 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>> $OUT/bench.err | python tools/bench_line.py
timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --no-kernel-timing 2>> $OUT/bench.err | python tools/bench_line.py
