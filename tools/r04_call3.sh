#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
for o in 8 6 5 4 3 0; do
echo "== fused, split $o"; timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --opt coarse_dense_row_split=$o 2>> $OUT/bench.err | python tools/bench_line.py
done
for o in 8 5 3; do
echo "== unfused, split $o"; timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 --opt coarse_dense_row_split=$o --opt pcg_fused_tail=0 2>> $OUT/bench.err | python tools/bench_line.py
done
timeout 600 python -m pytest -q -m gpu -x tests/test_gpu_parity.py -k "fused_pcg_tail or coarse_level_variants" tests/test_gpu_two_ranks.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
