#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04a; mkdir -p $OUT; cd $R
timeout 300 python tests/golden/reference_py/make_reprojection_golden.py dump gpurun_out/reproj_dump.npz > $OUT/dump.log 2>&1; echo "dump rc=$?"
timeout 1200 python -m pytest -q -m gpu -x tests/test_gpu_dense_inverse.py tests/test_lib_python.py tests/test_gpu_two_ranks.py "tests/test_gpu_baseline_configs.py::test_end_state_matches_the_oracle_solution" tests/test_reference_reprojection.py --durations=8 > $OUT/t.log 2>&1; echo "tests rc=$?"; tail -25 $OUT/t.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>> $OUT/bench.err | tee $OUT/b.json | python tools/bench_line.py
