#!/bin/bash
# scratch call (GPU): the temporally coarse level -- its tests, rebuild threshold of the dense level with it
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/temporal; mkdir -p $OUT; cd $R
exec < /dev/null
timeout 900 python -m pytest tests/test_gpu_temporal.py "tests/test_gpu_parity.py::test_fused_pcg_tail_matches_the_two_launch_path" tests/test_abi.py -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc $?" >> $OUT/tests.log
B="--no-cpu-baseline --no-secondary --steps 20 --warmup 3"
for x in 0 16 48 64 96 128; do timeout 300 python bench.py $B --opt coarse_rebuild_excess_dense=$x > $OUT/bench_x$x.json 2> $OUT/bench_x$x.err; done
timeout 300 python bench.py $B --opt coarse_level=2 > $OUT/bench_xl2.json 2> /dev/null
tail -15 $OUT/tests.log; for t in x0 x16 x48 x64 x96 x128 xl2; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$t.json").read().strip().splitlines()[-1]); print("$t", round(d["value"],1), round(d["ms_per_step"],3), d["config"].get("pcg_iterations_per_lm_iteration"), d["kernels_launches"]["block_inverse"])
except Exception as e: print("$t failed", e)
PY
done
