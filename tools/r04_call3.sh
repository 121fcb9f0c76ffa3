#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
for o in 1 0; do
echo "== finish_in_product $o"; timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --time-all-kernels --opt pcg_finish_in_product=$o 2>> $OUT/bench.err | tee $OUT/bfp$o.json | python tools/bench_line.py
python -c "
import json; d=json.loads(open('$OUT/bfp$o.json').read().strip().splitlines()[-1]); print(d['kernels_avg_ms']); print('secondary', d['secondary_1766_pairs']['value'], d['secondary_1766_pairs']['pcg_iterations_per_lm_iteration'])"
done
timeout 900 python -m pytest -q -m gpu -x tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -k "not config4" 2>&1 | tail -4
