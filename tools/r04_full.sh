#!/bin/bash
# full GPU check of the round: every -m gpu test, the default bench line, the all-kernels variant
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04full}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/t_all.log 2>&1; echo "all rc=$?"
tail -22 $OUT/t_all.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 400 python bench.py --steps 20 --warmup 3 2>> $OUT/bench.err | tee $OUT/b.json | python tools/bench_line.py
python -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('secondary', d['secondary_1766_pairs']['value'], d['secondary_1766_pairs']['pcg_iterations_per_lm_iteration']); print(d['kernels_avg_ms']); print(d['cpu_baseline'])"
