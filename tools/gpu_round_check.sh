#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03e}
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/t_all.log 2>&1; echo "all rc=$?"
tail -5 $OUT/t_all.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
timeout 300 $B 2>> $OUT/bench.err | tee $OUT/b.json | python tools/bench_line.py
python -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('secondary', d['secondary_1766_pairs']['value'], d['secondary_1766_pairs']['pcg_iterations_per_lm_iteration'])"
timeout 300 $B --no-secondary --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench_allkernels.json').read().strip().splitlines()[-1]); print(d['kernels_avg_ms'])"
