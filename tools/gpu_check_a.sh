#!/bin/bash
# round-3 GPU check: new kernels first, then the whole GPU suite, the bench line and the forced-sharded comparison
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03a}
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_dense_inverse.py -x -q -m gpu > $OUT/t_dense_inverse.log 2>&1; echo "dense_inverse rc=$?"
tail -5 $OUT/t_dense_inverse.log
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -x -q -m gpu > $OUT/t_two_ranks.log 2>&1; echo "two_ranks rc=$?"
tail -5 $OUT/t_two_ranks.log
timeout 1200 python -m pytest tests -q -m gpu > $OUT/t_all.log 2>&1; echo "all rc=$?"
tail -15 $OUT/t_all.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench.err
python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'pcg/it', d['config']['pcg_iterations_per_lm_iteration'], 'cold', d['cold_first_solve'], 'pipeline', d['pipeline'])
    print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'cpu', d.get('cpu_baseline',{}).get('cost_rel_diff_after_iteration'))
    print('secondary', d.get('secondary_1766_pairs'))
except Exception as e:
    print('bench parse failed', e)
PY
timeout 600 python tools/forced_dist_check.py 6 > $OUT/forced_dist.log 2>&1; echo "forced rc=$?"
tail -8 $OUT/forced_dist.log
