#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03d}
mkdir -p $OUT
cd $R
timeout 120 tools/dinv_bench.bin 2400 | head -12
timeout 1200 python -m pytest tests -q -m gpu -x > $OUT/t_all.log 2>&1; echo "all rc=$?"
tail -6 $OUT/t_all.log
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3"
for o in "" "--opt coarse_rebuild_excess=32" "--pairs-level 1"; do
  timeout 300 $B $o > $OUT/b.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('[$o]', 'value %.1f' % d['value'], 'ms %.3f' % d['ms_per_step'], 'pcg/it %.2f' % d['config']['pcg_iterations_per_lm_iteration'], 'pipeline %.4f' % d['pipeline']['seconds'], 'hot us %.1f' % (d['roofline']['avg_launch_ms']*1e3))
PY
done
timeout 300 $B --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench_allkernels.json').read().strip().splitlines()[-1]); print(d['kernels_avg_ms'])"
