#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03c}
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_dense_inverse.py tests/test_gpu_two_ranks.py -q -m gpu > $OUT/t_new.log 2>&1; echo "new tests rc=$?"
tail -3 $OUT/t_new.log
B="python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3"
for o in "" "--opt coarse_rebuild_excess=0" "--opt coarse_rebuild_excess=8" "--opt coarse_rebuild_excess=32" "--opt coarse_rebuild_excess=64"; do
  timeout 300 $B $o > $OUT/b.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('[$o]', 'value %.1f' % d['value'], 'ms %.3f' % d['ms_per_step'], 'pcg/it %.2f' % d['config']['pcg_iterations_per_lm_iteration'], 'pipeline %.4f' % d['pipeline']['seconds'], 'hot us %.1f' % (d['roofline']['avg_launch_ms']*1e3))
PY
done
timeout 300 $B --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench_allkernels.json').read().strip().splitlines()[-1]); print(d['kernels_avg_ms']); print(d['kernels_launches'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/trace.log 2>&1; echo "trace rc=$?"
python $R/tools/kernel_durations.py $OUT/trace r03c > $OUT/kernel_durations.txt 2>&1
rm -rf $OUT/trace
sed -n 4,12p $OUT/kernel_durations.txt | cut -c1-170
