#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03b}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_two_ranks.py -q -m gpu > $OUT/t_two_ranks.log 2>&1; echo "two_ranks rc=$?"
tail -5 $OUT/t_two_ranks.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "coarse or sharded or two_level" > $OUT/t_coarse.log 2>&1; echo "coarse rc=$?"
tail -3 $OUT/t_coarse.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 800 $OUT/bench.err
python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'pcg/it', d['config']['pcg_iterations_per_lm_iteration'], 'cold', d['cold_first_solve'], 'pipeline', d['pipeline'])
    print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'cpu', d.get('cpu_baseline',{}).get('cost_rel_diff_after_iteration'), d.get('cpu_baseline',{}).get('value'))
    print('secondary', d.get('secondary_1766_pairs'))
except Exception as e:
    print('bench parse failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/trace.log 2>&1; echo "trace rc=$?"
python $R/tools/kernel_durations.py $OUT/trace r03b > $OUT/kernel_durations.txt 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
head -30 $OUT/kernel_durations.txt | cut -c1-180
