#!/bin/bash
# Collects the round's committed evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic passes, SQ counters.
# usage (through gpurun): bash tools/profile_round.sh <tag>       (copy the summaries from gpurun_out/<tag> into profiles/)
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
python $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
$B --steps 20 --warmup 3 --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
$B --steps 20 --warmup 3 --pairs-level 1 --time-all-kernels > $OUT/bench_1766_allkernels.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 20 --warmup 3 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1766 -- $B --steps 20 --warmup 3 --pairs-level 1 > $OUT/trace1766.log 2>&1
# HBM traffic of the hot kernel: separate passes per counter (MI355X_MICROARCH.md), lockstep PCG (no early-exit launches)
CVD_PCG_LOCKSTEP=1 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_matvec_pairs_fast" --output-format csv -d $OUT/pmc_fetch -- $B --steps 4 --warmup 1 > $OUT/pmc_fetch.log 2>&1
CVD_PCG_LOCKSTEP=1 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_matvec_pairs_fast" --output-format csv -d $OUT/pmc_write -- $B --steps 4 --warmup 1 > $OUT/pmc_write.log 2>&1
# SQ counters of the PCG kernels (wave cycles: parked / issue-stalled / active; VALU and LDS activity)
K="k_matvec_pairs_fast|k_assemble_fast|k_cg_update|k_matvec_finish"
CVD_PCG_LOCKSTEP=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq_a -- $B --steps 4 --warmup 1 > $OUT/pmc_sq_a.log 2>&1
CVD_PCG_LOCKSTEP=1 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq_b -- $B --steps 4 --warmup 1 > $OUT/pmc_sq_b.log 2>&1
# BASELINE configs[4] (1000 frames 640x384, 16x12 grid), Cauchy and Huber; dense mode (configs[2] video, 60 and 300 frames)
python $R/bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_config4_cauchy.json 2>> $OUT/bench.err
python $R/bench.py --config 4 --robust huber --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_config4_huber.json 2>> $OUT/bench.err
python $R/bench.py --dense --frames 60 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --time-all-kernels > $OUT/bench_dense_60.json 2>> $OUT/bench.err
python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_dense_300.json 2>> $OUT/bench.err
python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --time-all-kernels > $OUT/bench_dense_300_allkernels.json 2>> $OUT/bench.err
CVD_DENSE_MATRIX_FREE=1 python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --time-all-kernels > $OUT/bench_dense_300_matrix_free.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dense -- python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/trace_dense.log 2>&1
python $R/tools/kernel_durations.py $OUT/trace_dense $TAG > $OUT/kernel_durations_dense.txt 2>&1
rm -rf $OUT/trace_dense
python $R/tools/lm_trace.py 300 > $OUT/pipeline.log 2>&1
CVD_PAIRS_LEVEL=6 python $R/tools/lm_trace.py 300 > $OUT/pipeline_4140.log 2>&1
# summaries (small, committed under profiles/)
python $R/tools/kernel_durations.py $OUT/trace $TAG > $OUT/kernel_durations.txt 2>&1
python $R/tools/kernel_durations.py $OUT/trace1766 $TAG > $OUT/kernel_durations_1766.txt 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_FETCH_SIZE.csv > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_write $OUT/pmc_WRITE_SIZE.csv > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_sq_a $OUT/pmc_SQ_a.csv > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_sq_b $OUT/pmc_SQ_b.csv > /dev/null 2>&1
python $R/tools/pmc_to_json.py $OUT $TAG > $OUT/pmc_matvec_pairs.json 2> $OUT/pmc_to_json.err
cp $OUT/trace/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
cp $OUT/trace1766/*/*kernel_stats.csv $OUT/bench_1766_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace $OUT/trace1766 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq_a $OUT/pmc_sq_b
tail -c 400 $OUT/bench.json; echo; for f in config4_cauchy config4_huber dense_60 dense_300 dense_300_matrix_free; do python -c "import sys,json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['constraints'])"; done; grep TOTAL $OUT/pipeline.log $OUT/pipeline_4140.log | cut -c1-220; head -12 $OUT/kernel_durations.txt | cut -c1-200; cat $OUT/pmc_matvec_pairs.json | head -12; head -6 $OUT/pmc_SQ_a.csv
