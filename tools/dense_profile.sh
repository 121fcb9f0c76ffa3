#!/bin/bash
# Dense-mode evidence on the GPU box: per-kernel durations (rocprofv3 --kernel-trace) and the SQ counters of the one-walk assembly.
# usage (through gpurun): bash tools/dense_profile.sh <tag> [frames]
TAG=${1:-dense}
FR=${2:-300}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
exec < /dev/null
cd /tmp && export TMPDIR=/tmp
CMD="python bench.py --dense --frames $FR --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dense -- python $R/bench.py --dense --frames $FR --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/trace_dense.log 2>&1
python $R/tools/kernel_durations.py $OUT/trace_dense $TAG "$CMD" > $OUT/kernel_durations_dense.txt 2>&1
rm -rf $OUT/trace_dense
K="k_dense_walk|k_dense_gg|k_dense_fold_cross|k_assemble_fast"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --dense --frames $FR --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_sq $OUT/pmc_SQ_dense.csv > /dev/null 2>&1
rm -rf $OUT/pmc_sq
# second SQ pass: where the issue stalls of the walk come from (LDS queue, matrix pipe, scalar unit)
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --kernel-include-regex "k_dense_walk|k_dense_gg" --output-format csv -d $OUT/pmc_sq_b -- python $R/bench.py --dense --frames $FR --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/pmc_sq_b.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_sq_b $OUT/pmc_SQ_dense_b.csv > /dev/null 2>&1
rm -rf $OUT/pmc_sq_b
# HBM traffic of the walk (separate passes per counter: MI355X_MICROARCH.md; FETCH_SIZE is doubled when read against a byte count)
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-include-regex "k_dense_walk|k_dense_gg" --output-format csv -d $OUT/pmc_$CNT -- python $R/bench.py --dense --frames $FR --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/pmc_$CNT.log 2>&1
  python $R/tools/pmc_summary.py $OUT/pmc_$CNT $OUT/pmc_${CNT}_dense.csv > /dev/null 2>&1
done
# -> profiles/pmc_dense_walk.json: what `bench.py --dense` reports as roofline.traffic (tied to the kernel sources by hash)
python $R/bench.py --dense --frames $FR --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_dense_cfg.json 2>> $OUT/bench.err   # (workload description)
python $R/tools/pmc_to_json.py $OUT $TAG dense > $OUT/pmc_dense_walk.json 2> $OUT/pmc_to_json_dense.err && cp $OUT/pmc_dense_walk.json $R/profiles/pmc_dense_walk.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
head -14 $OUT/kernel_durations_dense.txt | cut -c1-180
cat $OUT/pmc_SQ_dense.csv
cat $OUT/pmc_SQ_dense_b.csv
cat $OUT/pmc_FETCH_SIZE_dense.csv $OUT/pmc_WRITE_SIZE_dense.csv
