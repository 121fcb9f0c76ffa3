#!/bin/bash
# Collects the round's committed evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic passes.
# usage (through gpurun): bash tools/profile_round.sh <tag>
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1
CVD_PCG_LOCKSTEP=1 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_matvec_pairs_fast" --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_fetch.log 2>&1
CVD_PCG_LOCKSTEP=1 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_matvec_pairs_fast" --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc_write.log 2>&1
CVD_VERBOSE=0 python $R/tools/lm_trace.py 300 > $OUT/pipeline.log 2>&1
tail -1 $OUT/bench.json | cut -c1-300
grep TOTAL $OUT/pipeline.log | cut -c1-200
