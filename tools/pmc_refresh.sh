#!/bin/bash
# Regenerates profiles/pmc_matvec_pairs.json (HBM traffic of the hot kernel, tied to the kernel sources by hash) after a
# source change that leaves the kernel itself alone.  usage (through gpurun): bash tools/pmc_refresh.sh <tag>
TAG=${1:-pmc}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
$B --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err   # (workload description for pmc_to_json.py)
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_matvec_pairs_fast" --output-format csv -d $OUT/pmc_fetch -- $B --steps 4 --warmup 1 --pcg-lockstep > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_matvec_pairs_fast" --output-format csv -d $OUT/pmc_write -- $B --steps 4 --warmup 1 --pcg-lockstep > $OUT/pmc_write.log 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_FETCH_SIZE.csv > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_write $OUT/pmc_WRITE_SIZE.csv > /dev/null 2>&1
python $R/tools/pmc_to_json.py $OUT $TAG > $OUT/pmc_matvec_pairs.json 2> $OUT/pmc_to_json.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/pmc_matvec_pairs.json $R/profiles/pmc_matvec_pairs.json
python $R/bench.py --steps 20 --warmup 3 > $OUT/bench_full.json 2>> $OUT/bench.err
python -c "
import json; d=json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
