#!/bin/bash
# One GPU call of a round (through gpurun): the -m gpu suite `reps` times, the default bench line, optional extras.
# usage: bash tools/gpu_round_check.sh <tag> [reps] [extra: "reproj" | "w3" | "allk"]...
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; REPS=${2:-1}; shift; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
for i in $(seq 1 $REPS); do
  CVD_MARGIN_LOG=$OUT/margins_$i.jsonl timeout 1200 python -m pytest tests -q -m gpu --durations=8 > $OUT/t_all_$i.log 2>&1; echo "suite run $i rc=$?"
  tail -14 $OUT/t_all_$i.log | grep -v "$F"
done
[ -f $OUT/margins_1.jsonl ] && python tools/margin_report.py $OUT/margins_*.jsonl > $OUT/margins.log; head -12 $OUT/margins.log
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
timeout 300 $B 2>> $OUT/bench.err | tee $OUT/b.json | python tools/bench_line.py
python -c "
import json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('secondary', d['secondary_1766_pairs']['value'], d['secondary_1766_pairs']['pcg_iterations_per_lm_iteration'])"
for x in "$@"; do
  case $x in
    w3) timeout 300 $B --no-secondary --lib-variant w3 2>> $OUT/bench.err | tee $OUT/b_w3.json | python tools/bench_line.py ;;
    allk) timeout 300 $B --no-secondary --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
      python -c "
import json; d=json.loads(open('$OUT/bench_allkernels.json').read().strip().splitlines()[-1]); print(d['kernels_avg_ms'])" ;;
    reproj) timeout 300 python tests/golden/reference_py/make_reprojection_golden.py dump gpurun_out/reproj_dump.npz 2>&1 | grep -v "$F" | tail -2
      timeout 600 python tools/reproj_repeat.py 5 2>&1 | grep "^run" | tee $OUT/reproj_repeat.log ;;
  esac
done
