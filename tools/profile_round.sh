#!/bin/bash
# Collects the round's committed evidence on the GPU box: bench lines, rocprofv3 kernel stats, PMC traffic passes, SQ counters.
# usage (through gpurun): bash tools/profile_round.sh <tag>       (copy the summaries from gpurun_out/<tag> into profiles/)
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
exec < /dev/null
# 1. HBM traffic of the hot kernel (separate passes per counter, MI355X_MICROARCH.md; lockstep PCG: no early-exit launches)
#    -> profiles/pmc_matvec_pairs.json, which the bench line below reads for roofline.traffic (tied to the kernel sources by hash)
bash $R/tools/pmc_refresh.sh $TAG > $OUT/pmc_refresh.log 2>&1
mv $OUT/bench_full.json $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
CMD_MAIN="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing"
$B --steps 20 --warmup 3 --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 20 --warmup 3 --no-kernel-timing > $OUT/trace.log 2>&1
# SQ counters of the PCG kernels and the dense inverse (wave cycles: parked / issue-stalled / active; VALU and LDS activity)
K="k_matvec_pairs_fast|k_pcg_tail|k_cg_update|k_matvec_finish|k_dense_spd_inverse|k_tl_edges|k_coarse_edges_fast"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq_a -- $B --steps 4 --warmup 1 --pcg-lockstep > $OUT/pmc_sq_a.log 2>&1
# BASELINE configs[4] (1000 frames 640x384, 16x12 grid), Cauchy and Huber; dense mode (configs[2] video, 300 frames)
python $R/bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_config4_cauchy.json 2>> $OUT/bench.err
python $R/bench.py --config 4 --robust huber --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_config4_huber.json 2>> $OUT/bench.err
python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --time-all-kernels > $OUT/bench_dense_300.json 2>> $OUT/bench.err
CMD_DENSE="python bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dense -- python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing > $OUT/trace_dense.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_cross_matvec" --output-format csv -d $OUT/pmc_fetch_dense -- python $R/bench.py --dense --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --pcg-lockstep > $OUT/pmc_fetch_dense.log 2>&1
python $R/tools/kernel_durations.py $OUT/trace_dense $TAG "$CMD_DENSE" > $OUT/kernel_durations_dense.txt 2>&1
rm -rf $OUT/trace_dense
# the dense inverse alone, what one rank of an N-rank run computes, run-to-run spread of the dense-level solve
[ -x $R/tools/dinv_bench.bin ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-cuda-compat -I$R/include -I$R/robust_cvd_amd/csrc $R/tools/dinv_bench.hip -o $R/tools/dinv_bench.bin
for n in 1000 2400 4096; do timeout 100 $R/tools/dinv_bench.bin $n >> $OUT/dinv_bench.log 2>&1; done
timeout 300 python $R/tools/shard_sim.py 1 2 4 8 2>/dev/null | grep "^world" > $OUT/shard_sim.log
timeout 200 python $R/tools/shard_sim.py 8 --replicated 2>/dev/null | grep "^world" >> $OUT/shard_sim.log
# where the workgroups of the fused PCG tail kernel spend their life (stamp variant of the library, built before the call)
[ -f $R/robust_cvd_amd/lib/libcvd_hip_tailprof.so ] && timeout 200 python $R/tools/tail_profile.py 2>/dev/null | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > $OUT/tail_profile.log
# the two-launch tail for comparison (cvd_solver_options::pcg_fused_tail = 0)
$B --steps 20 --warmup 3 --time-all-kernels --opt pcg_fused_tail=0 > $OUT/bench_allkernels_two_launch_tail.json 2>> $OUT/bench.err
# the preconditioner's levels one by one: round 3's (exact dense pose-graph level, no temporal depth-grid level), + the depth-grid level,
# + the temporal pose level at weight 1 (the default adds temporal_weight = 0.7)
$B --steps 20 --warmup 3 --opt coarse_over_budget=1 --opt temporal_level=0 > $OUT/bench_levels_round3.json 2>> $OUT/bench.err
$B --steps 20 --warmup 3 --opt coarse_over_budget=1 --opt temporal_weight=1 > $OUT/bench_levels_depth_grid.json 2>> $OUT/bench.err
$B --steps 20 --warmup 3 --opt temporal_weight=1 > $OUT/bench_levels_temporal_pose.json 2>> $OUT/bench.err
timeout 300 python $R/tools/parity_probe.py "eta=1e-3" "eta=1e-3,temporal_weight=1" "eta=1e-3,coarse_over_budget=1,temporal_weight=1" "eta=1e-3,coarse_over_budget=1,temporal_level=0" 2>/dev/null | grep "^config" > $OUT/parity_probe_levels.log
timeout 200 python $R/tools/dense_coarse_probe.py 4 0 2>/dev/null | cut -c1-120 > $OUT/dense_coarse_probe.log
# summaries (small, committed under profiles/)
python $R/tools/kernel_durations.py $OUT/trace $TAG "$CMD_MAIN" > $OUT/kernel_durations.txt 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_sq_a $OUT/pmc_SQ_a.csv > /dev/null 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_fetch_dense $OUT/pmc_FETCH_SIZE_dense.csv > /dev/null 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_sq_a $OUT/pmc_fetch_dense
for f in round3 depth_grid temporal_pose; do python -c "import json; d=json.loads(open('$OUT/bench_levels_$f.json').read().strip().splitlines()[-1]); print('levels_$f', round(d['value'],1), d['ms_per_step'], d['config']['pcg_iterations_per_lm_iteration'])"; done
cat $OUT/parity_probe_levels.log
tail -c 300 $OUT/bench.json; echo; for f in config4_cauchy config4_huber dense_300; do python -c "import sys,json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['constraints'])"; done; head -12 $OUT/kernel_durations.txt | cut -c1-200; cat $OUT/pmc_matvec_pairs.json | head -12; head -6 $OUT/pmc_SQ_a.csv; cat $OUT/shard_sim.log $OUT/dense_coarse_probe.log $OUT/dinv_bench.log
