#!/bin/bash
# Collects the round's committed evidence on the GPU box: bench lines, rocprofv3 kernel stats, PMC traffic passes, SQ counters,
# LM timeline, workgroup stamps of the hot product.
# usage (through gpurun): bash tools/profile_round.sh <tag>       (then copy the summaries from gpurun_out/<tag> into profiles/)
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
exec < /dev/null
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids"
# 1. HBM traffic of the hot kernel (separate passes per counter, MI355X_MICROARCH.md; lockstep PCG: no early-exit launches)
#    -> profiles/pmc_matvec_pairs.json, which the bench line below reads for roofline.traffic (tied to the kernel sources by hash);
#    its last step is the FULL default bench line (cpu_baseline included) -> bench.json
bash $R/tools/pmc_refresh.sh $TAG > $OUT/pmc_refresh.log 2>&1
mv $OUT/bench_full.json $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
CMD_MAIN="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timing"
$B --steps 20 --warmup 3 --time-all-kernels > $OUT/bench_allkernels.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 20 --warmup 3 --no-kernel-timing > $OUT/trace.log 2>&1
python $R/tools/lm_timeline.py $OUT/trace > $OUT/lm_timeline.txt 2>&1
python $R/tools/solve_timeline.py $OUT/trace 3 > $OUT/solve_timeline.txt 2>&1
# SQ counters of the PCG kernels (wave cycles: parked / issue-stalled / active; VALU and LDS activity)
K="k_matvec_pairs_fast|k_pcg_tail|k_assemble_fast"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-include-regex "$K" --output-format csv -d $OUT/pmc_sq_a -- $B --steps 4 --warmup 1 --pcg-lockstep > $OUT/pmc_sq_a.log 2>&1
# BASELINE configs[4] (1000 frames 640x384, 16x12 grid), Cauchy and Huber; dense mode (configs[2] video, 300 frames)
python $R/bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_config4_cauchy.json 2>> $OUT/bench.err
python $R/bench.py --config 4 --robust huber --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_config4_huber.json 2>> $OUT/bench.err
# dense mode: kernel trace, SQ counters and HBM traffic of the one-walk assembly (tools/dense_profile.sh writes into the same directory
# and profiles/pmc_dense_walk.json, which the dense bench line below reads for roofline.traffic)
bash $R/tools/dense_profile.sh $TAG 300 > $OUT/dense_profile.log 2>&1
python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --time-all-kernels > $OUT/bench_dense_300.json 2>> $OUT/bench.err
python $R/bench.py --dense --steps 4 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_dense_300_plain.json 2>> $OUT/bench.err   # (only the hot kernel timed: the line's value without the event pairs around every class)
# what one rank of an N-rank run computes per PCG iteration (pair-sharded, phantom communicator)
timeout 300 python $R/tools/shard_sim.py 1 2 4 8 2>/dev/null | grep "^world" > $OUT/shard_sim.log
timeout 400 python $R/tools/shard_sim.py 1 2 4 8 --dense 2>/dev/null | grep "^world" >> $OUT/shard_sim.log
timeout 400 python $R/tools/shard_sim.py 1 2 4 8 --config4 2>/dev/null | grep "^world" >> $OUT/shard_sim.log
# where the workgroups of the hot product spend their life (stamp variant of the library, built before the call)
[ -f $R/robust_cvd_amd/lib/libcvd_hip_mvprof.so ] && timeout 200 python $R/tools/mv_profile.py 2>/dev/null | grep -v "$F" > $OUT/mv_profile.log
# the two-launch tail for comparison (cvd_solver_options::pcg_fused_tail = 0)
$B --steps 20 --warmup 3 --time-all-kernels --opt pcg_fused_tail=0 > $OUT/bench_allkernels_two_launch_tail.json 2>> $OUT/bench.err
# LDS read-modify-write throughput (tools/lds_atomic_bench.hip)
[ -x $R/tools/lds_atomic_bench.bin ] && timeout 100 $R/tools/lds_atomic_bench.bin > $OUT/lds_atomic_bench.log 2>&1
# summaries (small, committed under profiles/)
python $R/tools/kernel_durations.py $OUT/trace $TAG "$CMD_MAIN" > $OUT/kernel_durations.txt 2>&1
python $R/tools/pmc_summary.py $OUT/pmc_sq_a $OUT/pmc_SQ_a.csv > /dev/null 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace $OUT/pmc_sq_a
tail -c 400 $OUT/bench.json; echo
for f in config4_cauchy config4_huber dense_300; do python -c "import sys,json; d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['constraints'], d.get('dense_kernels'))"; done
head -8 $OUT/kernel_durations.txt | cut -c1-200; head -12 $OUT/pmc_matvec_pairs.json; head -5 $OUT/pmc_SQ_a.csv; cat $OUT/shard_sim.log; head -12 $OUT/lm_timeline.txt
