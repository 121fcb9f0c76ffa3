#!/bin/bash
# scratch call (GPU): the temporal pose level (coarse_level 3)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/temporal; mkdir -p $OUT; cd $R
exec < /dev/null
timeout 400 python tools/parity_probe.py "eta=1e-3" "eta=1e-3,coarse_level=3" "eta=1e-3,coarse_level=3,coarse_temporal_step=16" "eta=1e-3,coarse_level=3,coarse_temporal_step=4" "eta=1e-3,coarse_level=3,pcg_fused_tail=0" --configs=config2_4k,config2 > $OUT/probe.log 2>&1; echo "probe rc $?" >> $OUT/probe.log
B="--no-cpu-baseline --steps 20 --warmup 3"
for t in 1 3; do timeout 300 python bench.py $B --opt coarse_level=$t > $OUT/bench_c$t.json 2> $OUT/bench_c$t.err; done
timeout 300 python bench.py $B --opt coarse_level=3 --opt coarse_temporal_step=16 > $OUT/bench_c3s16.json 2>/dev/null
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" $OUT/probe.log | tail -14; for t in c1 c3 c3s16; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$t.json").read().strip().splitlines()[-1]); print("$t", round(d["value"],1), round(d["ms_per_step"],3), d["config"].get("pcg_iterations_per_lm_iteration"), d["kernels_avg_ms"], "secondary", round(d["secondary_1766_pairs"]["value"],1), d["secondary_1766_pairs"]["pcg_iterations_per_lm_iteration"])
except Exception as e: print("$t failed", e); print(open("$OUT/bench_$t.err").read()[-1500:] if "$t" in ("c1","c3") else "")
PY
done
