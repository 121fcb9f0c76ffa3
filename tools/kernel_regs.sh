#!/bin/bash
# Development aid: VGPR / SGPR / spill / LDS figures of the kernels of one translation unit (device-only compile, code-object metadata).
# usage: bash tools/kernel_regs.sh cvd_matvec [name regex] [extra -D flags]
U=$1; FILT=${2:-.}; shift; shift
T=$(mktemp -d)
R=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-cuda-compat --cuda-device-only --no-gpu-bundle-output -c $R/robust_cvd_amd/csrc/$U.hip -o $T/dev.co "$@" || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?(?=\.name:|\Z)', txt, re.S):
    blk=m.group(0); name=m.group(1)
    if not re.search(sys.argv[1], name): continue
    g=lambda k: (re.search(r'\.'+k+r':\s+(\d+)', blk) or [0,'?'])[1]
    print(name[:100], 'vgpr', g('vgpr_count'), 'agpr', g('agpr_count'), 'sgpr', g('sgpr_count'), 'spill', g('vgpr_spill_count'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'), 'sgpr_spill', g('sgpr_spill_count'))
" "$FILT"
rm -rf $T
