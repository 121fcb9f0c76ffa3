"""Code-generation guards (no GPU needed: hipcc cross-compiles gfx950): the hot kernels must not fall back to scratch memory.

Found the hard way this round: kernels that go through the generic `Sample<KD, KS>` keep its dynamically indexed tap arrays
in scratch (672 B per lane, stores + dependent reloads per constraint) -- 53 us instead of ~17 us for the candidate cost.  The
fast-path kernels hold their taps in registers; this test pins that property, the register budget of the blocked block-Jacobi
inverse (two 512-thread workgroups per CU need <= 128 VGPRs) and that it really issues f64 MFMAs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "robust_cvd_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

SOURCE = f'''
#include <hip/hip_runtime.h>
#include "{CSRC}/cvd_device.h"
#include "{CSRC}/cvd_kernels.h"
#include "{CSRC}/cvd_coarse.h"
#include "{CSRC}/cvd_cross.h"
#include "{CSRC}/cvd_dense_walk.h"
namespace cvd {{
template __global__ void k_dense_walk<4>(Layout, Table, DenseWalkList, const double*, const FrameConst*, double*, double*);
template __global__ void k_dense_gg<4>(Layout, Table, CrossPairs, const int*, const double*, int, int, double*);
template __global__ void k_cost_items_fast<4>(Layout, Table, Items, const double*, const FrameConst*, double*);
template __global__ void k_coarse_edges_fast<4>(Layout, Table, Items, const double*, const FrameConst*, const int*, double*, double*);
template __global__ void k_matvec_pairs_fast<4, 128>(Layout, Table, Items, const double*, const FrameConst*, const double*,
                                                      const double*, const double*, const double*, int, double*, CoarseView);
template __global__ void k_matvec_pairs_fast<4, 256, 1>(Layout, Table, Items, const double*, const FrameConst*, const double*,
                                                         const double*, const double*, const double*, int, double*, CoarseView);
template __global__ void k_block_inverse_mfma<8, 10>(Layout, const double*, const double*, float*, int*);
template __global__ void k_cost_items<4, 0>(Layout, Table, Items, const double*, const FrameConst*, double*);
}}
'''


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("codegen")
    src, out = d / "k.hip", d / "k.s"
    src.write_text(SOURCE)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-S", "--cuda-device-only",
                    "-o", str(out), str(src)], check=True, capture_output=True, timeout=600)
    return out.read_text()


def kernel_info(asm, name):
    """(.amdhsa descriptor fields, body text) of the one kernel whose mangled name contains `name`."""
    m = [b for b in re.findall(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", asm, re.S) if name in b[0]]
    assert len(m) == 1, (name, [b[0] for b in m])
    mangled, desc = m[0]
    fields = {k: int(v) for k, v in re.findall(r"\.amdhsa_(\w+) (\d+)", desc)}
    body = asm[asm.index(f"\n{mangled}:"):]
    body = body[:body.index("s_endpgm")]
    return fields, body


@pytest.mark.parametrize("name", ["17k_cost_items_fast", "19k_coarse_edges_fast", "19k_matvec_pairs_fastILi4ELi128ELi0",
                                  "19k_matvec_pairs_fastILi4ELi256ELi1"])
def test_fast_kernels_use_no_scratch(asm, name):
    """No scratch-resident arrays.  The specialised product keeps the next table record in flight (RecordStream: 6 registers)
    inside its 168-register budget and the allocator parks ONE 8-byte value in scratch for it (one reload per trip of 64
    constraints, measured 50.2 -> 48.1 us with it): allowed, nothing beyond."""
    fields, body = kernel_info(asm, name)
    spec = name.endswith("ILi4ELi256ELi1")
    assert fields["private_segment_fixed_size"] <= (16 if spec else 0), fields
    assert body.count("scratch_") <= (4 if spec else 0)


def test_specialised_pairs_product_fits_three_waves_per_simd(asm):
    """The default pipeline's variant of the hot kernel (SPEC = 1: one value parameter, ReproDisparity, Cauchy fixed at compile
    time; the frames' rotations / translations handed to the loop as scalars) must stay within 168 VGPRs = three waves per SIMD
    (the runtime-variant kernel needs ~190: two)."""
    fields, body = kernel_info(asm, "19k_matvec_pairs_fastILi4ELi256ELi1")
    assert fields["next_free_vgpr"] <= 168, fields
    assert "v_readfirstlane_b32" in body


def test_generic_cost_kernel_is_the_one_with_scratch_resident_taps(asm):
    """Documents WHY the fast variants exist (if this ever turns 0 the generic path has been fixed and they can go)."""
    fields, _ = kernel_info(asm, "12k_cost_items")
    assert fields["private_segment_fixed_size"] > 0


def test_block_inverse_runs_on_the_f64_matrix_cores_within_its_register_budget(asm):
    fields, body = kernel_info(asm, "20k_block_inverse_mfma")
    assert fields["next_free_vgpr"] <= 128, fields          # 4 waves per SIMD = two 512-thread workgroups per CU
    assert body.count("v_mfma_f64_16x16x4") >= 8            # -T panel (4) + rank-16 update (4 per tile slot)
    assert "row_newbcast" in body or "row_share" in body    # pivot row by DPP broadcast, not through LDS
    assert fields["private_segment_fixed_size"] <= 512, fields  # a few spilled address temporaries, not the tile registers


def test_update_kernel_register_budget(asm):
    """k_cg_update runs 768-thread workgroups (12 waves) at B = 177 and, with the dense coarse level fused in, F + F / 2 of them:
    they are all resident at once only with TWO workgroups per CU, i.e. 6 waves per SIMD = at most 80 VGPRs.  Measured with
    the f64 inverse, first batch of row loads + a load-use loop over the rest: 12 loads in flight per thread (116 VGPRs)
    26.8 us, 8 (92) 27.2 us, 6 (78) 24.8 us, 4 25.4 us; the whole row in batches: 6 (94 VGPRs) 27.3 us, 4 (78) 24.2 us."""
    fields, body = kernel_info(asm, "11k_cg_update")
    assert fields["next_free_vgpr"] <= 80, fields
    assert fields["private_segment_fixed_size"] == 0, fields


def test_dense_mode_block_kernels(asm):
    """Explicit cross blocks (cvd_cross.h): the streaming product is light (no scratch, <= 64 VGPRs: latency is hidden by
    occupancy)."""
    fields, _ = kernel_info(asm, "14k_cross_matvec")
    assert fields["private_segment_fixed_size"] == 0 and fields["next_free_vgpr"] <= 64, fields


def test_dense_walk_uses_the_matrix_pipe_and_stays_in_registers(asm):
    """The dense mode's one-walk assembly (cvd_dense_walk.h): the pose Gram tile is accumulated by v_mfma_f64_16x16x4 (not by
    per-lane accumulators: 45 of them would not fit), everything with a grid vertex by ds_add_f64; two waves per SIMD (<= 256 VGPRs)
    with at most a handful of spilled values; the grid x grid kernel is small enough for sixteen waves per CU."""
    fields, body = kernel_info(asm, "12k_dense_walk")
    assert "v_mfma_f64_16x16x4" in body and "ds_add_f64" in body
    assert fields["next_free_vgpr"] <= 256, fields
    assert fields["private_segment_fixed_size"] <= 128, fields
    fields, body = kernel_info(asm, "10k_dense_gg")
    assert fields["next_free_vgpr"] <= 128 and fields["private_segment_fixed_size"] == 0, fields
