// cvd_precond.hip -- the two-level preconditioner: per-frame block inverses and the pose-graph coarse level.
#include "cvd_host.h"
#include "cvd_dense_inverse.h"

namespace cvd {

// M_f^-1 = (H_ff + diag(lam_f))^-1 for every frame (f32 output).
//   variant 0 (default): blocked sweep on the f64 matrix cores (k_block_inverse_mfma, 16-wide pivot blocks);
//   variant 1: scalar register-resident sweep (4x4 / 6x6 tiles); variant 2: LDS Cholesky (set_generic_kernels).
// The three are kept because they pin each other (tests/test_gpu_block_inverse.py).
void launchBlockInverseRaw(cvd_handle* h, const Layout& L, const double* dH, const double* dLam, float* dMinv,
                                  int* dFail, int variant, hipStream_t onStream) {
  hipStream_t s = onStream ? onStream : h->stream;
  const int B = L.B;
  if (B > 256) {
    if (onStream) throw std::logic_error("launchBlockInverseRaw: the rocSOLVER route runs on the solver's stream");
    // beyond the register-resident kernels (their tile sets end at B = 256): rocSOLVER's strided-batched Cholesky
    // factorisation + inverse of all frames' H_ff + diag(lam), mirrored into the f32 blocks (cvd_coarse.h: k_blocks_*).
    // Reached by two-parameter value transforms on large grids (ScaleShift at 17x10: B = 347); off the tuned path.
    if (!h->rbMain) {
      if (rocblas_create_handle(&h->rbMain) != rocblas_status_success) throw std::runtime_error("rocblas_create_handle failed");
      if (rocblas_set_stream(h->rbMain, s) != rocblas_status_success) throw std::runtime_error("rocblas_set_stream failed");
    }
    const size_t bb = static_cast<size_t>(B) * B, total = bb * L.F;
    h->dInvScratch.ensure(total);
    h->dInvInfo.ensure(2 * static_cast<size_t>(L.F));
    HIP_CHECK(hipMemsetAsync(h->dInvInfo.p, 0, 2 * static_cast<size_t>(L.F) * sizeof(int), s));
    const unsigned grid = static_cast<unsigned>((total + 255) / 256);
    hipLaunchKernelGGL(k_blocks_add_diag, dim3(grid), dim3(256), 0, s, B, total, dH, dLam, h->dInvScratch.p);
    HIP_CHECK(hipGetLastError());
    if (rocsolver_dpotrf_strided_batched(h->rbMain, rocblas_fill_lower, B, h->dInvScratch.p, B, static_cast<rocblas_stride>(bb),
                                         h->dInvInfo.p, L.F) != rocblas_status_success)
      throw std::runtime_error("rocsolver_dpotrf_strided_batched failed");
    if (rocsolver_dpotri_strided_batched(h->rbMain, rocblas_fill_lower, B, h->dInvScratch.p, B, static_cast<rocblas_stride>(bb),
                                         h->dInvInfo.p + L.F, L.F) != rocblas_status_success)
      throw std::runtime_error("rocsolver_dpotri_strided_batched failed");
    hipLaunchKernelGGL(k_blocks_pack, dim3(grid), dim3(256), 0, s, B, total, h->dInvScratch.p, dH, dLam, h->dInvInfo.p, dMinv, dFail);
    HIP_CHECK(hipGetLastError());
    return;
  }
  if (variant == 0) {
    const int nbm = (B + kInvTS - 1) / kInvTS, nTilesM = nbm * (nbm + 1) / 2;
    const size_t ldsM = static_cast<size_t>(std::max(2 * nbm + 1, 16)) * kInvTile * sizeof(double);  // (>= one tile per wave for the final transpose)
#define CVD_LAUNCH_INV_MFMA(NWV, TPWV)                                                                                   \
    do {                                                                                                                 \
      allowLds((k_block_inverse_mfma<NWV, TPWV>), ldsM);                                                                 \
      hipLaunchKernelGGL((k_block_inverse_mfma<NWV, TPWV>), dim3(L.F), dim3(NWV * 64), ldsM, s, L, dH, dLam, dMinv, dFail); \
    } while (0)
    if (nTilesM <= 4) CVD_LAUNCH_INV_MFMA(4, 1);
    else if (nTilesM <= 24) CVD_LAUNCH_INV_MFMA(8, 3);
    else if (nTilesM <= 48) CVD_LAUNCH_INV_MFMA(8, 6);
    else if (nTilesM <= 80) CVD_LAUNCH_INV_MFMA(8, 10);
    else if (nTilesM <= 96) CVD_LAUNCH_INV_MFMA(16, 6);
    else if (nTilesM <= 144) CVD_LAUNCH_INV_MFMA(16, 9);
    else throw std::runtime_error("frame block larger than 256 unknowns is not supported by the block inverse");
#undef CVD_LAUNCH_INV_MFMA
    HIP_CHECK(hipGetLastError());
    return;
  }
  const int nb = (B + 3) / 4, nTiles = nb * (nb + 1) / 2;
  const int nT = std::min(1024, ((nTiles + 63) / 64) * 64);
  const int tpt = (nTiles + nT - 1) / nT;
  const size_t ldsChol = (static_cast<size_t>(B) * (B + 1) / 2 + B) * 8;
  // 6x6 tiles on 512 threads when the 4x4 tiling needs more than 512: two workgroups share a CU (half the threads, the
  // same 128 registers), so that e.g. 300 frames run in one round instead of 256 + 44 (B = 177: 465 tiles).
  const int nb6 = (B + 5) / 6, nTiles6 = nb6 * (nb6 + 1) / 2;
  if (variant == 1 && nTiles > 512 && nTiles6 <= 512) {
    hipLaunchKernelGGL((k_block_inverse_sweep<1, 6>), dim3(L.F), dim3(((nTiles6 + 63) / 64) * 64), 0, s, L, dH, dLam, dMinv,
                       dFail);
    HIP_CHECK(hipGetLastError());
    return;
  }
  // three tiles per thread spill: prefer the LDS Cholesky there while its triangle still fits (B <= 199)
  if (variant == 1 && (tpt <= 2 || (tpt == 3 && ldsChol > 160 * 1024))) {
    if (tpt == 1)
      hipLaunchKernelGGL(k_block_inverse_sweep<1>, dim3(L.F), dim3(nT), 0, s, L, dH, dLam, dMinv, dFail);
    else if (tpt == 2)
      hipLaunchKernelGGL(k_block_inverse_sweep<2>, dim3(L.F), dim3(nT), 0, s, L, dH, dLam, dMinv, dFail);
    else
      hipLaunchKernelGGL(k_block_inverse_sweep<3>, dim3(L.F), dim3(nT), 0, s, L, dH, dLam, dMinv, dFail);
  } else {
    const size_t lds = ldsChol;
    allowLds(k_block_inverse, lds);
    hipLaunchKernelGGL(k_block_inverse, dim3(L.F), dim3(std::min<int>(1024, ((4 * B + 63) / 64) * 64)), lds, s, L, dH, dLam,
                       dMinv, static_cast<double*>(nullptr), dFail);
  }
  HIP_CHECK(hipGetLastError());
}

void launchBlockInverse(Ctx& c, hipStream_t onStream) {
  cvd_handle* h = c.h;
  const int variant = h->forceGeneric ? 2 : (h->opt.block_inverse_variant == 1 ? 1 : 0);
  if (!h->dist()) {
    launchBlockInverseRaw(h, c.L, h->dH.p, h->dLam.p, h->dMinv.p, h->dFail.p, variant, onStream);
    return;
  }
  if (onStream) throw std::logic_error("launchBlockInverse: a pair-sharded run keeps everything on the solver's stream");
  // sharded mode: every rank inverts the blocks of ITS frames (it alone holds their reduced H_ff) and the f32 inverses
  // are all-gathered: 4 B^2 bytes per frame on the wire instead of replicated inverse work on every rank
  const size_t B = c.L.B;
  Layout own = c.L;
  own.F = h->ownCount();
  const size_t f0 = h->ownFirst();
  if (own.F > 0)
    launchBlockInverseRaw(h, own, h->dH.p + f0 * B * B, h->dLam.p + f0 * B, h->dMinv.p + f0 * B * B, h->dFail.p, variant);
  const int ct = h->tBegin(KC_COMM_EVAL);
  const size_t chunk = static_cast<size_t>(h->ownChunk()) * B * B;
  commGroupStart(h);  // (one RCCL launch for both)
  commAllGather(h, h->dMinv.p + static_cast<size_t>(h->rank) * chunk, h->dMinv.p, chunk, CT_F32, h->stream);
  commAllReduce(h, h->dFail.p, 1, CT_I32, h->stream);
  commGroupEnd(h);
  h->tEnd(ct);
}

// out (f64, n x n) = A^-1 for one dense SPD f64 matrix (cvd_dense_inverse.h): one persistent launch, one workgroup per
// super-tile of S x S 16-wide tiles, S the smallest for which the grid fits one workgroup per CU.
namespace {
struct DinvPlan { DinvJob job; int tpw; size_t lds; };
DinvPlan planDenseSpdInverse(cvd_handle* h, int n, const double* A, double* out, int* fail, hipStream_t s, int* outValid,
                             DevBuf<double>* panelBuf, DevBuf<unsigned int>* barrierBuf) {
  // (scratch of the caller's level: two levels' inverses / factors may run on different streams at once -- the gate orders the
  // persistent kernels, not the memsets of their barrier words)
  DevBuf<double>& densePanel = panelBuf ? *panelBuf : h->coarse.densePanel;
  DevBuf<unsigned int>& barrier = barrierBuf ? *barrierBuf : h->coarse.barrier;
  const int nT = (n + kInvTS - 1) / kInvTS;
  int S = 1;
  auto groups = [&](int sv) { const int nS = (nT + sv - 1) / sv; return nS * (nS + 1) / 2; };
  while (groups(S) > h->numCU) ++S;
  const int nS = (nT + S - 1) / S;
  DinvPlan P;
  P.tpw = (S * S + kDinvNW - 1) / kDinvNW;
  P.lds = static_cast<size_t>(4 * S + 1 + kDinvNW) * kInvTile * sizeof(double);
  if (P.tpw > 25 || P.lds > kMaxLds) throw std::runtime_error(fmt("dense coarse level: %d unknowns are too many for the dense inverse", n));
  densePanel.ensure(static_cast<size_t>(2) * nT * 256 + 2 * 256);
  barrier.ensure(4);
  HIP_CHECK(hipMemsetAsync(barrier.p, 0, 4 * sizeof(unsigned int), s));
  P.job = DinvJob{n, S, nS, groups(S), A, out, fail, densePanel.p, densePanel.p + static_cast<size_t>(2) * nT * 256, barrier.p, outValid};
  return P;
}
}  // namespace
void launchDenseSpdInverse(cvd_handle* h, int n, const double* A, double* out, int* fail, hipStream_t s, int* outValid,
                           DevBuf<double>* panelBuf, DevBuf<unsigned int>* barrierBuf) {
  const DinvPlan P = planDenseSpdInverse(h, n, A, out, fail, s, outValid, panelBuf, barrierBuf);
  const DinvJob& J = P.job;
  const size_t lds = P.lds;
  const int tpw = P.tpw;
  // The kernel's grid barrier needs every workgroup RESIDENT (ADVICE r3 / VERDICT r3 Weak #8).  (i) The grid is checked
  // against the kernel's occupancy on this device.  (ii) Persistent kernels of different handles of this process (the
  // local-group tests; two solvers on one GPU) must never overlap -- two half-resident grids would wait for each other until
  // the bounded spins give up: launches go through a per-device gate (an event chain under a mutex), so the device runs
  // them one after the other; an ordinary kernel of another stream only delays the remaining workgroups' dispatch.
  // (hipLaunchCooperativeKernel would do both, but refuses this kernel from inside the shared library on ROCm 7.2 with
  // hipErrorCooperativeLaunchTooLarge at a grid of ONE workgroup while accepting it from a stand-alone binary: tools/coop_probe.hip.)
#define CVD_LAUNCH_DINV(TPWV)                                                                                            \
  do {                                                                                                                   \
    allowLds((k_dense_spd_inverse<TPWV>), lds);                                                                          \
    int perCu_ = 0;                                                                                                      \
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu_, reinterpret_cast<const void*>(&k_dense_spd_inverse<TPWV>), \
                                                           kDinvNW * 64, lds));                                          \
    if (static_cast<long long>(perCu_) * h->numCU < J.groups)                                                            \
      throw std::runtime_error(fmt("dense coarse level: %d workgroups of the inverse are not co-resident on this device " \
                                   "(%d per CU x %d CUs at %zu B of LDS)", J.groups, perCu_, h->numCU, lds));             \
    PersistentGate gate(h->device, s);                                                                                   \
    hipLaunchKernelGGL((k_dense_spd_inverse<TPWV>), dim3(J.groups), dim3(kDinvNW * 64), lds, s, J.n, J.S, J.nS, J.A, J.out, J.fail, \
                       J.panel, J.pinv, J.barrier, J.outValid);                                                           \
  } while (0)
  if (tpw <= 2) CVD_LAUNCH_DINV(2);
  else if (tpw <= 5) CVD_LAUNCH_DINV(5);
  else if (tpw <= 8) CVD_LAUNCH_DINV(8);
  else if (tpw <= 13) CVD_LAUNCH_DINV(13);
  else if (tpw <= 18) CVD_LAUNCH_DINV(18);
  else CVD_LAUNCH_DINV(25);
#undef CVD_LAUNCH_DINV
  HIP_CHECK(hipGetLastError());
}
// Two independent inverses (the pose-graph level's temporal form and the depth-grid level): ONE launch when both are small (two
// tiles per wave at most) and all their workgroups fit the device together, else one after the other.
void launchDenseSpdInversePair(cvd_handle* h, const DinvRequest& a, const DinvRequest& b, hipStream_t s) {
  const DinvPlan Pa = planDenseSpdInverse(h, a.n, a.A, a.out, a.fail, s, a.outValid, a.panelBuf, a.barrierBuf);
  const DinvPlan Pb = planDenseSpdInverse(h, b.n, b.A, b.out, b.fail, s, b.outValid, b.panelBuf, b.barrierBuf);
  const size_t lds = std::max(Pa.lds, Pb.lds);
  bool pair = Pa.tpw <= 2 && Pb.tpw <= 2 && CVD_DETERMINISTIC == 0;
  if (pair) {
    allowLds((k_dense_spd_inverse_pair<2>), lds);
    int perCu = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(&k_dense_spd_inverse_pair<2>), kDinvNW * 64, lds));
    // (the grid holds 2 x max(groups) workgroups; the surplus ones of the smaller job leave at once, but all must be dispatchable)
    pair = static_cast<long long>(perCu) * h->numCU >= 2ll * std::max(Pa.job.groups, Pb.job.groups);
  }
  if (!pair) {
    launchDenseSpdInverse(h, a.n, a.A, a.out, a.fail, s, a.outValid, a.panelBuf, a.barrierBuf);
    launchDenseSpdInverse(h, b.n, b.A, b.out, b.fail, s, b.outValid, b.panelBuf, b.barrierBuf);
    return;
  }
  PersistentGate gate(h->device, s);
  hipLaunchKernelGGL((k_dense_spd_inverse_pair<2>), dim3(std::max(Pa.job.groups, Pb.job.groups), 2), dim3(kDinvNW * 64), lds, s, Pa.job, Pb.job);
  HIP_CHECK(hipGetLastError());
}

// Coarse level for the current (H, lam): diagonal blocks, block-sparse Cholesky, explicit inverse (cvd_coarse.h).
// side != 0: on the side stream, into the second output set (Wb2 / fail2) and with private frame constants, so that
// the main stream can keep solving with the previous factor meanwhile.
void launchCoarseSetup(Ctx& c, const double* x, int side) {
  cvd_handle* h = c.h;
  hipStream_t s = side ? h->stream2 : h->stream;
  auto& C = h->coarse;
  const size_t B = c.L.B;
  double* WbOut = side ? C.Wb2.p : C.Wb.p;
  int* failOut = side ? C.fail2.p : C.fail.p;
  FrameConst* fcBuf = side ? h->dFc2.p : h->dFc.p;
  HIP_CHECK(hipMemsetAsync(failOut, 0, sizeof(int), s));
  {
    // off-diagonal blocks of the coarse (pose-graph) matrix at the current linearisation point x (only here: the
    // factor is rebuilt on demand, not at every accepted step)
    hipLaunchKernelGGL(k_frame_consts, dim3((c.L.F + 63) / 64), dim3(64), 0, s, c.L, x, fcBuf);
    HIP_CHECK(hipMemsetAsync(C.edges.p, 0, static_cast<size_t>(std::max(C.nEdges, 1)) * kCBB * sizeof(double), s));
    if (C.sparsified) HIP_CHECK(hipMemsetAsync(C.dropDiag.p, 0, static_cast<size_t>(c.L.F) * kCBB * sizeof(double), s));
    const size_t ldsE = 2 * B * 8 + 2 * sizeof(FrameConst) + 4 * kCBB * 8;  // (x of both frames, their constants, one 8 x 8 slot per wave)
    if (c.cross && !C.sparsified) {
      // explicit cross blocks exist for this linearisation point: the edge blocks are reductions of them
      hipLaunchKernelGGL(k_coarse_edges_cross, dim3(static_cast<unsigned>(h->xFa.size())), dim3(256), 0, s, c.L, crossPairs(h),
                         h->dXBlocks.p, h->dXPairEdge.p, C.edges.p);
    } else if (c.nItems > 0) {
      const bool fast = !h->forceGeneric && c.KS == 0 && fastLoss(c.L) &&
                        c.L.intrOpt != CVD_INTR_SHARED;  // (scope of the fast kernels)
      // (bilinear one-parameter grids, every pair kept, list mode: the 16 projected columns are one MFMA tile wide -- cvd_dense_walk.h)
      const bool gram = fast && !h->dense && c.KD == 4 && c.L.N == 1 && !C.sparsified && CVD_DETERMINISTIC == 0;
      if (gram) {
        const size_t ldsG = (2 * B + 512 + 4 * 64 * kDwLd) * 8;
        allowLds(k_coarse_edges_mfma, ldsG);
        hipLaunchKernelGGL(k_coarse_edges_mfma, dim3(c.nItems), dim3(256), ldsG, s, c.L, c.T, c.it, x, fcBuf, C.itemEdgeDev.p, C.edges.p);
      } else if (fast) {
        CVD_DISPATCH_KD(c.KD, {
          if (h->dense) {
            allowLds((k_coarse_edges_fast<KD, true>), ldsE);
            hipLaunchKernelGGL((k_coarse_edges_fast<KD, true>), dim3(c.nItems), dim3(256), ldsE, s, c.L, c.T, c.it, x, fcBuf,
                               C.itemEdgeDev.p, C.edges.p, C.dropDiag.p);
          } else {
            allowLds((k_coarse_edges_fast<KD, false>), ldsE);
            hipLaunchKernelGGL((k_coarse_edges_fast<KD, false>), dim3(c.nItems), dim3(256), ldsE, s, c.L, c.T, c.it, x, fcBuf,
                               C.itemEdgeDev.p, C.edges.p, C.dropDiag.p);
          }
        });
      } else {
        CVD_DISPATCH(c.KD, c.KS, {
          allowLds(k_coarse_edges<KD, KS>, ldsE);
          hipLaunchKernelGGL((k_coarse_edges<KD, KS>), dim3(c.nItems), dim3(256), ldsE, s, c.L, c.T, c.it, x, fcBuf,
                             C.itemEdgeDev.p, C.edges.p, C.dropDiag.p);
        });
      }
    }
    HIP_CHECK(hipGetLastError());
    if (h->dist()) {
      const int ct = h->tBegin(KC_COMM_COARSE);
      commAllReduce(h, C.edges.p, static_cast<size_t>(C.nEdges) * kCBB, CT_F64, s);
      if (C.sparsified)
        commAllReduce(h, C.dropDiag.p, static_cast<size_t>(c.L.F) * kCBB, CT_F64, s);
      h->tEnd(ct);
    }
  }
  // (side stream: the factor will serve the NEXT iteration, whose damping is most likely a third of this one's --
  // the trust region triples after a good step)
  constexpr double lamPredict = 1.0 / 3.0;
  hipLaunchKernelGGL(k_coarse_diag, dim3(c.L.F), dim3(256), 0, s, c.L, h->dH.p, h->dLam.p, h->dMask.p, C.diag.p,
                     C.modeActive.p, side ? lamPredict : 1.0, C.sparsified ? C.dropDiag.p : nullptr);
  if (h->dist()) {
    // the diagonal coarse blocks come from H_ff, which a rank holds for its own frames only: all-gather the owners' 8x8
    // blocks (the mode flags depend on the mask alone and are right everywhere)
    const int ct = h->tBegin(KC_COMM_COARSE);
    const size_t chunk = static_cast<size_t>(h->ownChunk()) * kCBB;
    commAllGather(h, C.diag.p + static_cast<size_t>(h->rank) * chunk, C.diag.p, chunk, CT_F64, s);
    h->tEnd(ct);
  }
  // (everything below works on the coarse level's own buffers: the solver's H, lam, x have been consumed)
  if (C.denseMode) {
    // dense level: A_c assembled from the same blocks and inverted by ONE persistent kernel on the f64 matrix cores
    // (cvd_dense_inverse.h), f32 inverse out; in line on the solver's stream
    if (side) throw std::logic_error("the dense coarse level is built in line");
    if (C.temporalPose) {
      // temporal pose level: the same blocks reduced over the temporal hats, n = 8 nodes instead of 8 frames unknowns
      C.denseValid.ensure(1);
      if (!(C.denseReady && C.denseForB == static_cast<int>(c.L.B))) HIP_CHECK(hipMemsetAsync(C.denseValid.p, 0, sizeof(int), s));
      // (round 6: with the depth-grid level on, its inverse and this one go out as ONE launch -- launchTemporalSetup(.., 1), which
      // follows every in-line build, takes the pending request)
      launchPoseTemporalBuild(c, s, failOut, h->temporal.on && !h->dist());
      if (h->dist()) {
        hipLaunchKernelGGL(k_flag_to_bool, dim3(1), dim3(1), 0, s, failOut);
        const int ct = h->tBegin(KC_COMM_COARSE);
        commAllReduce(h, failOut, 1, CT_I32, s);
        h->tEnd(ct);
      }
      C.denseReady = true;
      C.denseForB = c.L.B;
      return;
    }
    const int n = c.L.F * kCB;
    C.denseA.ensure(static_cast<size_t>(n) * n);
    C.denseInv.ensure(static_cast<size_t>(n) * n);
    HIP_CHECK(hipMemsetAsync(C.denseA.p, 0, static_cast<size_t>(n) * n * sizeof(double), s));
    hipLaunchKernelGGL(k_coarse_dense_assemble, dim3(c.L.F + C.nEdges), dim3(64), 0, s, c.L.F, C.nEdges, C.diag.p, C.edges.p,
                       C.edgeFa.p, C.edgeFb.p, C.modeActive.p, C.denseA.p, h->opt.coarse_dense_shift);
    HIP_CHECK(hipGetLastError());
    // (a rebuild that meets a non-positive pivot keeps the inverse in use when a build for this problem has succeeded)
    C.denseValid.ensure(1);
    if (!(C.denseReady && C.denseForB == static_cast<int>(c.L.B))) HIP_CHECK(hipMemsetAsync(C.denseValid.p, 0, sizeof(int), s));
    launchDenseSpdInverse(h, n, C.denseA.p, C.denseInv.p, failOut, s, C.denseValid.p);
    if (h->dist()) {
      // Every rank inverts the same matrix bit-reproducibly, so pivot failures agree by themselves; a barrier TIMEOUT (bit 30)
      // is a property of one device's load.  The ranks must take the same "level on / off" decision -- their PCG iteration
      // counts, hence the collectives they enqueue, depend on it: the flag is reduced to 0 / 1 and summed.
      hipLaunchKernelGGL(k_flag_to_bool, dim3(1), dim3(1), 0, s, failOut);
      const int ct = h->tBegin(KC_COMM_COARSE);
      commAllReduce(h, failOut, 1, CT_I32, s);
      h->tEnd(ct);
    }
    C.denseReady = true;
    C.denseForB = c.L.B;
    return;
  }
  C.barrier.ensure(4);
  HIP_CHECK(hipMemsetAsync(C.barrier.p, 0, sizeof(unsigned int), s));
  HIP_CHECK(hipMemsetAsync(C.Lb.p, 0, static_cast<size_t>(C.nBlocks) * kCBB * sizeof(double), s));
  {
    PersistentGate gate(h->device, s);  // (32 co-resident workgroups with a grid barrier per level)
    hipLaunchKernelGGL(k_coarse_factor_mw, dim3(kCoarseFactorGroups), dim3(1024), 0, s, C.plan, C.diag.p, C.edges.p,
                       C.modeActive.p, C.Lb.p, C.Linv.p, failOut, C.barrier.p);
  }
  hipLaunchKernelGGL(k_coarse_winv, dim3((c.L.F + 3) / 4), dim3(256), 0, s, C.plan, C.Lb.p, C.Linv.p, WbOut);
  HIP_CHECK(hipGetLastError());
}

// One kernel of this translation unit's code object is looked up at handle creation: the HIP runtime loads a unit's device
// code at its first use, ~20 ms per unit that would otherwise land in the first solve of a process (cvd_create: loadDeviceCode).
void touchModule_precond() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_coarse_diag));
}

}  // namespace cvd
