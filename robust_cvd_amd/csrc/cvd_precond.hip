// cvd_precond.hip -- the two-level preconditioner: per-frame block inverses and the pose-graph coarse level.
#include "cvd_host.h"

namespace cvd {

// M_f^-1 = (H_ff + diag(lam_f))^-1 for every frame (f32 output).
//   variant 0 (default): blocked sweep on the f64 matrix cores (k_block_inverse_mfma, 16-wide pivot blocks);
//   variant 1: scalar register-resident sweep (4x4 / 6x6 tiles); variant 2: LDS Cholesky (set_generic_kernels).
// The three are kept because they pin each other (tests/test_gpu_block_inverse.py).
void launchBlockInverseRaw(cvd_handle* h, const Layout& L, const double* dH, const double* dLam, float* dMinv,
                                  int* dFail, int variant) {
  hipStream_t s = h->stream;
  const int B = L.B;
  if (B > 256) {
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
  static const bool noTs6 = std::getenv("CVD_BLOCK_INVERSE_TS4") != nullptr;  // development knob
  if (variant == 1 && !noTs6 && nTiles > 512 && nTiles6 <= 512) {
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

void launchBlockInverse(Ctx& c) {
  cvd_handle* h = c.h;
  static const bool scalarSweep = std::getenv("CVD_BLOCK_INVERSE_SWEEP") != nullptr;  // comparison: the scalar sweep
  const int variant = h->forceGeneric ? 2 : (scalarSweep ? 1 : 0);
  if (!h->dist()) {
    launchBlockInverseRaw(h, c.L, h->dH.p, h->dLam.p, h->dMinv.p, h->dFail.p, variant);
    return;
  }
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
  NCCL_CHECK(ncclAllGather(h->dMinv.p + static_cast<size_t>(h->rank) * chunk, h->dMinv.p, chunk, ncclFloat, h->comm, h->stream));
  NCCL_CHECK(ncclAllReduce(h->dFail.p, h->dFail.p, 1, ncclInt, ncclSum, h->comm, h->stream));
  h->tEnd(ct);
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
    const size_t ldsE = 2 * B * 8 + 2 * sizeof(FrameConst) + kCBB * 8;
    static const bool crossEdgesOff = std::getenv("CVD_COARSE_EDGES_MATRIX_FREE") != nullptr;  // comparison knob
    if (c.cross && !C.sparsified && !crossEdgesOff) {
      // explicit cross blocks exist for this linearisation point: the edge blocks are reductions of them
      hipLaunchKernelGGL(k_coarse_edges_cross, dim3(static_cast<unsigned>(h->xFa.size())), dim3(256), 0, s, c.L, crossPairs(h),
                         h->dXBlocks.p, h->dXPairEdge.p, C.edges.p);
    } else if (c.nItems > 0) {
      static const bool genericEdges = std::getenv("CVD_COARSE_EDGES_GENERIC") != nullptr;  // comparison knob
      const bool fast = !h->forceGeneric && !genericEdges && c.KS == 0 && fastLoss(c.L) &&
                        c.L.intrOpt != CVD_INTR_SHARED;  // (scope of the fast kernels)
      if (fast) {
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
      NCCL_CHECK(ncclAllReduce(C.edges.p, C.edges.p, static_cast<size_t>(C.nEdges) * kCBB, ncclDouble, ncclSum, h->comm, s));
      if (C.sparsified)
        NCCL_CHECK(ncclAllReduce(C.dropDiag.p, C.dropDiag.p, static_cast<size_t>(c.L.F) * kCBB, ncclDouble, ncclSum, h->comm, s));
      h->tEnd(ct);
    }
  }
  // (side stream: the factor will serve the NEXT iteration, whose damping is most likely a third of this one's --
  // the trust region triples after a good step)
  static const double lamPredict = []() { const char* e = std::getenv("CVD_COARSE_LAM_PREDICT"); return e ? std::atof(e) : 1.0 / 3.0; }();
  hipLaunchKernelGGL(k_coarse_diag, dim3(c.L.F), dim3(256), 0, s, c.L, h->dH.p, h->dLam.p, h->dMask.p, C.diag.p,
                     C.modeActive.p, side ? lamPredict : 1.0, C.sparsified ? C.dropDiag.p : nullptr);
  if (h->dist()) {
    // the diagonal coarse blocks come from H_ff, which a rank holds for its own frames only: all-gather the owners' 8x8
    // blocks (the mode flags depend on the mask alone and are right everywhere)
    const int ct = h->tBegin(KC_COMM_COARSE);
    const size_t chunk = static_cast<size_t>(h->ownChunk()) * kCBB;
    NCCL_CHECK(ncclAllGather(C.diag.p + static_cast<size_t>(h->rank) * chunk, C.diag.p, chunk, ncclDouble, h->comm, s));
    h->tEnd(ct);
  }
  // (everything below works on the coarse level's own buffers: the solver's H, lam, x have been consumed)
  if (side) HIP_CHECK(hipEventRecord(h->evCoarseRead, s));
  if (C.denseMode) {
    const int n = c.L.F * kCB;
    C.denseA.ensure(static_cast<size_t>(n) * n);
    C.denseInv.ensure(static_cast<size_t>(n) * n);
    C.denseInv2.ensure(static_cast<size_t>(n) * n);
    C.denseInfo.ensure(2);
    const int F = c.L.F, nEdges = C.nEdges;
    // (everything the job needs by value: it may still be enqueuing while the caller's frame moves on)
    auto job = [h, s, side, n, F, nEdges, failOut]() {
      auto& C = h->coarse;
      HIP_CHECK(hipSetDevice(h->device));
      if (!C.rb[side]) {
        if (rocblas_create_handle(&C.rb[side]) != rocblas_status_success) throw std::runtime_error("rocblas_create_handle failed");
        if (rocblas_set_stream(C.rb[side], s) != rocblas_status_success) throw std::runtime_error("rocblas_set_stream failed");
      }
      // memsets + assembly + potrf + potri: ~250 small launches, ~2.3 ms of host time when issued one by one.  Beside the
      // solver (side stream) the sequence is captured ONCE into a hipGraph and replayed with a single launch; the graph is
      // keyed on every pointer / size baked into its nodes.  A capture that rocSOLVER does not support falls back to direct
      // calls for good (state -1).
      auto direct = [&](hipStream_t st) {
        HIP_CHECK(hipMemsetAsync(C.denseA.p, 0, static_cast<size_t>(n) * n * sizeof(double), st));
        HIP_CHECK(hipMemsetAsync(C.denseInfo.p, 0, 2 * sizeof(int), st));
        hipLaunchKernelGGL(k_coarse_dense_assemble, dim3(F + nEdges), dim3(64), 0, st, F, nEdges, C.diag.p, C.edges.p,
                           C.edgeFa.p, C.edgeFb.p, C.modeActive.p, C.denseA.p);
        HIP_CHECK(hipGetLastError());
        // A_c = L L^T, A_c^-1 (rocSOLVER; symmetric input, so the row-major array serves as its own column-major view)
        if (rocsolver_dpotrf(C.rb[side], rocblas_fill_lower, n, C.denseA.p, n, C.denseInfo.p) != rocblas_status_success)
          throw std::runtime_error("rocsolver_dpotrf failed");
        if (rocsolver_dpotri(C.rb[side], rocblas_fill_lower, n, C.denseA.p, n, C.denseInfo.p + 1) != rocblas_status_success)
          throw std::runtime_error("rocsolver_dpotri failed");
      };
      static const bool graphOff = std::getenv("CVD_COARSE_NO_GRAPH") != nullptr;  // comparison knob
      const std::array<const void*, 8> key{C.denseA.p, C.denseInfo.p, C.diag.p, C.edges.p, C.edgeFa.p, C.modeActive.p,
                                           reinterpret_cast<const void*>(static_cast<size_t>(n)),
                                           reinterpret_cast<const void*>(static_cast<size_t>(nEdges))};
      if (!side || graphOff || C.denseGraphState < 0) {
        direct(s);
      } else if (C.denseGraphState == 0) {
        direct(s);  // (first call on this handle: rocBLAS sizes its workspace, loads its kernels -- not capturable)
        C.denseGraphState = 1;
      } else {
        if (C.denseGraph != nullptr && C.denseGraphKey != key) {
          (void)hipGraphExecDestroy(C.denseGraph);
          C.denseGraph = nullptr;
        }
        if (C.denseGraph == nullptr) {
          // Captured on a PRIVATE stream that nothing else ever touches: while the side stream itself were capturing, the
          // main thread's waits on events recorded there (evCoarseRead, evCoarseDone) would be capture-isolation errors --
          // it reaches them during the capture whenever the PCG beside it is short (eta = 0.1: 15 iterations).
          if (!h->streamCapture) HIP_CHECK(hipStreamCreateWithFlags(&h->streamCapture, hipStreamNonBlocking));
          hipStream_t sc = h->streamCapture;
          hipGraph_t g = nullptr;
          bool ok = rocblas_set_stream(C.rb[side], sc) == rocblas_status_success &&
                    hipStreamBeginCapture(sc, hipStreamCaptureModeThreadLocal) == hipSuccess;
          if (ok) {
            try { direct(sc); } catch (...) { ok = false; }
            if (hipStreamEndCapture(sc, &g) != hipSuccess || g == nullptr) ok = false;
          }
          if (rocblas_set_stream(C.rb[side], s) != rocblas_status_success) throw std::runtime_error("rocblas_set_stream failed");
          if (ok && hipGraphInstantiate(&C.denseGraph, g, nullptr, nullptr, 0) != hipSuccess) {
            ok = false;
            C.denseGraph = nullptr;
          }
          if (g != nullptr) (void)hipGraphDestroy(g);
          (void)hipGetLastError();
          if (!ok) {
            C.denseGraphState = -1;
            C.denseGraph = nullptr;
          } else {
            C.denseGraphKey = key;
          }
        }
        if (C.denseGraph != nullptr) HIP_CHECK(hipGraphLaunch(C.denseGraph, s));
        else direct(s);
      }
      hipLaunchKernelGGL(k_coarse_dense_pack, dim3(static_cast<unsigned>((static_cast<size_t>(n) * n + 255) / 256)), dim3(256), 0, s, n,
                         C.denseA.p, C.denseInfo.p, side ? C.denseInv2.p : C.denseInv.p, failOut,
                         side ? C.denseInv.p : nullptr);
      HIP_CHECK(hipGetLastError());
      if (side) HIP_CHECK(hipEventRecord(h->evCoarseDone, s));
    };
    static const bool noWorker = std::getenv("CVD_COARSE_NO_WORKER") != nullptr;  // comparison knob
    if (side && !noWorker) {
      h->sideWorker.submit(job);  // ~250 launches: enqueued by the helper thread while this one enqueues the PCG
    } else {
      h->sideWorker.wait();
      job();
    }
    return;
  }
  static const bool singleWg = std::getenv("CVD_COARSE_FACTOR_1WG") != nullptr;  // comparison / fallback
  if (singleWg) {
    hipLaunchKernelGGL(k_coarse_factor, dim3(1), dim3(1024), 0, s, C.plan, C.diag.p, C.edges.p, C.modeActive.p, C.Lb.p,
                       C.Linv.p, failOut);
  } else {
    C.barrier.ensure(1);
    HIP_CHECK(hipMemsetAsync(C.barrier.p, 0, sizeof(unsigned int), s));
    HIP_CHECK(hipMemsetAsync(C.Lb.p, 0, static_cast<size_t>(C.nBlocks) * kCBB * sizeof(double), s));
    hipLaunchKernelGGL(k_coarse_factor_mw, dim3(kCoarseFactorGroups), dim3(1024), 0, s, C.plan, C.diag.p, C.edges.p,
                       C.modeActive.p, C.Lb.p, C.Linv.p, failOut, C.barrier.p);
  }
  hipLaunchKernelGGL(k_coarse_winv, dim3((c.L.F + 3) / 4), dim3(256), 0, s, C.plan, C.Lb.p, C.Linv.p, WbOut);
  HIP_CHECK(hipGetLastError());
}

}  // namespace cvd
