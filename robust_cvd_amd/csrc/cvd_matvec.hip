// cvd_matvec.hip -- the PCG product q = (J^T J + diag(lam)) p: pair-major partial products, per-frame finish, exchange.
#include "cvd_host.h"

namespace cvd {

bool coarseFusedConsumers() {
  return false;  // (c_f formed inside the consumers: measured slower than the k_coarse_apply_wt launch; kept as a code path of CoarseView)
}
// Pair-sharded mode: the product's exchange carries [q | Z^T q | p.q] in one all-reduce unless the SPARSE coarse level is on
// (its column products need the reduced Z^T q: k_dot_pq).  Layout: [q (F B) | Z^T q (8 F, dense coarse level only) | p.q].
bool fusedExchange(cvd_handle* h, bool withCoarse) { return !withCoarse || h->coarse.denseMode; }
// (sharded: the q part is padded to world x chunk frames, so that its reduce-scatter / all-gather chunks are equal)
size_t exchangeOffsetQc(const Ctx& c) { return c.h->dist() ? static_cast<size_t>(c.h->framesPadded()) * c.L.B : c.n; }
size_t exchangeOffsetPq(const Ctx& c, bool withDenseCoarse) {
  return exchangeOffsetQc(c) + (withDenseCoarse ? static_cast<size_t>(c.L.F) * kCB : 0);
}
// Owner-sharded PCG iteration: with the fused exchange the product's q is REDUCE-SCATTERED to
// the frames' owners ([Z^T q | p.q] all-reduced beside it, one RCCL group), every rank updates x, r, z (and its rows of the dense
// coarse level) for ITS frames only, and z / c / the r^T z shares are all-gathered (one group): two collectives per iteration,
// the per-frame update work divided by the number of ranks.
size_t exchangeTemporalCount(cvd_handle* h, bool withCoarse) {
  const TlStep ts = temporalStep(h);
  return (ts.Ainv != nullptr && h->dist() && fusedExchange(h, withCoarse)) ? static_cast<size_t>(h->F) * ts.S : 0;
}
bool ownerShardedUpdate(cvd_handle* h, bool withCoarse) {
  return h->dist() && fusedExchange(h, withCoarse) && h->opt.dist_owner_update != 0;
}
CoarseView coarseView(cvd_handle* h, bool on, bool walk) {
  auto& C = h->coarse;
  CoarseView v = on ? CoarseView{C.pos.p, C.wPtr.p, C.wRow.p, walk ? C.Wb.p : nullptr, C.y.p, C.modeActive.p, C.fail.p, C.c.p}
                    : CoarseView{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (h->temporal.on && h->temporal.built) {  // third level: its per-frame coefficients and the vertex table of the prolongation
    v.tl = h->temporal.tl.p;
    v.tlW = h->temporal.vW.p;
    v.tlIdx = h->temporal.vIdx.p;
    v.tlS = h->temporal.S;
  }
  return v;
}

// Fills the regulariser Jacobian cache for the products at linearisation point x (before runPcg / the J^T J hook).
void prepareMatvec(Ctx& c, const double* x) {
  cvd_handle* h = c.h;
  const Layout& L = c.L;
  // the per-frame constants must be those of x: a rejected LM step leaves the candidate's behind (evalCost)
  launchFrameConsts(c, x);
  int nr = 0;
  if (L.scaleRegSqrt > 0.0) nr += L.sregX * L.sregY;
  if (L.focalRegSqrt > 0.0) nr += 1;
  if (L.depthDeformW > 0.0 && L.depthType == CVD_DEPTH_GRID) nr += gridNumEdges(L.gx, L.gy, L.gz) * L.N;
  if (L.spatialDeformW > 0.0) nr += L.nS;
  const int stride = std::max(2, c.KD * std::max(1, L.N));
  const size_t entries = static_cast<size_t>(L.F) * stride * std::max(nr, 1);
  h->dRegJac.ensure(entries);
  h->dRegCol.ensure(entries);
  h->dRegCnt.ensure(static_cast<size_t>(L.F) * std::max(nr, 1));
  h->regCache = RegCache{h->dRegJac.p, h->dRegCol.p, h->dRegCnt.p, nr, stride};
  HIP_CHECK(hipMemsetAsync(h->dScal.p + S_DONE, 0, sizeof(double), h->stream));
  if (nr == 0) return;
  CVD_DISPATCH_KD(c.KD, {
    hipLaunchKernelGGL((k_reg_cache<KD>), dim3(L.F), dim3(256), static_cast<size_t>(L.B) * 8, h->stream, L, x, h->dMedian.p, h->dRegOwner.p,
                       h->regCache);
  });
  HIP_CHECK(hipGetLastError());
}

void launchMatvec(Ctx& c, const double* x, const double* z, const double* pOld, double* pNew, int useBeta,
                         const double* lam, double* q, bool withCoarse, bool tailFused) {
  cvd_handle* h = c.h;
  hipStream_t s = h->stream;
  const CoarseView cF = coarseView(h, withCoarse, coarseFusedConsumers());  // z + Z c: the coarse part of the preconditioned residual
  const size_t B = c.L.B;
  if (c.cross) {
    // explicit cross blocks (dense mode): one workgroup per undirected pair streams its B x B block
    hipEvent_t evStart, evStop;
    (void)h->tReserve(KC_MATVEC_PAIRS, evStart, evStop);
    const size_t ldsX = (3 * B + (kCrossThreads / 64) * B + 2 * kCB + 2 * kTlMaxS) * 8;
    // (+ one workgroup per frame for its own block H_ff: see k_cross_matvec)
    const unsigned nP = static_cast<unsigned>(h->xFa.size()) + (h->xDiagRows ? static_cast<unsigned>(c.L.F) : 0u);
    const int* diagSlot = h->xDiagRows ? h->dXDiagSlot.p : nullptr;
    if (evStart)
      hipExtLaunchKernelGGL(k_cross_matvec, dim3(nP), dim3(kCrossThreads), ldsX, s, evStart, evStop, 0, c.L, crossPairs(h),
                            h->dXBlocks.p, h->dH.p, diagSlot, h->dMask.p, z, pOld, h->dScal.p, useBeta, h->dQPart.p, cF);
    else
      hipLaunchKernelGGL(k_cross_matvec, dim3(nP), dim3(kCrossThreads), ldsX, s, c.L, crossPairs(h), h->dXBlocks.p, h->dH.p, diagSlot,
                         h->dMask.p, z, pOld, h->dScal.p, useBeta, h->dQPart.p, cF);
    HIP_CHECK(hipGetLastError());
  } else if (c.L.includeStatic && c.nItems > 0) {
    const size_t lds = 6 * B * 8 + 2 * sizeof(FrameConst) + (18 + 4 * 24 + 8 + 2 * kCB) * 8;
    const size_t ldsFast = 6 * B * 8 + 2 * sizeof(FrameConst) + (32 + static_cast<size_t>(kRedVals) * kRedStride) * 8;  // (the two FrameConst: CVD_MV_SLOAD = 0 only)
    hipEvent_t evStart, evStop;
    (void)h->tReserve(KC_MATVEC_PAIRS, evStart, evStop);
    const bool fast = !h->forceGeneric && c.KS == 0 && fastLoss(c.L);
    // (plain launches unless the launch is timed: hipExtLaunchKernelGGL is not used inside a graph capture)
    const FrameConst* fcp = h->dFc.p;
    const double* maskp = h->dMask.p;
    const double* scalp = h->dScal.p;
    if (fast) {
      // Workgroup size: a work item keeps its slot for ~20 us at 256 threads and the register budget allows two
      // waves per SIMD, i.e. 2 x CUs slots of 256 threads or 4 x CUs slots of 128.  When the items need more than one
      // round at 256 threads but fit into one round of 128-thread workgroups the launch has no ragged second round
      // (benchmark: 883 items, 44 -> 39.5 us).
      // SPEC = 1 / 2: the default pipeline's variant (one value parameter, ReproDisparity; Cauchy / Huber) fixed at compile time
      const int spec = (c.L.N == 1 && c.L.lossType == CVD_STATIC_REPRO_DISPARITY) ? (c.L.robustKind == 0 ? 1 : 2) : 0;
      // Slots of 256-thread workgroups on the device: waves per SIMD the variant is compiled for (cvd_kernels.h) x CUs; 128-thread
      // workgroups: twice as many, as far as the LDS of a CU holds them.  One round of 256 when the items fit; one round of 128
      // when only that fits; else 256 (rounds 2-4 assumed two waves per SIMD for every variant).
      const int wavesPerSimd = spec ? ((c.KD <= 4) ? CVD_MV_WAVES : 3) : 2;
      const long long slots256 = static_cast<long long>(wavesPerSimd) * h->numCU;
      const long long slots128 = std::min<long long>(2 * slots256, static_cast<long long>(kMaxLds / std::max<size_t>(ldsFast, 1)) * h->numCU);
#ifndef CVD_MV_NT_RULE
#define CVD_MV_NT_RULE 1   // 0: the rule of rounds 2-4 (2 x CUs slots assumed)
#endif
      const int nt = CVD_MV_NT_RULE ? ((c.nItems <= slots256 || c.nItems > slots128) ? 256 : 128)
                                    : (c.nItems > 2 * h->numCU && c.nItems <= 4 * h->numCU ? 128 : 256);
#define CVD_LAUNCH_PAIRS_FAST_S(NTV, SPECV)                                                                              \
      CVD_DISPATCH_KD(c.KD, {                                                                                            \
        allowLds((k_matvec_pairs_fast<KD, NTV, SPECV>), ldsFast);                                                        \
        if (evStart)                                                                                                     \
          hipExtLaunchKernelGGL((k_matvec_pairs_fast<KD, NTV, SPECV>), dim3(c.nItems), dim3(NTV), ldsFast, s, evStart, evStop, 0, \
                                c.L, c.T, c.it, x, fcp, maskp, z, pOld, scalp, useBeta, h->dQPart.p, cF);                \
        else                                                                                                             \
          hipLaunchKernelGGL((k_matvec_pairs_fast<KD, NTV, SPECV>), dim3(c.nItems), dim3(NTV), ldsFast, s, c.L, c.T, c.it, x, \
                             fcp, maskp, z, pOld, scalp, useBeta, h->dQPart.p, cF);                                      \
      })
#define CVD_LAUNCH_PAIRS_FAST(NTV) do { if (spec == 1) CVD_LAUNCH_PAIRS_FAST_S(NTV, 1); else if (spec == 2) CVD_LAUNCH_PAIRS_FAST_S(NTV, 2); else CVD_LAUNCH_PAIRS_FAST_S(NTV, 0); } while (0)
      if (h->dense) {
        // dense mode: flow / mask / depth read directly (17 B per pixel pair), grid columns in 8 lane-keyed private copies
        const size_t ldsDense = ldsFast + 8 * 2 * B * 8;
#define CVD_LAUNCH_PAIRS_DENSE(SPECV)                                                                                     \
        CVD_DISPATCH_KD(c.KD, {                                                                                          \
          if constexpr (KD <= 4) { /* (dense mode: Global and bilinear grids) */                                         \
            allowLds((k_matvec_pairs_fast<KD, 256, SPECV, true>), ldsDense);                                             \
            if (evStart)                                                                                                 \
              hipExtLaunchKernelGGL((k_matvec_pairs_fast<KD, 256, SPECV, true>), dim3(c.nItems), dim3(256), ldsDense, s, evStart, \
                                    evStop, 0, c.L, c.T, c.it, x, fcp, maskp, z, pOld, scalp, useBeta, h->dQPart.p, cF); \
            else                                                                                                         \
              hipLaunchKernelGGL((k_matvec_pairs_fast<KD, 256, SPECV, true>), dim3(c.nItems), dim3(256), ldsDense, s, c.L, c.T, \
                                 c.it, x, fcp, maskp, z, pOld, scalp, useBeta, h->dQPart.p, cF);                         \
          }                                                                                                              \
        })
        if (spec == 1) CVD_LAUNCH_PAIRS_DENSE(1);
        else if (spec == 2) CVD_LAUNCH_PAIRS_DENSE(2);
        else CVD_LAUNCH_PAIRS_DENSE(0);  // (the other reprojection losses: runtime branches)
#undef CVD_LAUNCH_PAIRS_DENSE
      }
#if CVD_DETERMINISTIC
      else { (void)nt; CVD_LAUNCH_PAIRS_FAST(64); }  // (one wave per work item: its LDS atomics land in program order)
#else
      else if (nt == 128) CVD_LAUNCH_PAIRS_FAST(128);
      else CVD_LAUNCH_PAIRS_FAST(256);
#endif
#undef CVD_LAUNCH_PAIRS_FAST_S
#undef CVD_LAUNCH_PAIRS_FAST
    } else {
      CVD_DISPATCH(c.KD, c.KS, {
        allowLds(k_matvec_pairs<KD, KS>, lds);
        if (evStart)
          hipExtLaunchKernelGGL((k_matvec_pairs<KD, KS>), dim3(c.nItems), dim3(256), lds, s, evStart, evStop, 0, c.L, c.T,
                                c.it, x, fcp, maskp, z, pOld, scalp, useBeta, h->dQPart.p, cF);
        else
          hipLaunchKernelGGL((k_matvec_pairs<KD, KS>), dim3(c.nItems), dim3(256), lds, s, c.L, c.T, c.it, x, fcp, maskp, z,
                             pOld, scalp, useBeta, h->dQPart.p, cF);
      });
    }
    HIP_CHECK(hipGetLastError());
  }
  if (c.trip && c.TT.nGroups > 0) {
    const size_t ldsT = (9 * B + 3 * kCB) * 8;
    CVD_DISPATCH(c.KD, c.KS, {
      allowLds(k_matvec_triplets<KD, KS>, ldsT);
      hipLaunchKernelGGL((k_matvec_triplets<KD, KS>), dim3(c.TT.nGroups), dim3(256), ldsT, s, c.L, c.TT, x, h->dFc.p,
                         h->dMask.p, z, pOld, h->dScal.p, useBeta, h->dQPart.p, cF);
    });
    HIP_CHECK(hipGetLastError());
  }
  if (!tailFused) {
    if (B > 512) throw std::runtime_error("frame block larger than 512 unknowns is not supported by k_matvec_finish");
    const TlStep ts = temporalStep(h);
    // xf, pf, qf + red[6] + flag + coarse correction (+ third level: its coefficients and the restriction's products)
    const size_t lds = 3 * B * 8 + (8 + kCB) * 8 + (ts.Ainv != nullptr ? (kTlMaxS + static_cast<size_t>(ts.S) * ts.width) * 8 : 0);
    // column half of the fused coarse update y <- y - alpha W (Z^T q) (the row half is in k_cg_update)
    const bool fusedCoarse = withCoarse && !h->coarse.denseMode;
    const bool denseFused = withCoarse && h->coarse.denseMode;  // (needs Z^T q: DenseStep)
    // pair-sharded mode, FUSED exchange (dense coarse level or none): this rank's q, Z^T q and p.q lie contiguously behind
    // q and travel in one all-reduce; k_cg_update forms alpha from the reduced p.q.  With the sparse coarse level the
    // column products need the REDUCED Z^T q: q alone is exchanged and k_dot_pq finishes the product.
    const bool fusedX = h->dist() && fusedExchange(h, withCoarse);
    double* qcX = q + exchangeOffsetQc(c);
    double* pqX = q + exchangeOffsetPq(c, denseFused);
    const CoarseColumns cc{h->coarse.pos.p, h->coarse.wPtr.p, h->coarse.wSlot.p, fusedCoarse ? h->coarse.Wb.p : nullptr,
                           h->coarse.wq.p};
    const int slot = h->tBegin(KC_MATVEC_FINISH);
    CVD_DISPATCH_KD(c.KD, {
      hipLaunchKernelGGL((k_matvec_finish<KD>), dim3(c.L.F), dim3(256), lds, s, c.L, x, h->dMask.p, lam,
                         h->dMedian.p, h->dRegOwner.p, h->dInRange.p, c.cross ? h->dXFiOff.p : h->dFiOff.p, h->dFiList.p,
                         h->dQPart.p, z, pOld, pNew, h->dScal.p, h->dCounters.p, useBeta, q, h->dFdot.p,
                         h->dist() ? (h->rank == 0 ? 1 : 2) : 0, c.cross ? static_cast<int>(h->xFa.size()) * 2 : h->qRows,
                         h->regCache, cF,
                         fusedX ? (denseFused ? qcX : nullptr) : (((fusedCoarse || denseFused) && !h->dist()) ? h->coarse.qc.p : nullptr), cc,
                         c.cross ? h->dH.p : nullptr, fusedX ? pqX : nullptr, h->dist() ? h->ownFirst() : 0,
                         h->dist() ? h->ownCount() : ((c.cross && h->xDiagRows) ? 0 : c.L.F), ts.Ainv != nullptr ? temporalStepDev(h) : static_cast<const TlStep*>(nullptr));
    });
    HIP_CHECK(hipGetLastError());
    if (fusedX && ownerShardedUpdate(h, withCoarse)) {
      // q to the frames' owners (in place: the reduced chunk lands where the frames' q lives), [Z^T q | p.q] to everybody
      const int ct = h->tBegin(KC_COMM_PRODUCT);
      const size_t chunk = static_cast<size_t>(h->ownChunk()) * B;
      commGroupStart(h);
      commReduceScatter(h, q, q + static_cast<size_t>(h->rank) * chunk, chunk, CT_F64, s);
      commAllReduce(h, qcX, exchangeOffsetPq(c, denseFused) - exchangeOffsetQc(c) + 1 + exchangeTemporalCount(h, withCoarse), CT_F64, s);
      commGroupEnd(h);
      h->tEnd(ct);
    } else if (fusedX) {
      const int ct = h->tBegin(KC_COMM_PRODUCT);
      commAllReduce(h, q, exchangeOffsetPq(c, denseFused) + 1 + exchangeTemporalCount(h, withCoarse), CT_F64, s);
      h->tEnd(ct);
    } else if (h->dist()) {
      // per-product exchange: q (F x B doubles) summed over the pair shards, then p.q / alpha on the reduced vector
      const int ct = h->tBegin(KC_COMM_PRODUCT);
      commAllReduce(h, q, c.n, CT_F64, s);
      h->tEnd(ct);
      hipLaunchKernelGGL(k_dot_pq, dim3(c.L.F), dim3(256), 0, s, c.L, pNew, q, h->dScal.p, h->dCounters.p, h->dFdot.p,
                         withCoarse ? h->coarse.qc.p : nullptr, h->coarse.modeActive.p, cc,
                         ts.Ainv != nullptr ? temporalStepDev(h) : static_cast<const TlStep*>(nullptr));
      HIP_CHECK(hipGetLastError());
    }
    h->tEnd(slot);
  }
}

// ---- k_pcg_tail: finish + update of a PCG iteration in one launch (cvd_kernels.h) ----------------------------------------
// Scope and launch geometry: one GPU, frame block <= 256, dense coarse level or none, and every workgroup of the launch
// resident at once (the kernel has a grid barrier): checked against the kernel's occupancy on this device.
bool pcgTailScope(Ctx& c, bool coarse, int nThreads, size_t& lds, int& ldsFinish, int& ldsScratch) {
  cvd_handle* h = c.h;
  const int F = c.L.F, B = c.L.B;
  if (!h->opt.pcg_fused_tail || h->tailDisabled || h->dist() || B > 256 || h->forceGeneric) return false;
  if (coarse && !h->coarse.denseMode) return false;
  const int split = denseRowSplit(h);
  const TlStep ts = temporalStep(h);
  // (third level: S more workgroups; the restriction's products of a frame use the update half's partial-sum region)
  if (ts.Ainv != nullptr && (ts.S * ts.width > cgUpdatePartDoubles(B, nThreads) || tlRowsLds(ts.NT) > B + cgUpdatePartDoubles(B, nThreads)))
    return false;
  const bool poseT = coarse && h->coarse.temporalPose;  // (coarse_level 3: kCB workgroups walk the node-reduced inverse, no dense-level ones)
  if (poseT && tlRowsLds(h->coarse.ptN) > B + cgUpdatePartDoubles(B, nThreads)) return false;
  const int grid = F + (coarse && !poseT && split > 0 ? (F + kDenseFramesPerGroup - 1) / kDenseFramesPerGroup : 0) +
                   (ts.Ainv != nullptr ? ts.S * ts.parts : 0) + (poseT ? kCB * tlParts(h->coarse.ptNn) : 0);
  const int update = B + cgUpdatePartDoubles(B, nThreads) + 48 + 17 * kCB;  // k_cg_update's region (cvd_solve.hip: ldsU)
  const int finish = 3 * B + 8 + kCB + (nThreads / 256 - 1) * 256 + kTlMaxS;  // k_matvec_finish's + the partial sums of its row walk + the third level's coefficients
  const int denseEnd = (coarse && !poseT) ? F * kCB + nThreads + 16 : 0;    // the dense-level workgroups' (Z^T q + partial sums)
  ldsFinish = update;
  // (frame workgroups that walk rows of the dense level themselves: Z^T q + partial sums behind their two regions)
  ldsScratch = std::max(update + finish + (coarse && split < kCB ? denseEnd : 0), denseEnd);
  lds = static_cast<size_t>(ldsScratch + 24) * sizeof(double);
  if (lds > kMaxLds) return false;
  // occupancy of this (kernel, block size, LDS size) on this device: asked once
  static std::map<std::tuple<int, int, int, size_t>, int> perCuCache;
  static std::mutex cacheMutex;
  int perCu = 0;
  {
    std::lock_guard<std::mutex> lock(cacheMutex);
    const auto key = std::make_tuple(h->device, c.KD, nThreads, lds);
    auto it = perCuCache.find(key);
    if (it == perCuCache.end()) {
      CVD_DISPATCH_KD(c.KD, {
        allowLds((k_pcg_tail<KD>), lds);
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(&k_pcg_tail<KD>), nThreads, lds));
      });
      perCuCache[key] = perCu;
    } else {
      perCu = it->second;
    }
  }
  return static_cast<long long>(perCu) * h->numCU >= grid;
}

void launchPcgTail(Ctx& c, const double* x, const double* pOld, double* pNew, int useBeta, const double* lam, double* q,
                   bool withCoarse, int nThreads, size_t lds, int ldsFinish, int ldsScratch, double tol2) {
  cvd_handle* h = c.h;
  hipStream_t s = h->stream;
  const int F = c.L.F;
  const CoarseView cF = coarseView(h, withCoarse, false);
  const int split = denseRowSplit(h);
  const int finishDoubles = 3 * static_cast<int>(c.L.B) + 8 + kCB + (nThreads / 256 - 1) * 256 + kTlMaxS;  // (pcgTailScope)
  const TlStep ts = temporalStep(h);
  const bool poseT = withCoarse && h->coarse.temporalPose;
  const DenseStep ds = (withCoarse && !poseT) ? DenseStep{h->coarse.denseInv.p, h->coarse.qc.p, h->coarse.rc.p, h->coarse.c.p, h->coarse.dotPart.p,
                                               h->coarse.modeActive.p, h->coarse.fail.p, split, ldsFinish + finishDoubles,
                                               h->coarse.dotPart.p + F}
                                  : DenseStep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kCB, 0, nullptr};
  double* fd = h->dFdot.p;
  const TailUpdate U{h->dMinv.p, h->dDx.p, h->dR.p, h->dZ.p, fd + F, fd + 2 * F, tol2, h->coarse.modeActive.p, h->hPcg,
                     h->dCounters.p + 1, h->dTailBar.p, h->dFdot.p + 4 * static_cast<size_t>(F) + 32, ldsFinish, ldsScratch, ds,
                     ts.Ainv != nullptr ? temporalStepDev(h) : static_cast<const TlStep*>(nullptr),
                     poseT ? poseTemporalStepDev(h) : static_cast<const TlStep*>(nullptr),
                     (c.cross && h->xDiagRows) ? 1 : 0};
  const int grid = F + (withCoarse && !poseT && split > 0 ? (F + kDenseFramesPerGroup - 1) / kDenseFramesPerGroup : 0) +
                   (ts.Ainv != nullptr ? ts.S * ts.parts : 0) + (poseT ? kCB * tlParts(h->coarse.ptNn) : 0);
  const int slot = h->tBegin(KC_CG_UPDATE);  // (timed under the update class: the finish class stays empty on this path)
  {
    // several handles of this process on one device: their grid-barrier kernels must not overlap (PersistentGate).  A handle
    // that is alone launches under the slot's mutex only (no events in the stream): cvd_create counts a second handle under the
    // same mutex and drains the device, so no ungated launch can still be in flight once another handle exists.
    std::unique_ptr<PersistentGate> gate;
    std::unique_lock<std::mutex> alone(PersistentGate::slot(h->device).m);
    if (liveHandles(h->device) > 1) {
      alone.unlock();
      gate.reset(new PersistentGate(h->device, s));
    }
    CVD_DISPATCH_KD(c.KD, {
      hipLaunchKernelGGL((k_pcg_tail<KD>), dim3(grid), dim3(nThreads), lds, s, c.L, x, h->dMask.p, lam, h->dMedian.p, h->dRegOwner.p,
                         h->dInRange.p, c.cross ? h->dXFiOff.p : h->dFiOff.p, h->dFiList.p, h->dQPart.p, pOld, pNew, h->dScal.p,
                         useBeta, q, fd, c.cross ? static_cast<int>(h->xFa.size()) * 2 : h->qRows, h->regCache, cF,
                         withCoarse ? h->coarse.qc.p : static_cast<double*>(nullptr),
                         c.cross ? static_cast<const double*>(h->dH.p) : static_cast<const double*>(nullptr), U);
    });
    HIP_CHECK(hipGetLastError());
  }
  h->tEnd(slot);
}

// One kernel of this translation unit's code object is looked up at handle creation: the HIP runtime loads a unit's device
// code at its first use, ~20 ms per unit that would otherwise land in the first solve of a process (cvd_create: loadDeviceCode).
void touchModule_matvec() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_dot_pq));
}

}  // namespace cvd

#ifdef CVD_TAIL_PROFILE
extern "C" int32_t cvd_debug_tail_profile(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cvd::g_tailProf), sizeof(unsigned long long) * 1024 * 8) == hipSuccess ? 0 : 1;
}
#endif
#ifdef CVD_MV_PROFILE
extern "C" int32_t cvd_debug_mv_profile(unsigned long long* out) {  // (same translation unit as the launches: the symbol is per unit)
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cvd::g_mvProf), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : 1;
}
#endif
