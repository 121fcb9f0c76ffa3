// cvd_eval.hip -- cost / gradient / frame-diagonal blocks (and the dense mode's explicit cross blocks) at a point.
#include "cvd_host.h"

namespace cvd {

void launchFrameConsts(Ctx& c, const double* x) {
  hipLaunchKernelGGL(k_frame_consts, dim3((c.L.F + 63) / 64), dim3(64), 0, c.h->stream, c.L, x, c.h->dFc.p);
  HIP_CHECK(hipGetLastError());
}

// The LM loop is latency-bound on its read-backs (a handful of scalars per decision): poll instead of the
// interrupt-driven hipStreamSynchronize / hipEventSynchronize, whose wake-up costs tens of microseconds.
void spinStream(hipStream_t s) {
  hipError_t e;
  while ((e = hipStreamQuery(s)) == hipErrorNotReady) {}
  HIP_CHECK(e);
}
void spinEvent(hipEvent_t ev) {
  hipError_t e;
  while ((e = hipEventQuery(ev)) == hipErrorNotReady) {}
  HIP_CHECK(e);
}
void readScalars(Ctx& c) {
  HIP_CHECK(hipMemcpyAsync(c.h->hScal, c.h->dScal.p, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, c.h->stream));
  spinStream(c.h->stream);
}

// cost only at x: enqueueCost leaves it in S_COST on the device, evalCost also reads it back
void enqueueCost(Ctx& c, const double* x) {
  cvd_handle* h = c.h;
  hipStream_t s = h->stream;
  launchFrameConsts(c, x);
  const int slot = h->tBegin(KC_COST);
  if (c.L.includeStatic && c.nItems > 0) {
    const size_t lds = (2 * c.L.B) * 8 + 2 * sizeof(FrameConst) + 8 * 8;
    const bool fast = !h->forceGeneric && c.KS == 0 && fastLoss(c.L);  // (scope of the fast kernels)
    if (fast) {
      CVD_DISPATCH_KD(c.KD, {
        if (h->dense) {
          allowLds((k_cost_items_fast<KD, true>), lds);
          hipLaunchKernelGGL((k_cost_items_fast<KD, true>), dim3(c.nItems), dim3(256), lds, s, c.L, c.T, c.it, x, h->dFc.p,
                             h->dCostItem.p);
        } else {
          allowLds((k_cost_items_fast<KD, false>), lds);
          hipLaunchKernelGGL((k_cost_items_fast<KD, false>), dim3(c.nItems), dim3(256), lds, s, c.L, c.T, c.it, x, h->dFc.p,
                             h->dCostItem.p);
        }
      });
    } else {
      CVD_DISPATCH(c.KD, c.KS, {
        allowLds(k_cost_items<KD, KS>, lds);
        hipLaunchKernelGGL((k_cost_items<KD, KS>), dim3(c.nItems), dim3(256), lds, s, c.L, c.T, c.it, x, h->dFc.p,
                           h->dCostItem.p);
      });
    }
    HIP_CHECK(hipGetLastError());
  }
  CVD_DISPATCH_KD(c.KD, {
    hipLaunchKernelGGL((k_cost_frames<KD>), dim3(c.L.F), dim3(256), static_cast<size_t>(c.L.B) * 8, s, c.L, x, h->dMedian.p, h->dRegOwner.p, h->dInRange.p,
                       h->dCostFrame.p);
  });
  HIP_CHECK(hipGetLastError());
  if (c.trip && c.TT.nGroups > 0) {
    // scene-flow smoothness: group costs are added to the centre frames' entries
    CVD_DISPATCH(c.KD, c.KS, {
      hipLaunchKernelGGL((k_cost_triplets<KD, KS>), dim3(c.TT.nGroups), dim3(256), 0, s, c.L, c.TT, x, h->dFc.p,
                         h->dCostFrame.p);
    });
    HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, s, h->dCostItem.p, (c.L.includeStatic ? c.nItems : 0),
                     h->dCostFrame.p, c.L.F, h->dScal.p, S_COST);
  HIP_CHECK(hipGetLastError());
  if (h->dist()) commAllReduce(h, h->dScal.p + S_COST, 1, CT_F64, s);
  h->tEnd(slot);
}
double evalCost(Ctx& c, const double* x) {
  enqueueCost(c, x);
  readScalars(c);
  return c.h->hScal[S_COST];
}

// step statistics (k_step_stats) into the device scalars; the caller reads them back
void enqueueStats(Ctx& c) {
  cvd_handle* h = c.h;
  const int G = static_cast<int>(std::min<size_t>(128, (c.n + 511) / 512));
  h->dStatPart.ensure(7 * 128);
  hipLaunchKernelGGL(k_step_stats, dim3(G), dim3(256), 0, h->stream, c.n, h->dDx.p, h->dG.p, h->dR.p, h->dLam.p,
                     h->dX.p, h->dHd.p, h->dScal.p, h->dStatPart.p, h->dCounters.p + 2);
  HIP_CHECK(hipGetLastError());
}

// cost + gradient + diagonal blocks at x
// Dense mode with explicit cross blocks (cvd_cross.h) whenever the problem is in its scope.  (Bilinear grids only: at the
// Global level every pixel hits the one vertex -- same-address LDS atomics, 52 ms per assembly measured -- and the 8 x 8
// problem is cheap to solve matrix-free.)
bool crossScope(cvd_handle* h, const Ctx& c) {
  const bool off = h->opt.dense_matrix_free != 0;  // comparison variant
  return h->dense && !off && !h->forceGeneric && c.L.includeStatic && !h->xFa.empty() && c.KS == 0 && fastLoss(c.L) &&
         c.L.N == 1 && c.L.nD > 0 && (c.KD == 4 || (c.KD == 1 && c.L.nD == 1)) && c.L.intrOpt != CVD_INTR_SHARED && !c.trip && !(c.L.positionRegSqrt > 0.0) &&
         c.L.B <= 256 && (static_cast<size_t>(c.L.B) * (c.L.B + 1) / 2 + 2 * c.L.B + 4 * 36) * 8 <= kMaxLds &&  // (the fold's packed block)
         dwLdsBytes(c.L.nD, c.L.B, kDwThreads) <= kMaxLds;
}
CrossPairs crossPairs(cvd_handle* h) {
  return CrossPairs{h->dXFa.p, h->dXFb.p, h->dXRange.p, h->dXSlot.p, static_cast<int>(h->xFa.size())};
}
// Dense mode, explicit-block scope: the ONE walk over the pixels (cvd_dense_walk.h) -- every directed pair's record and the
// per-pixel grid x grid scalars at the linearisation point x (frame constants in dFc are those of x).
DenseRecords denseRecords(cvd_handle* h) {
  return DenseRecords{h->dDwRecords.p, h->dDwRecOff.p, h->dFpOff.p, h->dFpList.p};
}
void launchDenseWalk(Ctx& c, const double* x) {
  cvd_handle* h = c.h;
  const int G = c.L.nD, B = c.L.B;
  h->dDwRecords.ensure(static_cast<size_t>(std::max(1, h->nDwRecords)) * dwRecordDoubles(G));
  h->dDwGg.ensure(static_cast<size_t>(std::max<long long>(1, h->C)));
  if (h->nDwRecords == 0) return;
  const size_t lds = dwLdsBytes(G, B, kDwThreads);
  const DenseWalkList wl{h->dDwPair.p, h->nDwRecords, denseLaneMap(h->W, h->H)};
  allowLds((k_dense_walk<4>), lds);
  const int slot = h->tBegin(KC_DENSE_WALK);
  hipLaunchKernelGGL((k_dense_walk<4>), dim3(h->nDwRecords), dim3(kDwThreads), lds, h->stream, c.L, c.T, wl, x, h->dFc.p,
                     h->dDwRecords.p, h->dDwGg.p);
  h->tEnd(slot);
  HIP_CHECK(hipGetLastError());
}
// X_ab of every undirected pair from the walk's records (pose rows / columns) and scalars (grid x grid, in column panels of the
// largest width that fits the LDS)
void launchCrossAssemble(Ctx& c, const double* x) {
  (void)x;
  cvd_handle* h = c.h;
  const size_t B = c.L.B, G = c.L.nD;
  h->dXBlocks.ensure(h->xFa.size() * B * B);
  const unsigned nP = static_cast<unsigned>(h->xFa.size());
  if (nP == 0) return;
  hipLaunchKernelGGL(k_dense_fold_cross, dim3(nP), dim3(256), 0, h->stream, c.L, crossPairs(h), h->dXDir.p, h->dDwRecOff.p,
                     h->dDwRecords.p, h->dXBlocks.p);
  int panelW = static_cast<int>(((kMaxLds - 4096) / 8) / G);
  panelW = std::max(1, std::min<int>(panelW, static_cast<int>(G)));
  const int nPanels = static_cast<int>((G + panelW - 1) / panelW);
  panelW = static_cast<int>((G + nPanels - 1) / nPanels);  // (even panels)
  const size_t ldsGrid = static_cast<size_t>(panelW) * G * 8;
  allowLds((k_dense_gg<4>), ldsGrid);
  const int slot = h->tBegin(KC_DENSE_GG);
  for (int dir = 0; dir < 2; ++dir)   // (a -> b: panels of rows, written; b -> a: panels of columns, added)
    hipLaunchKernelGGL((k_dense_gg<4>), dim3(nP, nPanels), dim3(kGgThreads), ldsGrid, h->stream, c.L, c.T, crossPairs(h), h->dXDir.p,
                       h->dDwGg.p, panelW, dir, h->dXBlocks.p);
  h->tEnd(slot);
  HIP_CHECK(hipGetLastError());
}

// noReadBack: everything is enqueued and the host does not wait (the LM loop reads |g|_max / |x| of the new point with the next
// PCG solve's scalars, which recompute them: cvd_solve.hip); returns 0.
double evalFull(Ctx& c, const double* x, bool withStats, bool noReadBack) {
  cvd_handle* h = c.h;
  hipStream_t s = h->stream;
  launchFrameConsts(c, x);
  const size_t B = c.L.B;
  const int slot = h->tBegin(KC_ASSEMBLE);
  const bool fast = !h->forceGeneric && c.KS == 0 && fastLoss(c.L);
  const size_t ldsFast = (B * (B + 1) / 2 + 2 * B + 4 * 36) * 8;
  // generic kernel: the packed triangle in row panels when it does not fit the LDS in one piece (B > 199)
  const size_t ldsRest = 3 * B * 8 + 2 * sizeof(FrameConst) + 4 * 36 * 8;
  int panelCap = 0;
  const bool fastFits = ldsFast <= kMaxLds;
  if (fast && fastFits) {
    h->dAsmScratch.ensure(static_cast<size_t>(h->nAsmSlots) * (B * (B + 1) / 2 + B + 4));
    const AsmWork work{h->dAsmParts.p, h->dAsmUnits.p, h->dAsmScratch.p, h->dAsmCount.p};
    // STAGE: the other frame's parameters in a per-wave LDS buffer whenever the LDS holds 8 B doubles more (cvd_kernels.h)
    const size_t ldsStage = ldsFast + static_cast<size_t>(kAsmThreads / 64) * B * 8;
    const bool stage = ldsStage <= kMaxLds;
#define CVD_LAUNCH_ASM(DENSEV, STAGEV)                                                                                     \
    CVD_DISPATCH_KD(c.KD, {                                                                                              \
      allowLds((k_assemble_fast<KD, DENSEV, STAGEV>), STAGEV ? ldsStage : ldsFast);                                        \
      hipLaunchKernelGGL((k_assemble_fast<KD, DENSEV, STAGEV>), dim3(h->nAsmParts), dim3(kAsmThreads), STAGEV ? ldsStage : ldsFast, s, \
                         c.L, c.T, x, h->dFc.p, h->dMask.p, h->dMedian.p, h->dRegOwner.p, h->dInRange.p, work, h->dG.p, h->dH.p,  \
                         h->dCostFrame.p, h->dFocal.p, h->dFocal.p + c.L.F);                                             \
    })
#ifndef CVD_ASM_STAGE
#define CVD_ASM_STAGE 1
#endif
    if (c.cross) {
      // explicit-block scope of the dense mode: one walk over the pixels, then a per-frame fold of the pairs' records
      // (a Global transform -- the first level of the default schedule -- is walked as a 1 x 1 grid; its fold and regularisers
      // are the one-tap instantiation's.  Until round 6 that level ran matrix-free: 1.6 ms per PCG product over the 144 M pixels
      // for an 8 x 8 block per pair, more than half of the dense pipeline's time.)
      launchDenseWalk(c, x);
      if (c.KD == 1) {
        allowLds((k_assemble_fast<1, true, false, true>), ldsFast);
        hipLaunchKernelGGL((k_assemble_fast<1, true, false, true>), dim3(c.L.F), dim3(kAsmThreads), ldsFast, s, c.L, c.T, x, h->dFc.p,
                           h->dMask.p, h->dMedian.p, h->dRegOwner.p, h->dInRange.p, work, h->dG.p, h->dH.p, h->dCostFrame.p,
                           h->dFocal.p, h->dFocal.p + c.L.F, denseRecords(h));
      } else {
        allowLds((k_assemble_fast<4, true, false, true>), ldsFast);
        hipLaunchKernelGGL((k_assemble_fast<4, true, false, true>), dim3(c.L.F), dim3(kAsmThreads), ldsFast, s, c.L, c.T, x, h->dFc.p,
                           h->dMask.p, h->dMedian.p, h->dRegOwner.p, h->dInRange.p, work, h->dG.p, h->dH.p, h->dCostFrame.p,
                           h->dFocal.p, h->dFocal.p + c.L.F, denseRecords(h));
      }
    } else if (h->dense) { if (stage && CVD_ASM_STAGE) CVD_LAUNCH_ASM(true, true); else CVD_LAUNCH_ASM(true, false); }
    else { if (stage && CVD_ASM_STAGE) CVD_LAUNCH_ASM(false, true); else CVD_LAUNCH_ASM(false, false); }
#undef CVD_LAUNCH_ASM
  } else {
    if (h->dense && c.L.includeStatic) throw std::logic_error("dense mode: the generic assembly has no images to read (DenseListScope should have taken this solve)");
    const AsmPanels panels = makePanels(static_cast<int>(B), (kMaxLds - ldsRest) / 8, panelCap);
    const size_t lds = static_cast<size_t>(panelCap) * 8 + ldsRest;
    CVD_DISPATCH(c.KD, c.KS, {
      allowLds(k_assemble<KD, KS>, lds);
      hipLaunchKernelGGL((k_assemble<KD, KS>), dim3(c.L.F), dim3(256), lds, s, c.L, c.T, x, h->dFc.p, h->dMask.p,
                         h->dMedian.p, h->dRegOwner.p, h->dInRange.p, h->dFpOff.p, h->dFpList.p, h->dG.p, h->dH.p, h->dCostFrame.p,
                       h->dFocal.p, h->dFocal.p + c.L.F, panels, panelCap);
    });
  }
  HIP_CHECK(hipGetLastError());
  if (c.trip && c.TT.nGroups > 0) {
    // scene-flow smoothness: its share of g / H_ff is added to the pair assembly's output, its cost to the frame sums
    int panelCapT = 0;
    const AsmPanels panelsT = makePanels(static_cast<int>(B), (kMaxLds - B * 8 - 256) / 8, panelCapT);
    const size_t ldsT = (static_cast<size_t>(panelCapT) + B) * 8;
    CVD_DISPATCH(c.KD, c.KS, {
      allowLds(k_assemble_triplets<KD, KS>, ldsT);
      hipLaunchKernelGGL((k_assemble_triplets<KD, KS>), dim3(c.L.F), dim3(256), ldsT, s, c.L, c.TT, x, h->dFc.p,
                         h->dMask.p, h->dFtOff.p, h->dFtList.p, h->dG.p, h->dH.p, h->dFocal.p, h->dFocal.p + c.L.F,
                         panelsT, panelCapT);
      hipLaunchKernelGGL((k_cost_triplets<KD, KS>), dim3(c.TT.nGroups), dim3(256), 0, s, c.L, c.TT, x, h->dFc.p,
                         h->dCostFrame.p);
    });
    HIP_CHECK(hipGetLastError());
  }
  if (c.L.intrOpt == CVD_INTR_SHARED) {  // (after the triplet assembly: it adds its share of the focal sums)
    hipLaunchKernelGGL(k_shared_focal_fixup, dim3(1), dim3(256), 0, s, c.L, h->dFocal.p, h->dFocal.p + c.L.F, h->dMask.p,
                       h->dG.p, h->dH.p);
    HIP_CHECK(hipGetLastError());
  }
  if (c.cross) launchCrossAssemble(c, x);  // (same timing class: it is part of the Jacobian evaluation)
  h->tEnd(slot);
  if (h->dist()) {
    // The exchange step of the pair-sharded mode, once per Jacobian evaluation: the gradient and the per-frame costs
    // are all-reduced (F x B + F doubles); the frame blocks H_ff are REDUCE-SCATTERED to the frames' owners (75 MB at
    // B = 177: each rank receives 1 / world of it), which extract the diagonal, invert their own blocks and all-gather
    // the results -- diag(H) here (F x B doubles), the f32 inverses after the damping is known (launchBlockInverse), the
    // 8x8 coarse diagonal blocks in launchCoarseSetup.  Against one all-reduce of H_ff: 3/4 of the bytes on the wire and
    // 1 / world of the inverse work per rank.
    const int ct = h->tBegin(KC_COMM_EVAL);
    const size_t chunkH = static_cast<size_t>(h->ownChunk()) * B * B;
    commGroupStart(h);
    commAllReduce(h, h->dG.p, c.n, CT_F64, s);
    commAllReduce(h, h->dCostFrame.p, c.L.F, CT_F64, s);
    commReduceScatter(h, h->dH.p, h->dH.p + static_cast<size_t>(h->rank) * chunkH, chunkH, CT_F64, s);
    commGroupEnd(h);
    Layout own = c.L;
    own.F = h->ownCount();
    if (own.F > 0)
      hipLaunchKernelGGL(k_extract_diag, dim3((static_cast<size_t>(own.F) * B + 255) / 256), dim3(256), 0, s, own,
                         h->dH.p + static_cast<size_t>(h->ownFirst()) * B * B, h->dHd.p + static_cast<size_t>(h->ownFirst()) * B);
    const size_t chunkD = static_cast<size_t>(h->ownChunk()) * B;
    commAllGather(h, h->dHd.p + static_cast<size_t>(h->rank) * chunkD, h->dHd.p, chunkD, CT_F64, s);
    h->tEnd(ct);
  } else {
    hipLaunchKernelGGL(k_extract_diag, dim3((c.n + 255) / 256), dim3(256), 0, s, c.L, h->dH.p, h->dHd.p);
  }
  hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, s, h->dCostFrame.p, c.L.F, h->dCostFrame.p, 0, h->dScal.p, S_COST);
  HIP_CHECK(hipGetLastError());
  if (noReadBack) return 0.0;
  if (withStats) {  // |g|_max and |x| of the new point in the same read-back (lam = 0: only those two are used)
    HIP_CHECK(hipMemsetAsync(h->dLam.p, 0, c.n * sizeof(double), s));
    enqueueStats(c);
  }
  readScalars(c);
  return h->hScal[S_COST];
}

// One kernel of this translation unit's code object is looked up at handle creation: the HIP runtime loads a unit's device
// code at its first use, ~20 ms per unit that would otherwise land in the first solve of a process (cvd_create: loadDeviceCode).
void touchModule_eval() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_sum2));
}

}  // namespace cvd
