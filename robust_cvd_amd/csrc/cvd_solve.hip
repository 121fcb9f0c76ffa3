// cvd_solve.hip -- the PCG loop, the Levenberg-Marquardt driver (Ceres defaults) and the evaluation hook.
#include "cvd_host.h"

namespace cvd {

// PCG on (H + diag(lam)) dx = -g with the block-Jacobi preconditioner; returns iterations used.
// Three launches per iteration (pairs product, per-frame finish, per-frame update).  alpha / beta live on
// the device: the last workgroup of k_matvec_finish / k_cg_update reduces the per-frame partial dot products
// (agent-scope release/acquire ticket), so there is neither a scalar kernel nor a host round trip in the loop.
namespace {
struct TailStalled {};  // k_pcg_tail's grid barrier was abandoned (its workgroups were not co-resident: the device is shared)
}
static int runPcgAttempt(Ctx& c, const double* x, const std::function<void()>& tail);
int runPcg(Ctx& c, const double* x, const std::function<void()>& tail) {
  const size_t firstTimerSlot = c.h->evUsed;
  try {
    return runPcgAttempt(c, x, tail);
  } catch (const TailStalled&) {
    //  not an error: the fused tail is an optimisation.  The handle falls back to the two-launch path for good and
    // the solve is repeated from its start (the PCG's inputs -- g, lam, the block inverses, the levels -- are untouched; its
    // state vectors and last-workgroup tickets are re-initialised).
    //  EVERY ticket and counter a half-finished fused iteration may have left behind is reset -- the update / finish
    // tickets, the levels' (temporal node sums), the barrier words (the attempt re-zeroes them too) -- and the timer slots of the
    // abandoned attempt are dropped, so the kernel-time averages hold the repeated solve only.
    cvd_handle* h = c.h;
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipMemsetAsync(h->dCounters.p, 0, 8 * sizeof(unsigned int), h->stream));
    if (h->temporal.counter.p) HIP_CHECK(hipMemsetAsync(h->temporal.counter.p, 0, 4 * sizeof(unsigned int), h->stream));
    if (h->dTailBar.p)
      HIP_CHECK(hipMemsetAsync(h->dTailBar.p, 0, static_cast<size_t>(kTailBarStride) * (1 + kTailBarCopies) * sizeof(unsigned int), h->stream));
    h->curPcgIter = -1;
    h->tDropFrom(firstTimerSlot, 0);
    h->tailDisabled = true;
    fprintf(stderr, "[cvd] warning: k_pcg_tail's grid barrier was abandoned (device shared with other work?); this handle uses the "
                    "two-launch PCG tail from here on\n");
    return runPcgAttempt(c, x, tail);
  }
}
static int runPcgAttempt(Ctx& c, const double* x, const std::function<void()>& tail) {
  cvd_handle* h = c.h;
  hipStream_t s = h->stream;
  const int F = c.L.F;
  const size_t B = c.L.B;
  if (B > 512) throw std::runtime_error("frame block larger than 512 unknowns is not supported by k_cg_update");
  prepareMatvec(c, x);
  const int nChunks = static_cast<int>((B + 63) / 64);
  const int nThreads = (B > 256 ? 128 : 256) * nChunks;  // (k_cg_update: four segments per row up to B = 256, two beyond)
  double* fd = h->dFdot.p;
  size_t ldsU = (B + cgUpdatePartDoubles(static_cast<int>(B), nThreads) + 48 + 17 * kCB) * 8;
  const double tol2 = c.h->opt.pcg_relative_tolerance * c.h->opt.pcg_relative_tolerance;
  for (int i = 0; i < 9; ++i) h->hPcg[i] = 0.0;  // device progress mirror (pcgFinishScalars): nothing applied yet
  const bool coarse = h->coarseOn;
  double* rc = coarse ? h->coarse.rc.p : nullptr;
  // Coarse level per iteration: y = W Z^T r is kept up to date inside k_cg_update (CoarseStep: y <- y - alpha W Z^T q,
  // |y|^2 closes r^T z), so only c = W^T y remains as a launch; the first residual goes through k_coarse_apply_w.
  const bool denseCoarse = coarse && h->coarse.denseMode;
  const bool unfusedY = denseCoarse;  // (the dense level has no W to recur on: Z^T r is restricted every iteration)
  auto coarseC = [&](int init) {
    if (c.L.positionRegSqrt > 0.0 || c.trip || !coarseFusedConsumers())
      hipLaunchKernelGGL(k_coarse_apply_wt, dim3(F), dim3(256), 0, s, coarseView(h, true, true), F, h->coarse.c.p,
                         h->dScal.p, init);
  };
  // coarse_level 3: the temporal pose level -- dense-mode plumbing (exchange layout, in-line build), but its rows are walked by
  // kCB extra workgroups of the update launch (tlLevelRows) and no workgroup streams an 8F x 8F inverse
  const bool poseT = denseCoarse && h->coarse.temporalPose;
  auto coarseApply = [&](int init) {
    if (poseT) {
      launchPoseTemporalInit(c, tol2);  // (first residual only: the iterations carry the level inside k_cg_update / k_pcg_tail)
      return;
    }
    if (denseCoarse) {
      hipLaunchKernelGGL(k_coarse_dense_apply, dim3(F), dim3(256), 0, s, F, h->coarse.denseInv.p, h->coarse.rc.p, h->coarse.c.p,
                         h->coarse.modeActive.p, h->coarse.dotPart.p, h->dScal.p, h->dCounters.p + 3, h->coarse.fail.p, init,
                         tol2, h->hPcg);
      return;
    }
    hipLaunchKernelGGL(k_coarse_apply_w, dim3(F), dim3(1024), 0, s, h->coarse.plan, h->coarse.Wb.p, h->coarse.rc.p,
                       h->coarse.y.p, h->coarse.dotPart.p, h->dScal.p, h->dCounters.p + 3, h->coarse.fail.p, init, tol2,
                       h->hPcg);
    coarseC(init);
  };
  const CoarseStep csOff{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const CoarseStep csOn = (coarse && !unfusedY)
                              ? CoarseStep{h->coarse.wtPtr.p, h->coarse.wtBlk.p, h->coarse.wtFrame.p, h->coarse.Wb.p,
                                           h->coarse.qc.p, h->coarse.y.p, h->coarse.fdotY.p, h->coarse.fail.p, h->coarse.wq.p}
                              : csOff;
  const DenseStep dsOff{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, kCB, 0, nullptr};
  // dense level: c <- c - alpha A_c^-1 Z^T q inside k_cg_update (F extra workgroups) instead of a launch of its own
  const bool denseFused = denseCoarse;
  // pair-sharded mode with the fused exchange (cvd_matvec.hip): Z^T q and p.q arrive all-reduced behind q
  const bool fusedX = h->dist() && fusedExchange(h, coarse);
  const double* pqReduced = fusedX ? h->dQ.p + exchangeOffsetPq(c, denseFused) : nullptr;
  const bool denseRows = denseFused && !poseT;
  const DenseStep dsOn = denseRows ? DenseStep{h->coarse.denseInv.p, fusedX ? h->dQ.p + exchangeOffsetQc(c) : h->coarse.qc.p, h->coarse.rc.p, h->coarse.c.p,
                                                h->coarse.dotPart.p, h->coarse.modeActive.p, h->coarse.fail.p,
                                                // (two launches: the split costs 2 % -- the frame workgroups are this kernel's long
                                                // pole already; it pays in k_pcg_tail, whose DenseStep launchPcgTail builds)
                                                kCB, static_cast<int>(ldsU / 8), h->coarse.dotPart.p + F}
                                    : dsOff;
  if (denseRows) {
    // (dense-level workgroups: Z^T q + partial sums; frame workgroups walking rows of their own: the same again behind their region)
    const size_t dense = static_cast<size_t>(F) * kCB + nThreads + 16;
    ldsU = std::max((dsOn.rowSplit < kCB ? ldsU / 8 + dense : ldsU / 8), dense) * 8;
  }
  // third level (cvd_temporal.h): S more workgroups of the update launch (q_T and the coefficients of one coarse hat in LDS)
  const TlStep tsOn = temporalStep(h);
  const int nTlWg = tsOn.Ainv != nullptr ? tsOn.S * tsOn.parts : 0;
  if (nTlWg) ldsU = std::max(ldsU, static_cast<size_t>(tlRowsLds(tsOn.NT)) * 8);
  const int nPtWg = poseT ? kCB * tlParts(h->coarse.ptNn) : 0;
  if (poseT) ldsU = std::max(ldsU, static_cast<size_t>(tlRowsLds(h->coarse.ptN)) * 8);
  const TlStep* tpDev = poseT ? poseTemporalStepDev(h) : nullptr;
  allowLds(k_cg_update, ldsU);
  hipLaunchKernelGGL(k_cg_update, dim3(F), dim3(nThreads), ldsU, s, c.L, 1, h->dG.p, h->dMinv.p, h->dP0.p, h->dQ.p,
                     h->dScal.p, h->dCounters.p + 1, h->dDx.p, h->dR.p, h->dZ.p, fd + F, fd + 2 * F, tol2, rc,
                     h->coarse.modeActive.p, h->hPcg, csOff, dsOff, static_cast<const double*>(nullptr), 0, F,
                     static_cast<double*>(nullptr), nTlWg ? temporalStepDev(h) : static_cast<const TlStep*>(nullptr),
                     static_cast<const TlStep*>(nullptr));
  // (adds the level's part of r^T z before the pose-graph level's kernel closes the scalars -- or closes them itself)
  if (nTlWg) launchTemporalInit(c, !coarse, tol2);
  if (coarse) coarseApply(1);
  HIP_CHECK(hipGetLastError());
  double* pOld = h->dP0.p;
  double* pNew = h->dP1.p;
  // finish + update as ONE launch per iteration where the fused kernel's scope allows (k_pcg_tail, cvd_matvec.hip)
  size_t ldsTail = 0;
  int ldsFinish = 0, ldsScratch = 0;
  const bool fusedTail = pcgTailScope(c, coarse, nThreads, ldsTail, ldsFinish, ldsScratch);
  h->lastFusedTail = fusedTail;
  h->lastKD = c.KD;
  h->lastCross = c.cross;
  if (fusedTail) {  // (its grid barrier's words)
    h->dTailBar.ensure(static_cast<size_t>(kTailBarStride) * (1 + kTailBarCopies));
    HIP_CHECK(hipMemsetAsync(h->dTailBar.p, 0, static_cast<size_t>(kTailBarStride) * (1 + kTailBarCopies) * sizeof(unsigned int), s));
    // (the temporal level's node-sum records carry the barrier's generation, which restarts with its words)
    if (tsOn.rec != nullptr) HIP_CHECK(hipMemsetAsync(tsOn.rec, 0, 2 * static_cast<size_t>(tsOn.NT) * sizeof(double), s));
  }
  const bool ownerMode = ownerShardedUpdate(h, coarse);
  const int maxIt = std::max(1, c.h->opt.pcg_max_iterations);
  // Convergence is decided on the device (S_DONE, set by the last workgroup of k_cg_update); how far the host runs ahead of it
  // is described at the loop below.
  // (profiling aid: cvd_solver_options::pcg_lockstep checks after every iteration and never runs ahead, so that per-launch
  // counter averages contain no early-exit launches)
  const bool lockstep = h->dbg.pcg_lockstep != 0;
  const size_t firstTimerSlot = h->evUsed;
  int enq = 0;
  auto enqueueIteration = [&](int it, int useBeta) {
    h->curPcgIter = it;
    if (fusedTail) {
      launchMatvec(c, x, h->dZ.p, pOld, pNew, useBeta, h->dLam.p, h->dQ.p, coarse, true);
      launchPcgTail(c, x, pOld, pNew, useBeta, h->dLam.p, h->dQ.p, coarse, nThreads, ldsTail, ldsFinish, ldsScratch, tol2);
      std::swap(pOld, pNew);
      return;
    }
    launchMatvec(c, x, h->dZ.p, pOld, pNew, useBeta, h->dLam.p, h->dQ.p, coarse);
    if (ownerMode) {
      // ---- owner-sharded iteration: the exchange above left the reduced q of THIS rank's frames (and [Z^T q | p.q] everywhere);
      // update the own frames, gather z / c / the r^T z shares, finish the scalars
      const int f0 = h->ownFirst(), nOwn = h->ownCount();
      const int slotO = h->tBegin(KC_CG_UPDATE);
      if (nOwn > 0 || nTlWg || nPtWg)  // (a rank without frames still walks the temporal levels' rows: every rank keeps its own copy of t / tl)
        hipLaunchKernelGGL(k_cg_update, dim3((denseRows && dsOn.rowSplit > 0 ? nOwn + (nOwn + kDenseFramesPerGroup - 1) / kDenseFramesPerGroup : nOwn) + nTlWg + nPtWg),
                           dim3(nThreads), ldsU, s, c.L, 0, h->dG.p, h->dMinv.p, pNew, h->dQ.p, h->dScal.p, h->dCounters.p + 1,
                           h->dDx.p, h->dR.p, h->dZ.p, fd + F, fd + 2 * F, tol2, static_cast<double*>(nullptr),
                           h->coarse.modeActive.p, h->hPcg, csOff, dsOn, pqReduced, f0, nOwn, h->dOwnerScal.p + 2 * h->rank,
                           nTlWg ? temporalStepDev(h) : static_cast<const TlStep*>(nullptr), tpDev);
      else
        HIP_CHECK(hipMemsetAsync(h->dOwnerScal.p + 2 * h->rank, 0, 2 * sizeof(double), s));
      HIP_CHECK(hipGetLastError());
      h->tEnd(slotO);
      const int ct = h->tBegin(KC_COMM_PRODUCT);
      const size_t chunkF = static_cast<size_t>(h->ownChunk());
      commGroupStart(h);
      commAllGather(h, h->dZ.p + h->rank * chunkF * B, h->dZ.p, chunkF * B, CT_F64, s);
      if (denseRows) commAllGather(h, h->coarse.c.p + h->rank * chunkF * kCB, h->coarse.c.p, chunkF * kCB, CT_F64, s);
      commAllGather(h, h->dOwnerScal.p + 2 * h->rank, h->dOwnerScal.p, 2, CT_F64, s);
      commGroupEnd(h);
      h->tEnd(ct);
      hipLaunchKernelGGL(k_pcg_scalars_dist, dim3(1), dim3(64), 0, s, h->dOwnerScal.p, h->world, h->dScal.p, tol2, h->hPcg);
      HIP_CHECK(hipGetLastError());
      std::swap(pOld, pNew);
      return;
    }
    const int slot = h->tBegin(KC_CG_UPDATE);
    hipLaunchKernelGGL(k_cg_update, dim3((denseRows && dsOn.rowSplit > 0 ? F + (F + kDenseFramesPerGroup - 1) / kDenseFramesPerGroup : F) + nTlWg + nPtWg), dim3(nThreads), ldsU, s, c.L, 0, h->dG.p, h->dMinv.p, pNew,
                       h->dQ.p, h->dScal.p, h->dCounters.p + 1, h->dDx.p, h->dR.p, h->dZ.p, fd + F, fd + 2 * F, tol2,
                       (coarse && unfusedY && !denseFused) ? rc : nullptr, h->coarse.modeActive.p, h->hPcg, csOn, dsOn, pqReduced,
                       0, F, static_cast<double*>(nullptr), nTlWg ? temporalStepDev(h) : static_cast<const TlStep*>(nullptr), tpDev);
    if (coarse && !denseFused) { if (unfusedY) coarseApply(0); else coarseC(0); }
    HIP_CHECK(hipGetLastError());
    h->tEnd(slot);
    std::swap(pOld, pNew);
  };
  // (Replaying the batch as a hipGraph was tried: the ~6 us between the five dependent launches of an iteration are
  // device-side dependency resolution, not host launch latency -- no gain, removed.)
  // Convergence is decided on the device (S_DONE); the last workgroup of every iteration also mirrors its progress
  // into pinned host memory (pcgFinishScalars).  Iteration k is enqueued once the done flag after exactly
  // k - kRunAhead + 1 iterations is known to be clear: nothing but the PCG kernels is in the stream, the device is
  // never starved (kRunAhead iterations are queued ahead) and kRunAhead - 1 early-exit iterations are wasted per
  // solve.  The rule is a function of iteration counts only, hence identical on all ranks of a sharded run.
  const int kRunAhead = lockstep ? 1 : 2;  // (3-4 iterations ahead: no change, the host is not late)
  volatile double* prog = h->hPcg;
  while (enq < maxIt) {
    if (enq >= kRunAhead - 1) {
      const int need = enq - kRunAhead + 1;
      // slot 1 + (need & 7) = 4 (need + 1) + done flag once exactly `need` iterations have been applied
      double v;
      for (unsigned long long spins = 0; static_cast<long long>((v = prog[1 + (need & 7)]) * 0.25) != need + 1; ++spins) {
        if ((spins & 0xFFFFF) == 0xFFFFF) {
          // never spin forever on a mirror that cannot advance: a faulted stream reports here, and an idle stream
          // whose iterations did not publish progress is a logic error
          const hipError_t e = hipStreamQuery(s);
          if (e != hipErrorNotReady) {
            HIP_CHECK(e);
            if (static_cast<long long>(prog[1 + (need & 7)] * 0.25) != need + 1) {
              if (fusedTail) throw TailStalled{};
              throw std::runtime_error("PCG progress mirror stalled");
            }
          }
        }
      }
      if (v - 4.0 * (need + 1) != 0.0) break;
    }
    enqueueIteration(enq, enq > 0 ? 1 : 0);
    ++enq;
    // (test hook, cvd_debug_options::stall_fused_tail_once: the host behaves as if the fused tail's barrier had been abandoned in the middle of the
    // first solve it is used in -- exercises the recovery path above without needing a second tenant on the device)
    if (fusedTail && h->dbg.stall_fused_tail_once != 0 && enq == 3) throw TailStalled{};
    if (h->opt.verbose >= 2) {  // development trace: per-iteration scalars (synchronises every iteration)
      readScalars(c);
      std::printf("    pcg %3d  rz %.6e  rzpart %.6e  alpha %.6e  beta %.6e  pq %.6e  done %g\n", enq - 1, h->hScal[S_RZ],
                  h->hScal[S_RZPART], h->hScal[S_ALPHA], h->hScal[S_BETA], h->hScal[S_PQ], h->hScal[S_DONE]);
    }
  }
  h->curPcgIter = -1;
  if (ownerMode) {
    // the step and the residual of the frames live on their owners: everybody needs both (step statistics, candidate point)
    const int ct = h->tBegin(KC_COMM_PRODUCT);
    const size_t chunk = static_cast<size_t>(h->ownChunk()) * B;
    commGroupStart(h);
    commAllGather(h, h->dDx.p + h->rank * chunk, h->dDx.p, chunk, CT_F64, s);
    commAllGather(h, h->dR.p + h->rank * chunk, h->dR.p, chunk, CT_F64, s);
    commGroupEnd(h);
    h->tEnd(ct);
  }
  if (tail) tail();  // follow-up work that does not need the host's decision rides on the same read-back
  readScalars(c);    // drains the stream; S_DONE / S_ITERS are final
  if (h->hScal[S_DONE] == 2.0) throw std::runtime_error("PCG produced NaN");
  const int iters = static_cast<int>(h->hScal[S_ITERS]);
  // The run-ahead check above never looks at the last kRunAhead - 1 enqueued iterations: an abandon there shows as iterations that
  // were enqueued before convergence and never applied.
  if (fusedTail && h->hScal[S_DONE] == 0.0 && iters != enq) throw TailStalled{};
  h->tDropFrom(firstTimerSlot, iters);
  return iters;
}

// Scene-flow smoothness triplets of this problem (reference lib/PoseOptimizer.cpp:899): on when a weight is > 0.
bool wantsTriplets(const cvd_opt_params& p, ProblemKind kind) {
  return kind == PK_POSE_STEP && (p.smooth_static_weight > 0.0 || p.smooth_dynamic_weight > 0.0);
}
void bindTriplets(Ctx& c, const cvd_opt_params& p, ProblemKind kind) {
  cvd_handle* h = c.h;
  c.trip = wantsTriplets(p, kind) && c.L.includeStatic;
  if (!c.trip) return;
  c.TT = TripletTable{h->dTNdc.p, h->dTDsrc.p, h->dTStatic.p, h->dTOff.p, h->dTCenter.p, h->dTSlot.p,
                      static_cast<int>(h->tripActive.size()),
                      static_cast<int>(p.smooth_loss_type),  // (CVD_SMOOTH_* == kSmooth*)
                      std::sqrt(std::max(0.0, p.smooth_static_weight)), std::sqrt(std::max(0.0, p.smooth_dynamic_weight))};
}
void solve(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, ProblemKind kind) {
  const double t0 = nowSeconds();
  if (h->F <= 0) throw std::runtime_error("no video set");
  if (!h->poseParamsValid) posesToParams(h);
  const std::vector<int> range = rangeOf(p, h->F);
  // A coarse rebuild still running on the side stream when the solve ends (or throws) must finish before anything on the
  // main stream -- the next solve's in-line setup, a table upload -- touches the coarse buffers again.
  struct PendingGuard {
    cvd_handle* h;
    bool pending = false;
    bool inverseAside = false;   // the block inverses were forked to their own stream and the solve left before the join
    ~PendingGuard() {
      if (pending) (void)hipStreamWaitEvent(h->stream, h->evCoarseDone, 0);
      if (inverseAside) (void)hipStreamWaitEvent(h->stream, h->evInvDone, 0);
      if (h->temporal.sidePending) {  // (the third level's assembly was forked and the solve left before its join)
        (void)hipStreamWaitEvent(h->stream, h->temporal.evDone, 0);
        h->temporal.sidePending = false;
      }
    }
  } pendingGuard{h};
  const bool dbgSetup = h->opt.verbose >= 3;  // development: where a solve's fixed cost goes
  double tPhase = nowSeconds();
  auto phase = [&](const char* what) {
    if (!dbgSetup) return;
    const double t = nowSeconds();
    fprintf(stderr, "[setup] %-14s %8.1f us\n", what, (t - tPhase) * 1e6);
    tPhase = t;
  };
  Ctx c;
  c.h = h;
  c.L = makeLayout(h, p, depthDeformReg, kind);
  checkFrameBlock(static_cast<size_t>(c.L.B), "solve");
  tapCounts(c.L, c.KD, c.KS);
  phase("layout");
  // dense mode outside the fast kernels' scope: this solve runs on the device-materialised list (flipped back when the solve ends)
  DenseListScope denseList(h, h->dense && c.L.includeStatic && !denseFastScope(h, c.L, c.KS, wantsTriplets(p, kind)));
  compileTable(h, range, wantsTriplets(p, kind), kind == PK_NORMALIZE && c.L.includeStatic);
  orderTable(h, c.L, c.KD);
  phase("table");
  refreshMedians(h);
  phase("medians");
  c.T = makeTable(h);
  c.nItems = static_cast<int>(h->itemFa.size());
  c.it = Items{h->dItemFa.p, h->dItemFb.p, h->dItemRange.p, h->dItemSlot.p, c.nItems};
  c.n = static_cast<size_t>(c.L.F) * c.L.B;
  c.boundDepth0 = (kind == PK_NORMALIZE && c.L.N > 0) ? 1 : 0;
  bindTriplets(c, p, kind);
  if (c.L.includeStatic) checkDenseScope(h, c.L, c.KS, c.trip);
  c.cross = crossScope(h, c);
  h->coarseOn = h->opt.coarse_level != 0 && h->coarse.valid && c.L.includeStatic && h->coarse.nEdges > 0 && !h->forceGeneric &&
                kind == PK_POSE_STEP;  // (normalizeDepth's problems have no pose unknowns: the block-Jacobi level alone)
  ensureBuffers(c);
  phase("buffers");
  h->coarse.ptInvPending = false;   // (a request left behind by a solve that threw)
  h->temporal.on = temporalScope(c);
  if (h->temporal.on) h->temporal.on = temporalPrepare(c);
  phase("level 3 tables");
  if (h->coarseOn && h->coarse.temporalPose) poseTemporalPrepare(c);
  phase("level 2 tables");
  buildMask(h, c.L, p, kind, range);
  phase("mask");
  uploadState(h, c.L, h->dX);
  phase("upload");
  hipStream_t s = h->stream;
  HIP_CHECK(hipMemsetAsync(h->dDx.p, 0, c.n * sizeof(double), s));
  HIP_CHECK(hipMemsetAsync(h->dR.p, 0, c.n * sizeof(double), s));
  HIP_CHECK(hipMemsetAsync(h->dFail.p, 0, sizeof(int), s));

  cvd_solve_summary sum{};
  long long regBlocks = 0;
  {
    const long long nr = static_cast<long long>(range.size());
    if (c.L.scaleRegSqrt > 0.0) regBlocks += nr * c.L.sregX * c.L.sregY;
    if (c.L.focalRegSqrt > 0.0) regBlocks += nr;
    if (c.L.depthDeformW > 0.0 && c.L.depthType == CVD_DEPTH_GRID) regBlocks += nr;
    if (c.L.spatialDeformW > 0.0 && c.L.nS > 0) regBlocks += nr;
    if (c.L.positionRegSqrt > 0.0)
      for (int k = c.L.firstFrame; k < c.L.lastFrame - 1; ++k)
        regBlocks += (h->tableRange[k] && h->tableRange[k + 1] && h->tableRange[k + 2]) ? 1 : 0;
  }
  sum.num_residual_blocks = static_cast<int>((c.L.includeStatic ? h->numValid : 0) + (c.trip ? h->numValidTrip : 0) + regBlocks);

  double tEval = 0.0, tLin = 0.0;
  double te = nowSeconds();
  // first evaluation: cost, |g|_max, |x| and the number of active unknowns arrive with ONE read-back (until round 5: the cost, then the
  // statistics, then a downloaded copy of diag(H) counted on the host -- three round trips of 25 - 55 us at the start of every solve)
  double xCost = evalFull(c, h->dX.p, true, false);
  tEval += nowSeconds() - te;
  sum.initial_cost = xCost;
  phase("first eval");
  double gmax = h->hScal[S_GMAX];
  double xNorm = std::sqrt(h->hScal[S_XX]);
  sum.num_parameters = static_cast<int>(h->hScal[S_NACTIVE] + 0.5);
  phase("stats");

  double radius = Ceres::initial_radius;
  double decrease = 2.0;
  // After an accepted step the host does not wait for the new point's |g|_max / |x| (one idle round trip of ~60 us per LM
  // iteration, round 5): the step statistics of the NEXT PCG solve recompute both for the same point (k_step_stats) and arrive
  // with its read-back.  gradPending: the gradient-tolerance test of the last accepted step and its record's gradient norm are
  // still open.  (verbose runs keep the immediate read: the table prints the norm with its iteration)
  bool gradPending = false;
  const bool deferStats = h->opt.verbose == 0;
  int invalid = 0, iteration = 0, termination = 1;
  // Rebuild threshold of the coarse level in PCG iterations.  Sparse factor (side stream): the option.  DENSE level, built in
  // line: coarse_rebuild_excess_dense > 0 fixes it; 0 (default) = 32, a constant: identical inputs must take identical
  // rebuild decisions on every run and every rank (the wall-clock-derived threshold made PCG counts and end states
  // depend on host jitter).  -1 opts into pricing a rebuild at what THIS handle measured -- 1.5 x (the rebuild's duration / the
  // duration of a PCG iteration of the running solves), in steps of 8 (the PCG counts of an LM run grow by themselves as the
  // trust region opens: a threshold equal to the bare cost ratio, 22 at 300 frames, rebuilt every second LM iteration to save
  // 0.6 iterations per LM iteration; 1.5 x = 32 there).  Sharded runs ignore -1: every rank must decide alike.
  auto denseRebuildThreshold = [&]() {
    if (h->opt.coarse_rebuild_excess_dense > 0) return h->opt.coarse_rebuild_excess_dense;
    if (h->opt.coarse_rebuild_excess_dense == 0 || h->dist() || h->coarseRebuildMs <= 0.0 || h->pcgIterMs <= 0.0) return 32;
    const double ratio = 1.5 * h->coarseRebuildMs / h->pcgIterMs;
    return std::min(256, std::max(8, 8 * static_cast<int>(ratio / 8.0 + 0.5)));
  };
  const int kCoarseRebuildIters = std::max(0, h->coarse.denseMode ? denseRebuildThreshold() : h->opt.coarse_rebuild_excess);
  int coarseAge = -1, cgAfterRefresh = 0, cgExcess = 0;  // coarse level: LM iterations since the last rebuild
  bool& coarsePending = pendingGuard.pending;  // a rebuild is running on the side stream
  int factorUses = 0;          // PCG solves done with the factor in use
  double lastRelChange = 1.0;  // relative cost change of the last accepted step
  constexpr double asyncMaxChange = 1e-3;
  bool freshFactor = false;    // the factor was installed right before this iteration's PCG
  auto installPendingCoarse = [&]() {
    if (!coarsePending) return;
    HIP_CHECK(hipStreamWaitEvent(s, h->evCoarseDone, 0));  // (device-side wait: the host does not block)
    std::swap(h->coarse.Wb.p, h->coarse.Wb2.p);
    std::swap(h->coarse.Wb.n, h->coarse.Wb2.n);
    std::swap(h->coarse.fail.p, h->coarse.fail2.p);
    std::swap(h->coarse.fail.n, h->coarse.fail2.n);
    coarsePending = false;
    freshFactor = true;
    cgExcess = 0;
    coarseAge = 0;
    factorUses = 0;
  };
  bool scaleDone = false;
  cvd_iteration_record r0{};
  r0.cost = xCost;
  r0.gradient_max_norm = gmax;
  r0.trust_region_radius = radius;
  r0.step_is_successful = 1;
  h->records.push_back(r0);
  if (h->opt.verbose)
    printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter\n"
           "%4d % .6e    % .2e   % .2e   % .2e  % .2e % .2e   %5d\n", 0, xCost, 0.0, gmax, 0.0, 0.0, radius, 0);

  if (sum.num_parameters == 0 || (gmax <= Ceres::gradient_tolerance && !h->dbg.force_iterations)) {
    termination = 0;
  } else {
    while (true) {
      if (iteration >= p.max_iterations) { termination = 1; break; }
      if (radius < Ceres::min_radius) { termination = 0; break; }
      ++iteration;
      cvd_iteration_record rec{};
      rec.iteration = iteration;

      double tl = nowSeconds();
      hipLaunchKernelGGL(k_lm_diag, dim3((c.n + 255) / 256), dim3(256), 0, s, c.L, h->dHd.p, h->dScale.p,
                         scaleDone ? 0 : 1, radius, h->dLam.p);
      scaleDone = true;
      // The block-Jacobi level follows lam every LM iteration; the coarse level is rebuilt on demand (below).
      // (Lagging the block inverse as well is ~4% faster on the benchmark but makes the converged parameters
      // visibly sensitive to rounding noise along the weak gauge directions.)
      const bool willRefresh = !coarsePending &&
                               (!h->coarseOn || h->opt.coarse_level == 2 || coarseAge < 0 || cgExcess >= kCoarseRebuildIters);
      // This iteration builds the levels in line (the first iteration of a solve, or a rebuild while the iterates still move): the
      // frames' block inverses -- one workgroup per frame, a latency-bound sweep of 0.12 - 0.17 ms that leaves most of every CU
      // idle, and that only the PCG reads -- run on a stream of their own BESIDE the build instead of in front of it.
      // (a rebuild on the side stream, for the NEXT iteration: only in the slowly changing regime, see below)
      const bool asyncRebuild = lastRelChange < asyncMaxChange && !h->coarse.denseMode && h->opt.coarse_level != 2 && !h->dist();
      const bool inlineRebuild = h->coarseOn && willRefresh && !asyncRebuild;
      const bool inverseAside = inlineRebuild && !h->dist() && h->stream3 != nullptr && c.L.B <= 256;
      if (inverseAside) {
        HIP_CHECK(hipEventRecord(h->evInvIn, s));
        HIP_CHECK(hipStreamWaitEvent(h->stream3, h->evInvIn, 0));
        launchBlockInverse(c, h->stream3);
        HIP_CHECK(hipEventRecord(h->evInvDone, h->stream3));
        pendingGuard.inverseAside = true;
      } else {
        const int slot = h->tBegin(KC_INVERSE);
        launchBlockInverse(c);
        h->tEnd(slot);
      }
      if (h->coarseOn) {
        // The coarse factor is only a preconditioner: any SPD approximation of Z^T A Z serves, so it is kept
        // across LM iterations (lagged lam and linearisation point).  A rebuild costs about as much as
        // kCoarseRebuildIters PCG iterations; it is done once the iterations spent beyond the count observed
        // right after the last rebuild add up to that (coarse_level 2: rebuild every LM iteration).
        // When a factor already exists the rebuild runs on the side stream, concurrently with this iteration's PCG
        // (which keeps the old factor), and is installed for the next iteration: its ~0.8 ms leave the critical path.
        // Its inputs (H, lam, x, mask, the table) are not written before the install below; the frame constants,
        // which the main stream rewrites for the candidate point, are private to the side stream.
        if (willRefresh) {
          // (Only in the slowly changing regime -- the last accepted step changed the cost by less than 0.1 % --: while
          // the iterates still move a lot a factor that is one iteration late costs more PCG iterations than the
          // overlap saves, and there the rebuild stays in line.)
          // (the dense level's inverse is one persistent kernel that wants every CU: always in line)
          if (asyncRebuild) {
            h->dFc2.ensure(c.L.F);
            //  the third level is "rebuilt together with the pose-graph level": also when that rebuild runs on the side
            // stream -- at this linearisation point, for THIS iteration's PCG (it is small: two launches and a 0.1 ms inverse).
            //  its assembly goes to the side stream BEFORE the pose-graph level's rebuild: queued behind it, the main
            // stream's wait for the third level would have been a wait for the whole rebuild the side stream exists to hide.
            // Its inverse (a persistent kernel: gated per device) is enqueued before the rebuild's factorisation as well, so that the
            // gate's event chain makes the SIDE stream wait for the small inverse and not the solver's stream for the rebuild; the two
            // levels' inverses have scratch (panel, barrier words) of their own.
            if (h->temporal.on) {
              const int slotT = h->tBegin(KC_INVERSE);
              launchTemporalSetup(c, h->dX.p, 0);
              launchTemporalSetup(c, h->dX.p, 1);
              h->tEnd(slotT);
            }
            HIP_CHECK(hipEventRecord(h->evCoarseIn, s));
            HIP_CHECK(hipStreamWaitEvent(h->stream2, h->evCoarseIn, 0));
            launchCoarseSetup(c, h->dX.p, 1);
            HIP_CHECK(hipEventRecord(h->evCoarseDone, h->stream2));
            coarsePending = true;
            cgExcess = 0;
          } else {
            const int slot = h->tBegin(KC_INVERSE);  // preconditioner construction, same class as the block inverse
            const bool measure = h->coarse.denseMode && !h->dist() && h->opt.coarse_rebuild_excess_dense < 0;
            if (measure) {
              if (!h->evRebuild[0]) for (auto& e : h->evRebuild) HIP_CHECK(hipEventCreate(&e));
              HIP_CHECK(hipEventRecord(h->evRebuild[0], s));
            }
            // (third level: same linearisation point and damping; its assembly runs beside the pose-graph level's build)
            if (h->temporal.on) launchTemporalSetup(c, h->dX.p, 0);
            launchCoarseSetup(c, h->dX.p);
            if (inverseAside) {  // (joined before the temporal levels' persistent inverse pair: that kernel keeps the device to itself)
              HIP_CHECK(hipStreamWaitEvent(s, h->evInvDone, 0));
              pendingGuard.inverseAside = false;
            }
            if (h->temporal.on) launchTemporalSetup(c, h->dX.p, 1);
            if (measure) {
              HIP_CHECK(hipEventRecord(h->evRebuild[1], s));
              h->rebuildTimed = true;
            }
            h->tEnd(slot);
            coarseAge = 0;
            cgExcess = 0;
            freshFactor = true;
            factorUses = 0;
          }
        } else {
          ++coarseAge;
          if (h->temporal.on && h->opt.temporal_level == 2) {
            const int slot = h->tBegin(KC_INVERSE);
            launchTemporalSetup(c, h->dX.p, 0);
            launchTemporalSetup(c, h->dX.p, 1);
            h->tEnd(slot);
          }
        }
      }
      if (!h->coarseOn && h->temporal.on) {
        // third level without a pose-graph level (coarse_level 0): rebuilt by the same rule -- at the first iteration, then once
        // the PCG iterations spent beyond the count seen right after the last build add up to the threshold (level 2: every iteration)
        if (coarseAge < 0 || h->opt.temporal_level == 2 || cgExcess >= kCoarseRebuildIters) {
          const int slot = h->tBegin(KC_INVERSE);
          launchTemporalSetup(c, h->dX.p, 0);
          launchTemporalSetup(c, h->dX.p, 1);
          h->tEnd(slot);
          coarseAge = 0;
          cgExcess = 0;
          freshFactor = true;
        } else {
          ++coarseAge;
        }
      }
      // one read-back for the PCG result, the step statistics and the cost of the candidate point (the
      // candidate is formed speculatively; it is simply not used when the model decrease is invalid)
      const double tPcgStart = nowSeconds();
      const int cgIters = runPcg(c, h->dX.p, [&]() {
        enqueueStats(c);
        hipLaunchKernelGGL(k_apply_step, dim3((c.n + 255) / 256), dim3(256), 0, s, c.L, c.boundDepth0, h->dX.p,
                           h->dDx.p, h->dXc.p);
        HIP_CHECK(hipGetLastError());
        enqueueCost(c, h->dXc.p);
      });
      {
        const double tPcg = nowSeconds() - tPcgStart;
        if (cgIters > 0 && !h->dist()) {
          double ms = tPcg * 1e3;
          if (h->rebuildTimed) {  // (the rebuild ran in line right before this PCG: its time is not the iterations')
            float rb = 0.f;
            if (hipEventElapsedTime(&rb, h->evRebuild[0], h->evRebuild[1]) == hipSuccess && rb > 0.f) {
              h->coarseRebuildMs = h->coarseRebuildMs > 0.0 ? 0.5 * (h->coarseRebuildMs + rb) : rb;
              ms = std::max(0.0, ms - rb);
            }
            h->rebuildTimed = false;
          }
          const double per = ms / cgIters;
          h->pcgIterMs = h->pcgIterMs > 0.0 ? 0.75 * h->pcgIterMs + 0.25 * per : per;
        }
      }
      if (freshFactor) cgAfterRefresh = cgIters;
      else cgExcess += std::max(0, cgIters - cgAfterRefresh);
      freshFactor = false;
      ++factorUses;
      installPendingCoarse();
      tLin += nowSeconds() - tl;
      if (gradPending) {
        gradPending = false;
        gmax = h->hScal[S_GMAX];
        xNorm = std::sqrt(h->hScal[S_XX]);
        h->records.back().gradient_max_norm = gmax;
        if (gmax <= Ceres::gradient_tolerance && !h->dbg.force_iterations) {
          // the previous iteration ended the solve (Ceres tests the gradient right after a successful step): this iteration's
          // linear solve was speculative and is dropped -- x, the cost and the records are those of the previous iteration
          --iteration;
          termination = 0;
          break;
        }
      }
      rec.linear_iterations = cgIters;
      sum.total_linear_iterations += cgIters;
      const double dg = h->hScal[S_DG], dr = h->hScal[S_DR], dld = h->hScal[S_DLD], dd = h->hScal[S_DD];
      const double modelCostChange = -0.5 * dg + 0.5 * dr + 0.5 * dld;
      bool ok = std::isfinite(modelCostChange) && modelCostChange > 0.0 && std::isfinite(dd);
      if (!ok) {
        if (++invalid >= Ceres::max_consecutive_invalid) { termination = 2; break; }
        radius /= decrease;
        decrease *= 2.0;
        rec.cost = xCost;
        rec.trust_region_radius = radius;
        h->records.push_back(rec);
        continue;
      }
      invalid = 0;
      double candCost = h->hScal[S_COST];
      if (!std::isfinite(candCost)) candCost = std::numeric_limits<double>::max();
      const double stepNorm = std::sqrt(dd);
      rec.step_norm = stepNorm;
      rec.cost_change = xCost - candCost;
      rec.relative_decrease = (xCost - candCost) / modelCostChange;
      bool stop = false;
      if (stepNorm <= Ceres::parameter_tolerance * (xNorm + Ceres::parameter_tolerance)) stop = true;
      if (!stop && std::abs(xCost - candCost) <= Ceres::function_tolerance * xCost) stop = true;
      if (h->dbg.force_iterations) stop = false;
      if (stop) {
        rec.cost = xCost;
        rec.trust_region_radius = radius;
        h->records.push_back(rec);
        if (h->opt.verbose)
          printf("%4d % .6e    % .2e   % .2e   % .2e  % .2e % .2e   %5d\n", iteration, xCost, rec.cost_change, gmax,
                 stepNorm, rec.relative_decrease, radius, cgIters);
        termination = 0;
        break;
      }
      if (rec.relative_decrease > Ceres::min_relative_decrease) {
        std::swap(h->dX.p, h->dXc.p);
        std::swap(h->dX.n, h->dXc.n);
        lastRelChange = std::abs(xCost - candCost) / std::max(std::abs(xCost), 1e-300);
        xCost = candCost;
        te = nowSeconds();
        // (the last iteration allowed reads at once: nothing follows that would bring the statistics)
        const bool defer = deferStats && iteration < p.max_iterations;
        (void)evalFull(c, h->dX.p, true, defer);
        tEval += nowSeconds() - te;
        if (defer) {
          gradPending = true;   // (gmax / xNorm keep the previous point's values until the next solve's read-back)
        } else {
          gmax = h->hScal[S_GMAX];
          xNorm = std::sqrt(h->hScal[S_XX]);
        }
        ++sum.num_successful_steps;
        rec.step_is_successful = 1;
        const double q = rec.relative_decrease;
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * q - 1.0, 3));
        radius = std::min(Ceres::max_radius, radius);
        decrease = 2.0;
        rec.cost = xCost;
        rec.gradient_max_norm = gmax;
        rec.trust_region_radius = radius;
        h->records.push_back(rec);
        if (h->opt.verbose)
          printf("%4d % .6e    % .2e   % .2e   % .2e  % .2e % .2e   %5d\n", iteration, xCost, rec.cost_change, gmax,
                 stepNorm, rec.relative_decrease, radius, cgIters);
        if (!gradPending && gmax <= Ceres::gradient_tolerance && !h->dbg.force_iterations) { termination = 0; break; }
      } else {
        radius /= decrease;
        decrease *= 2.0;
        rec.cost = xCost;
        rec.trust_region_radius = radius;
        h->records.push_back(rec);
        if (h->opt.verbose)
          printf("%4d % .6e    % .2e   % .2e   % .2e  % .2e % .2e   %5d\n", iteration, xCost, rec.cost_change, gmax,
                 stepNorm, rec.relative_decrease, radius, cgIters);
      }
    }
  }
  if (gradPending) {  // (the loop left before another solve brought the last accepted point's statistics)
    HIP_CHECK(hipMemsetAsync(h->dLam.p, 0, c.n * sizeof(double), s));
    enqueueStats(c);
    readScalars(c);
    h->records.back().gradient_max_norm = h->hScal[S_GMAX];
    gradPending = false;
  }
  phase("LM loop");
  // a barrier timeout of the dense coarse inverse (bit 30) only costs PCG iterations -- but it should never pass unnoticed: its
  // status word travels with the state (one host wait for both; a round trip of its own cost ~25 us per solve)
  int* failWord = reinterpret_cast<int*>(h->hScal + S_COUNT);
  *failWord = 0;
  if (h->coarseOn && h->coarse.denseMode && h->coarse.fail.p != nullptr && !h->dist())
    HIP_CHECK(hipMemcpyAsync(failWord, h->coarse.fail.p, sizeof(int), hipMemcpyDeviceToHost, s));
  downloadState(h, c.L, h->dX);
  if (*failWord & 0x40000000)
    fprintf(stderr, "[cvd] warning: the dense coarse inverse timed out at its grid barrier (device shared with other work?); "
                    "the coarse level was off for the last LM iteration(s)\n");
  phase("download");
  h->tCollect();
  sum.num_iterations = iteration;
  sum.termination = termination;
  sum.final_cost = xCost;
  sum.total_seconds = nowSeconds() - t0;
  sum.evaluate_seconds = tEval;
  sum.linear_solve_seconds = tLin;
  h->summary = sum;
}
void evaluate(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, const double* pose7,
                     double* cost, int32_t* nres, double* gradient, double* hdiag, double* hfull) {
  if (pose7) {
    h->poseParams.resize(h->F);
    for (int f = 0; f < h->F; ++f)
      for (int i = 0; i < 7; ++i) h->poseParams[f][i] = pose7[f * 7 + i];
    h->poseParamsValid = true;
  } else {
    posesToParams(h);
  }
  const std::vector<int> range = rangeOf(p, h->F);
  Ctx c;
  c.h = h;
  c.L = makeLayout(h, p, depthDeformReg, PK_POSE_STEP);
  tapCounts(c.L, c.KD, c.KS);
  DenseListScope denseList(h, h->dense && !denseFastScope(h, c.L, c.KS, wantsTriplets(p, PK_POSE_STEP)));
  compileTable(h, range, wantsTriplets(p, PK_POSE_STEP));
  orderTable(h, c.L, c.KD);
  refreshMedians(h);
  c.T = makeTable(h);
  c.nItems = static_cast<int>(h->itemFa.size());
  c.it = Items{h->dItemFa.p, h->dItemFb.p, h->dItemRange.p, h->dItemSlot.p, c.nItems};
  c.n = static_cast<size_t>(c.L.F) * c.L.B;
  bindTriplets(c, p, PK_POSE_STEP);
  checkDenseScope(h, c.L, c.KS, c.trip);
  c.cross = crossScope(h, c);
  h->coarseOn = false;
  ensureBuffers(c);
  buildMask(h, c.L, p, PK_POSE_STEP, range);
  uploadState(h, c.L, h->dX);
  hipStream_t s = h->stream;
  double cst;
  if (!gradient && !hdiag && !hfull) {
    cst = evalCost(c, h->dX.p);
  } else {
    cst = evalFull(c, h->dX.p);
  }
  if (cost) *cost = cst;
  if (nres) {
    long long regBlocks = 0;
    const long long nr = static_cast<long long>(range.size());
    if (c.L.scaleRegSqrt > 0.0) regBlocks += nr * c.L.sregX * c.L.sregY;
    if (c.L.focalRegSqrt > 0.0) regBlocks += nr;
    if (c.L.depthDeformW > 0.0 && c.L.depthType == CVD_DEPTH_GRID) regBlocks += nr;
    if (c.L.spatialDeformW > 0.0 && c.L.nS > 0) regBlocks += nr;
    if (c.L.positionRegSqrt > 0.0)
      for (int k = c.L.firstFrame; k < c.L.lastFrame - 1; ++k)
        regBlocks += (h->tableRange[k] && h->tableRange[k + 1] && h->tableRange[k + 2]) ? 1 : 0;
    *nres = static_cast<int32_t>(h->numValid + (c.trip ? h->numValidTrip : 0) + regBlocks);
  }
  if (gradient) h->dG.download(gradient, c.n, s);
  if (hdiag) {
    if (h->dist()) {  // (parity hook: collect the owners' reduced blocks)
      const size_t chunk = static_cast<size_t>(h->ownChunk()) * c.L.B * c.L.B;
      commAllGather(h, h->dH.p + static_cast<size_t>(h->rank) * chunk, h->dH.p, chunk, CT_F64, s);
    }
    h->dH.download(hdiag, c.n * c.L.B, s);
  }
  HIP_CHECK(hipStreamSynchronize(s));
  if (hfull) {
    // column j of J^T J = matvec with the unit vector e_j (lam = 0)
    std::vector<double> e(c.n, 0.0), col(c.n);
    HIP_CHECK(hipMemsetAsync(h->dLam.p, 0, c.n * sizeof(double), s));
    prepareMatvec(c, h->dX.p);
    for (size_t j = 0; j < c.n; ++j) {
      e[j] = 1.0;
      h->dZ.upload(e.data(), c.n, s);
      launchMatvec(c, h->dX.p, h->dZ.p, h->dP0.p, h->dP1.p, 0, h->dLam.p, h->dQ.p);
      h->dQ.download(col.data(), c.n, s);
      HIP_CHECK(hipStreamSynchronize(s));
      for (size_t i = 0; i < c.n; ++i) hfull[i * c.n + j] = col[i];
      e[j] = 0.0;
    }
  }
}

// One kernel of this translation unit's code object is looked up at handle creation: the HIP runtime loads a unit's device
// code at its first use, ~20 ms per unit that would otherwise land in the first solve of a process (cvd_create: loadDeviceCode).
void touchModule_solve() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_cg_update));
}

}  // namespace cvd
