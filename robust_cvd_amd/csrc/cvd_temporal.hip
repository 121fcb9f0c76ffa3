// cvd_temporal.hip -- host side of the third preconditioner level (cvd_temporal.h): scope, tables and work lists, the build of
// A_T^-1 per (H, lam, x), the first residual of a PCG solve.  The per-iteration part lives inside the PCG launches.
#include "cvd_host.h"

namespace cvd {

static int autoCoarse(int g, int wanted) {
  const int s = wanted > 0 ? wanted : (g + 1) / 2;
  return std::min(g, std::max(2, s));
}

// Static scope (see cvd_solver_options::temporal_level): everything the kernels of the level assume.
bool temporalScope(const Ctx& c) {
  cvd_handle* h = c.h;
  const Layout& L = c.L;
  // (dense mode: with the explicit cross blocks only -- the pair part is then a projection of the blocks; the matrix-free dense
  // product would need the per-pixel walk with 64 lanes hitting the same hats)
  if (h->opt.temporal_level == 0 || h->forceGeneric || (h->dense && !c.cross) || c.trip) return false;
  if (c.KD != 4 || c.KS != 0 || !fastLoss(L) || L.intrOpt == CVD_INTR_SHARED) return false;
  if (L.N != 1 || L.gz != 1 || L.depthType != CVD_DEPTH_GRID || L.positionRegSqrt > 0.0 || !L.includeStatic) return false;
  if (L.B > 256 || L.nD > 256 || L.gx < 3 || L.gy < 3 || L.nD != L.gx * L.gy) return false;
  const int step = h->opt.temporal_step;
  if (L.F <= 2 * step) return false;  // (fewer than three interior nodes: nothing temporal to coarsen)
  const int Sx = autoCoarse(L.gx, h->opt.temporal_grid_x), Sy = autoCoarse(L.gy, h->opt.temporal_grid_y);
  if (Sx * Sy > kTlMaxS) return false;
  const int nn = (L.F - 1 + step - 1) / step + 1;
  if (nn * Sx * Sy > kDenseCoarseMaxUnknowns) return false;
  return true;
}

// 1-D hats of one axis: fine vertex i sits at i / ratio in coarse units
static void axisTable(int g, int S, std::vector<float>& hv, std::vector<int>& bv) {
  hv.assign(2 * static_cast<size_t>(g), 0.f);
  bv.assign(g, 0);
  const double ratio = static_cast<double>(g - 1) / static_cast<double>(S - 1);  // fine cells per coarse cell (>= 1)
  for (int i = 0; i < g; ++i) {
    const double pos = static_cast<double>(i) / ratio;
    const int j0 = std::min(S - 2, std::max(0, static_cast<int>(std::floor(pos + 1e-12))));
    const double fr = std::min(1.0, std::max(0.0, pos - j0));
    bv[i] = j0;
    hv[2 * i] = static_cast<float>(1.0 - fr);
    hv[2 * i + 1] = static_cast<float>(fr);
  }
}

// false: the chosen coarse grid is outside what the kernels hold (a hat covering more than kTlMaxWidth depth vertices) -- the
// solve runs without the level
bool temporalPrepare(Ctx& c) {
  cvd_handle* h = c.h;
  auto& T = h->temporal;
  const Layout& L = c.L;
  hipStream_t s = h->stream;
  const int step = h->opt.temporal_step;
  const int Sx = autoCoarse(L.gx, h->opt.temporal_grid_x), Sy = autoCoarse(L.gy, h->opt.temporal_grid_y);
  const int S = Sx * Sy, G = L.nD, F = L.F;
  const int nn = (F - 1 + step - 1) / step + 1;
  bool uploaded = false;
  if (T.tabGx != L.gx || T.tabGy != L.gy || T.tabSx != Sx || T.tabSy != Sy) {
    std::vector<float> hx, hy;
    std::vector<int> bx, by;
    axisTable(L.gx, Sx, hx, bx);
    axisTable(L.gy, Sy, hy, by);
    std::vector<float4> vW(G);
    std::vector<unsigned int> vIdx(G);
    std::vector<std::vector<std::pair<int, float>>> cols(S);
    for (int vy = 0; vy < L.gy; ++vy)
      for (int vx = 0; vx < L.gx; ++vx) {
        const int v = vx + vy * L.gx;
        float w[4];
        unsigned int idx = 0;
        for (int k = 0; k < 4; ++k) {
          const int i = k & 1, j = k >> 1;
          const int hat = (bx[vx] + i) + (by[vy] + j) * Sx;
          w[k] = hx[2 * vx + i] * hy[2 * vy + j];
          idx |= static_cast<unsigned int>(hat) << (8 * k);
          if (w[k] != 0.f) cols[hat].emplace_back(v, w[k]);
        }
        vW[v] = make_float4(w[0], w[1], w[2], w[3]);
        vIdx[v] = idx;
      }
    size_t width = 1;
    for (auto& col : cols) width = std::max(width, col.size());
    if (width > static_cast<size_t>(kTlMaxWidth)) return false;
    std::vector<float> elW(width * S, 0.f);
    std::vector<unsigned char> elV(width * S, 0);
    for (int hat = 0; hat < S; ++hat)
      for (size_t k = 0; k < cols[hat].size(); ++k) {
        elW[k * S + hat] = cols[hat][k].second;
        elV[k * S + hat] = static_cast<unsigned char>(cols[hat][k].first);
      }
    T.hx.upload(hx.data(), hx.size(), s);
    T.hy.upload(hy.data(), hy.size(), s);
    T.bx.upload(bx.data(), bx.size(), s);
    T.by.upload(by.data(), by.size(), s);
    T.vW.upload(vW.data(), vW.size(), s);
    T.vIdx.upload(vIdx.data(), vIdx.size(), s);
    T.elW.upload(elW.data(), elW.size(), s);
    T.elV.upload(elV.data(), elV.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));  // (the host vectors go out of scope)
    T.tabGx = L.gx; T.tabGy = L.gy; T.tabSx = Sx; T.tabSy = Sy;
    T.width = static_cast<int>(width);
    uploaded = true;
  }
  T.S = S; T.Sx = Sx; T.Sy = Sy; T.nn = nn; T.step = step; T.NT = nn * S;
  // (the units of the pair part: the work items of the constraint walk, or -- explicit cross blocks -- the undirected pairs)
  const std::vector<int>& unitFa = c.cross ? h->xFa : h->itemFa;
  const std::vector<int>& unitFb = c.cross ? h->xFb : h->itemFb;
  if (uploaded || T.grpF != F || T.grpStep != step || T.grpFa != unitFa || T.grpFb != unitFb) {
    // groups of items by the pair of node intervals their frames fall into, <= kGroup items each
    constexpr size_t kGroup = 48;
    std::map<std::pair<int, int>, std::vector<int>> cells;
    for (size_t i = 0; i < unitFa.size(); ++i) cells[{unitFa[i] / step, unitFb[i] / step}].push_back(static_cast<int>(i));
    std::vector<int> gOff{0}, gItems;
    std::map<std::pair<int, int>, std::vector<int>> blocks;  // (a, b), a <= b -> gather entries
    for (int a = 0; a < nn; ++a) {
      blocks[{a, a}];
      if (a + 1 < nn) blocks[{a, a + 1}];
    }
    for (auto& cell : cells) {
      const auto& items = cell.second;
      for (size_t o = 0; o < items.size(); o += kGroup) {
        const int g = static_cast<int>(gOff.size()) - 1;
        gItems.insert(gItems.end(), items.begin() + o, items.begin() + std::min(items.size(), o + kGroup));
        gOff.push_back(static_cast<int>(gItems.size()));
        for (int k = 0; k < 4; ++k) {
          const int na = cell.first.first + (k >> 1), nb = cell.first.second + (k & 1);
          if (na >= nn || nb >= nn) continue;  // (its temporal weight is zero)
          const int en = 2 * (g * 4 + k);
          if (na < nb) blocks[{na, nb}].push_back(en);
          else if (na > nb) blocks[{nb, na}].push_back(en + 1);
          else { blocks[{na, na}].push_back(en); blocks[{na, na}].push_back(en + 1); }
        }
      }
    }
    std::vector<int> blkA, blkB, gPtr{0}, gather;
    for (auto& b : blocks) {
      blkA.push_back(b.first.first);
      blkB.push_back(b.first.second);
      gather.insert(gather.end(), b.second.begin(), b.second.end());
      gPtr.push_back(static_cast<int>(gather.size()));
    }
    if (gather.empty()) gather.push_back(0);
    if (gItems.empty()) gItems.push_back(0);
    T.gOff.upload(gOff.data(), gOff.size(), s);
    T.gItems.upload(gItems.data(), gItems.size(), s);
    T.blkA.upload(blkA.data(), blkA.size(), s);
    T.blkB.upload(blkB.data(), blkB.size(), s);
    T.gPtr.upload(gPtr.data(), gPtr.size(), s);
    T.gather.upload(gather.data(), gather.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    T.nGroups = static_cast<int>(gOff.size()) - 1;
    T.nBlocks = static_cast<int>(blkA.size());
    T.grpF = F; T.grpStep = step; T.grpFa = unitFa; T.grpFb = unitFb;
  }
  const size_t SS = static_cast<size_t>(S) * S, NT = static_cast<size_t>(T.NT);
  T.Cf.ensure(static_cast<size_t>(F) * SS);
  T.E.ensure(std::max<size_t>(1, unitFa.size()) * SS);
  T.part.ensure(std::max<size_t>(1, static_cast<size_t>(T.nGroups)) * 4 * SS);
  T.A.ensure(NT * NT);
  T.Ainv.ensure(NT * NT);
  T.sq.ensure(static_cast<size_t>(F) * S);
  T.tl.ensure(static_cast<size_t>(F) * S);
  T.rT.ensure(NT);
  T.t.ensure(2 * NT);
  T.dotPart.ensure(static_cast<size_t>(S) * tlParts(nn));
  T.rec.ensure(2 * NT);
  T.fail.ensure(1);
  T.valid.ensure(1);
  if (!T.counter.p) {
    T.counter.ensure(4);
    HIP_CHECK(hipMemsetAsync(T.counter.p, 0, 4 * sizeof(unsigned int), s));
  }
  HIP_CHECK(hipMemsetAsync(T.valid.p, 0, sizeof(int), s));  // (an inverse of another problem is never kept)
  T.built = false;
  // pair-sharded run with the fused exchange: the ranks' restricted products travel behind [q | Z^T q | p.q] in the same all-reduce
  T.sqPtr = (h->dist() && fusedExchange(h, h->coarseOn)) ? h->dQ.p + exchangeOffsetPq(c, h->coarseOn && h->coarse.denseMode) + 1 : T.sq.p;
  return true;
}

static void temporalInverse(Ctx& c);
static TlTables temporalTables(cvd_handle* h) {
  auto& T = h->temporal;
  return TlTables{T.hx.p, T.bx.p, T.hy.p, T.by.p, T.vW.p, T.vIdx.p, T.elW.p, T.elV.p, T.Sx, T.Sy, T.S, T.width};
}

TlStep temporalStep(cvd_handle* h) {
  if (h == nullptr || !h->temporal.on || !h->temporal.built)
    return TlStep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 1, 0, 0, 0, 1, kTlSpan, nullptr, 1.0};
  auto& T = h->temporal;
  return TlStep{T.Ainv.p, T.sqPtr, T.rT.p, T.t.p, T.tl.p, T.dotPart.p, T.fail.p, T.elW.p, T.elV.p, T.S, T.nn, T.step, T.NT, T.NT, T.width,
                tlParts(T.nn), kTlSpan, (!h->dist() && T.step <= 32) ? T.rec.p : nullptr,  // (one wave's lanes cover a node's 2 step - 1 frames)
                h->opt.temporal_weight};
}

const TlStep* temporalStepDev(cvd_handle* h) {
  return (h != nullptr && h->temporal.on && h->temporal.built) ? h->temporal.stepDev.p : nullptr;
}

// Two halves.  The first -- everything up to the assembled matrix -- reads only what a build of the pose-graph level reads
// (H, lam, mask, x, the table) and writes the level's own buffers: on one GPU it runs on the side stream BESIDE that build (the
// dense inverse of the pose-graph level is one latency-bound persistent kernel that leaves the CUs mostly idle; ~0.25 ms of
// small kernels disappear behind it).  The second half -- the level's own inverse, a gated persistent kernel as well -- follows
// on the solver's stream.  A pair-sharded run keeps everything on the solver's stream: its collectives must stay in one order.
void launchTemporalSetup(Ctx& c, const double* x, int half) {
  cvd_handle* h = c.h;
  auto& T = h->temporal;
  const Layout& L = c.L;
  const bool side = !h->dist() && h->stream2 != nullptr;
  hipStream_t s = (half == 0 && side) ? h->stream2 : h->stream;
  if (half == 1) {
    if (side && T.sidePending) {
      HIP_CHECK(hipStreamWaitEvent(h->stream, T.evDone, 0));
      T.sidePending = false;
    }
    temporalInverse(c);
    return;
  }
  const TlTables tb = temporalTables(h);
  const size_t SS = static_cast<size_t>(T.S) * T.S;
  FrameConst* fcBuf = h->dFc.p;
  if (side) {  // (its own frame constants: the solver's stream rewrites dFc; its own events: the sparse pose-graph level's
               // asynchronous rebuild uses the handle's)
    if (!T.evIn) {
      HIP_CHECK(hipEventCreateWithFlags(&T.evIn, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&T.evDone, hipEventDisableTiming));
    }
    h->dFc2.ensure(L.F);
    fcBuf = h->dFc2.p;
    HIP_CHECK(hipEventRecord(T.evIn, h->stream));
    HIP_CHECK(hipStreamWaitEvent(s, T.evIn, 0));
  }
  hipLaunchKernelGGL(k_frame_consts, dim3((L.F + 63) / 64), dim3(64), 0, s, L, x, fcBuf);
  const size_t ldsD = static_cast<size_t>(L.nD) * T.S * 8 + static_cast<size_t>(T.width) * T.S * 5 + 16;
  allowLds(k_tl_diag, ldsD);
  hipLaunchKernelGGL(k_tl_diag, dim3(L.F), dim3(256), ldsD, s, L, h->dH.p, h->dLam.p, h->dMask.p, tb, T.Cf.p,
                     h->dist() ? h->ownFirst() : 0, h->dist() ? h->ownCount() : L.F);
  HIP_CHECK(hipGetLastError());
  if (c.cross) {
    if (!h->xFa.empty()) {
      allowLds(k_tl_edges_cross, ldsD);
      hipLaunchKernelGGL(k_tl_edges_cross, dim3(static_cast<unsigned>(h->xFa.size())), dim3(256), ldsD, s, L, h->dXFa.p, h->dXFb.p,
                         h->dXBlocks.p, h->dMask.p, tb, T.E.p);
      hipLaunchKernelGGL(k_tl_reduce, dim3(T.nGroups), dim3(256), 0, s, T.S, T.step, T.gOff.p, T.gItems.p, h->dXFa.p, h->dXFb.p, T.E.p,
                         T.part.p);
      HIP_CHECK(hipGetLastError());
    }
  } else if (c.nItems > 0) {
    const size_t ldsE = 2 * static_cast<size_t>(L.B) * 8 + 2 * sizeof(FrameConst) + SS * 8 + 3 * static_cast<size_t>(L.gx + L.gy) * 4 + 16;
    allowLds(k_tl_edges<false>, ldsE);
    hipLaunchKernelGGL(k_tl_edges<false>, dim3(c.nItems), dim3(256), ldsE, s, L, c.T, c.it, x, fcBuf, h->dMask.p, tb, T.E.p);
    HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_tl_reduce, dim3(T.nGroups), dim3(256), 0, s, T.S, T.step, T.gOff.p, T.gItems.p, h->dItemFa.p, h->dItemFb.p,
                       T.E.p, T.part.p);
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipMemsetAsync(T.A.p, 0, static_cast<size_t>(T.NT) * T.NT * sizeof(double), s));
  hipLaunchKernelGGL(k_tl_assemble, dim3(T.nBlocks, static_cast<unsigned>((SS + 255) / 256)), dim3(256), 0, s, T.S, T.nn, T.step, L.F, T.NT, T.blkA.p, T.blkB.p, T.gPtr.p,
                     T.gather.p, T.Cf.p, T.part.p, T.A.p);
  HIP_CHECK(hipGetLastError());
  if (h->dist()) {  // every rank holds the blocks of ITS frames and pairs: the matrices add up
    const int ct = h->tBegin(KC_COMM_COARSE);
    commAllReduce(h, T.A.p, static_cast<size_t>(T.NT) * T.NT, CT_F64, s);
    h->tEnd(ct);
  }
  hipLaunchKernelGGL(k_tl_shift_diag, dim3((T.NT + 255) / 256), dim3(256), 0, s, T.NT, T.NT, T.A.p, h->opt.coarse_dense_shift);
  HIP_CHECK(hipGetLastError());
  if (side) {
    HIP_CHECK(hipEventRecord(T.evDone, s));
    T.sidePending = true;
  }
}

static void temporalInverse(Ctx& c) {
  cvd_handle* h = c.h;
  auto& T = h->temporal;
  hipStream_t s = h->stream;
  HIP_CHECK(hipMemsetAsync(T.fail.p, 0, sizeof(int), s));
  if (h->coarse.ptInvPending) {
    auto& C = h->coarse;
    C.ptInvPending = false;
    DinvRequest a, b;
    a.n = C.ptN; a.A = C.ptMat.p; a.out = C.ptInv.p; a.fail = C.ptInvFail; a.outValid = C.denseValid.p;
    b.n = T.NT; b.A = T.A.p; b.out = T.Ainv.p; b.fail = T.fail.p; b.outValid = T.valid.p; b.panelBuf = &T.invPanel; b.barrierBuf = &T.invBarrier;
    launchDenseSpdInversePair(h, a, b, s);
  } else {
    launchDenseSpdInverse(h, T.NT, T.A.p, T.Ainv.p, T.fail.p, s, T.valid.p, &T.invPanel, &T.invBarrier);
  }
  if (h->dist()) {  // (the ranks must agree on "level on / off": see the dense pose-graph level, launchCoarseSetup)
    hipLaunchKernelGGL(k_flag_to_bool, dim3(1), dim3(1), 0, s, T.fail.p);
    const int ct = h->tBegin(KC_COMM_COARSE);
    commAllReduce(h, T.fail.p, 1, CT_I32, s);
    h->tEnd(ct);
  }
  if (!T.built) {
    T.built = true;
    const TlStep ts = temporalStep(h);
    T.stepDev.ensure(1);
    // (a solve of the same problem finds its record on the device: no upload, and no host wait behind the inverse just enqueued)
    if (T.stepHost.size() != sizeof(TlStep) || std::memcmp(T.stepHost.data(), &ts, sizeof(TlStep)) != 0) {
      HIP_CHECK(hipMemcpyAsync(T.stepDev.p, &ts, sizeof(TlStep), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));  // (`ts` is a local)
      T.stepHost.assign(reinterpret_cast<const unsigned char*>(&ts), reinterpret_cast<const unsigned char*>(&ts) + sizeof(TlStep));
    }
  }
}

// closeScalars: no pose-graph level follows whose kernel would finish the PCG scalars of the first residual
void launchTemporalInit(Ctx& c, bool closeScalars, double tol2) {
  cvd_handle* h = c.h;
  auto& T = h->temporal;
  hipStream_t s = h->stream;
  const TlStep ts = temporalStep(h);
  if (ts.Ainv == nullptr) return;
  const size_t ldsR = (static_cast<size_t>(c.L.B) + static_cast<size_t>(T.S) * T.width) * 8;
  hipLaunchKernelGGL(k_tl_restrict, dim3(c.L.F), dim3(256), ldsR, s, c.L, h->dR.p, temporalStepDev(h));
  const size_t ldsI = static_cast<size_t>(tlRowsLds(T.NT)) * 8;
  allowLds(k_tl_rows_init, ldsI);
  hipLaunchKernelGGL(k_tl_rows_init, dim3(T.S * tlParts(T.nn)), dim3(768), ldsI, s, temporalStepDev(h), c.L.F, h->dScal.p, T.counter.p,
                     closeScalars ? 1 : 0, tol2, h->hPcg);
  HIP_CHECK(hipGetLastError());
}

// ---- temporal pose level (coarse_level 3; k_pt_assemble in cvd_temporal.h) ------------------------------------------------------
static int poseTemporalStep(const cvd_handle* h) { return std::max(2, h->opt.coarse_temporal_step); }

// Node-pair lists for the edge graph of the compiled table (compileTable, after buildCoarsePlan): a frame lies under the hats of
// node f / step and -- unless it sits on a node -- of the next one.
void poseTemporalPlan(cvd_handle* h) {
  auto& C = h->coarse;
  hipStream_t s = h->stream;
  const int step = poseTemporalStep(h), F = h->F;
  const int nn = (F - 1 + step - 1) / step + 1;
  std::map<std::pair<int, int>, std::vector<int>> blocks;  // (a, b), a <= b -> entries 2 * edge + transposed
  for (int a = 0; a < nn; ++a) {
    blocks[{a, a}];
    if (a + 1 < nn) blocks[{a, a + 1}];
  }
  auto nodes = [&](int f, int out[2]) {
    out[0] = f / step;
    int n = 1;
    if (f % step != 0 && out[0] + 1 < nn) out[n++] = out[0] + 1;
    return n;
  };
  for (size_t e = 0; e < C.edgeFaHost.size(); ++e) {
    int na[2], nb[2];
    const int ca = nodes(C.edgeFaHost[e], na), cb = nodes(C.edgeFbHost[e], nb);
    for (int x = 0; x < ca; ++x)
      for (int y = 0; y < cb; ++y) {
        const int en = 2 * static_cast<int>(e);
        if (na[x] < nb[y]) blocks[{na[x], nb[y]}].push_back(en);
        else if (na[x] > nb[y]) blocks[{nb[y], na[x]}].push_back(en + 1);
        else { blocks[{na[x], na[x]}].push_back(en); blocks[{na[x], na[x]}].push_back(en + 1); }
      }
  }
  std::vector<int> blkA, blkB, ptr{0}, list;
  for (auto& b : blocks) {
    blkA.push_back(b.first.first);
    blkB.push_back(b.first.second);
    list.insert(list.end(), b.second.begin(), b.second.end());
    ptr.push_back(static_cast<int>(list.size()));
  }
  if (list.empty()) list.push_back(0);
  C.ptA.upload(blkA.data(), blkA.size(), s);
  C.ptB.upload(blkB.data(), blkB.size(), s);
  C.ptPtr.upload(ptr.data(), ptr.size(), s);
  C.ptList.upload(list.data(), list.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));
  C.ptNn = nn;
  C.ptStepFrames = step;
  C.ptN = nn * kCB;
  C.ptBlocks = static_cast<int>(blkA.size());
}

void poseTemporalPrepare(Ctx& c) {
  cvd_handle* h = c.h;
  auto& C = h->coarse;
  hipStream_t s = h->stream;
  const size_t n = static_cast<size_t>(C.ptN);
  C.ptMat.ensure(n * n);
  C.ptInv.ensure(n * n);
  C.ptR.ensure(n);
  C.ptT.ensure(2 * n);
  C.ptDot.ensure(static_cast<size_t>(kCB) * tlParts(C.ptNn));
  C.ptRec.ensure(2 * n);
  if (!C.ptCounter.p) {
    C.ptCounter.ensure(4);
    HIP_CHECK(hipMemsetAsync(C.ptCounter.p, 0, 4 * sizeof(unsigned int), s));
  }
  // restricted products: Z^T q where the finish half of the product leaves it (behind q in a pair-sharded run: the fused
  // exchange), Z^T r of the first residual in rc; the frames' corrections go where the consumers of the exact level read them (c)
  double* qc = (h->dist() && fusedExchange(h, true)) ? h->dQ.p + exchangeOffsetQc(c) : C.qc.p;
  TlStep st[2];
  st[0] = TlStep{C.ptInv.p, qc, C.ptR.p, C.ptT.p, C.c.p, C.ptDot.p, C.fail.p, nullptr, nullptr, kCB, C.ptNn, C.ptStepFrames, C.ptN, C.ptN, 0,
                 tlParts(C.ptNn), kTlSpan, nullptr,  // (few frames per node: every workgroup sums them itself)
                 h->opt.temporal_weight};
  st[1] = st[0];
  st[1].sq = C.rc.p;
  C.ptStepDev.ensure(2);
  // (padding bytes of the records are not compared: built from zeroed storage)
  unsigned char img[sizeof(st)];
  std::memset(img, 0, sizeof(img));
  TlStep* rec = reinterpret_cast<TlStep*>(img);
  rec[0] = st[0];
  rec[1] = st[1];
  if (C.ptStepHost.size() == sizeof(img) && std::memcmp(C.ptStepHost.data(), img, sizeof(img)) == 0) return;
  HIP_CHECK(hipMemcpyAsync(C.ptStepDev.p, img, sizeof(img), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipStreamSynchronize(s));
  C.ptStepHost.assign(img, img + sizeof(img));
}

void launchPoseTemporalBuild(Ctx& c, hipStream_t s, int* failOut, bool deferInverse) {
  cvd_handle* h = c.h;
  auto& C = h->coarse;
  HIP_CHECK(hipMemsetAsync(C.ptMat.p, 0, static_cast<size_t>(C.ptN) * C.ptN * sizeof(double), s));
  hipLaunchKernelGGL(k_pt_assemble, dim3(C.ptBlocks), dim3(256), 0, s, C.ptNn, C.ptStepFrames, c.L.F, C.ptN, C.ptA.p, C.ptB.p, C.ptPtr.p,
                     C.ptList.p, C.diag.p, C.edges.p, C.edgeFa.p, C.edgeFb.p, C.modeActive.p, C.ptMat.p);
  hipLaunchKernelGGL(k_tl_shift_diag, dim3((C.ptN + 255) / 256), dim3(256), 0, s, C.ptN, C.ptN, C.ptMat.p, h->opt.coarse_dense_shift);
  HIP_CHECK(hipGetLastError());
  if (deferInverse && s == h->stream) {   // (the depth-grid level's inverse follows on this stream: temporalInverse launches both)
    C.ptInvPending = true;
    C.ptInvFail = failOut;
    return;
  }
  launchDenseSpdInverse(h, C.ptN, C.ptMat.p, C.ptInv.p, failOut, s, C.denseValid.p);
}

void launchPoseTemporalInit(Ctx& c, double tol2) {
  cvd_handle* h = c.h;
  auto& C = h->coarse;
  const size_t lds = static_cast<size_t>(tlRowsLds(C.ptN)) * 8;
  allowLds(k_tl_rows_init, lds);
  hipLaunchKernelGGL(k_tl_rows_init, dim3(kCB * tlParts(C.ptNn)), dim3(768), lds, h->stream, C.ptStepDev.p + 1, c.L.F, h->dScal.p, C.ptCounter.p, 1, tol2,
                     h->hPcg);
  HIP_CHECK(hipGetLastError());
}

const TlStep* poseTemporalStepDev(cvd_handle* h) {
  return (h != nullptr && h->coarseOn && h->coarse.temporalPose && h->coarse.denseReady) ? h->coarse.ptStepDev.p : nullptr;
}

void touchModule_temporal() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_tl_reduce));
}

}  // namespace cvd
