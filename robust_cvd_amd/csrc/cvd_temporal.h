// cvd_temporal.h -- third level of the PCG preconditioner: a TEMPORALLY COARSE DEPTH-GRID level.
//
// What the block-Jacobi level + the pose-graph level (cvd_coarse.h: 7 pose unknowns + ONE uniform depth-scale mode per frame)
// leave slow are depth-grid patterns that are spatially structured inside a frame (orthogonal to the uniform mode) and
// temporally SMOOTH across many frames (profiles/r04_pcg_lab_recycling_and_temporal_level.log: 99 % of the energy of the
// smallest Ritz vectors sits in the grid unknowns, spread over all frames).  The level below spans exactly those:
//     P_T = (temporal hat functions, one node every `step` frames) x (bilinear hats of a coarse Sx x Sy grid on the depth grid),
// NT = nn * S unknowns (nn nodes, S = Sx Sy hats; 11 x 45 = 495 at 300 frames, 17 x 10 grid, step 32), additive:
//     M^-1 = blockdiag(M_f^-1) + Z A_c^-1 Z^T + P_T A_T^-1 P_T^T,     A_T = P_T^T (J^T J + diag(lam)) P_T   (Galerkin).
// On the CPU model of the benchmark problem it takes the PCG iterations of the two LM iterations that were examined from 36 / 64
// to 26 / 39.
//
// Galerkin matrix from the pieces the solver already has, no products with the operator:
//   * frame-diagonal part  C_f = Hs^T (H_ff|grid + diag(lam_f)) Hs  from the assembled frame blocks      k_tl_diag
//   * pair part            E_item[s][s'] = sum over the item's constraints  rho' (dr2/dtheta_fa . dr2/dtheta_fb) u_s v_s'
//     (only the disparity row of a flow constraint depends on BOTH frames' depth grids; u, v = the constraint's bilinear taps
//     composed with the hats: separable, <= 3 x 3 hats per side)                                          k_tl_edges
//     (dense mode with explicit cross blocks: a projection of the pair's block instead                     k_tl_edges_cross)
//   * temporal reduction   groups of items whose frames lie in the same pair of node intervals are summed with the four
//     products of their temporal weights (k_tl_reduce), the (node, node) blocks gather the groups' sums and the frames'
//     C_f (k_tl_assemble); A_T is block-banded in the node index (|a - b| <= 2), stored dense, unknown e = s * nn + a.
//   * inverse              k_dense_spd_inverse (cvd_dense_inverse.h), f64.
// Per PCG iteration the level lives inside the existing launches (TlStep / CoarseView::tl in cvd_device.h, tlLevelRows in
// cvd_kernels.h): nothing here.
#pragma once

#include "cvd_kernels.h"

namespace cvd {

// 1-D tables of the separable hats: fine vertex i has the (<= 2) coarse hats b[i], b[i] + 1 with weights h[2 i], h[2 i + 1]
struct TlTables {
  const float* hx;
  const int* bx;
  const float* hy;
  const int* by;
  const float4* vW;          // vertex table (CoarseView::tlW / tlIdx)
  const unsigned int* vIdx;
  const float* elW;          // transposed (TlStep::elW / elV)
  const unsigned char* elV;
  int Sx, Sy, S, width;
};

// <= 3 consecutive coarse hats of one axis at a constraint's position (fine cell i, fraction r): w[k] belongs to hat j0 + k
__device__ __forceinline__ void tlAxis(const float* __restrict__ h, const int* __restrict__ b, int i, double r, int& j0, double w[3]) {
  j0 = b[i];
  const int o = b[i + 1] - j0;  // 0 or 1 (the coarse grid is not finer than the fine one)
  const double lo = 1.0 - r;
  w[0] = lo * static_cast<double>(h[2 * i]);
  w[1] = lo * static_cast<double>(h[2 * i + 1]);
  w[2] = 0.0;
  const double h0 = r * static_cast<double>(h[2 * i + 2]), h1 = r * static_cast<double>(h[2 * i + 3]);
  if (o == 0) {
    w[0] += h0;
    w[1] += h1;
  } else {
    w[1] += h0;
    w[2] += h1;
  }
}

// ---- frame-diagonal part: C_f = Hs^T (H_ff restricted to the depth grid + diag(lam)) Hs, S x S per frame ---------------------
// One workgroup per frame.  T1 = (H + lam) Hs in LDS ([G][S]; thread = row v, the <= 4 hats of every column vertex v' come from
// the vertex table), then C = Hs^T T1 through the transposed table.  Frames whose depth unknowns are masked give zero.
// (pair-sharded run: the reduced H_ff lives on the frame's owner rank -- frames outside [ownFirst, ownFirst + ownCount) give zero
// here and the ranks' matrices are summed)
inline __global__ __launch_bounds__(256) void k_tl_diag(Layout L, const double* __restrict__ hBlocks, const double* __restrict__ lam,
                                                 const double* __restrict__ mask, TlTables T, double* __restrict__ Cf, int ownFirst,
                                                 int ownCount) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B, G = L.nD, S = T.S, f = blockIdx.x, tid = threadIdx.x;
  double* T1 = sm;                                              // [G][S]
  float* ew = reinterpret_cast<float*>(T1 + static_cast<size_t>(G) * S);   // [width][S]
  unsigned char* ev = reinterpret_cast<unsigned char*>(ew + T.width * S);
  double* out = Cf + static_cast<size_t>(f) * S * S;
  if (mask[static_cast<size_t>(f) * B + 7] == 0.0 || f < ownFirst || f >= ownFirst + ownCount) {
    for (int e = tid; e < S * S; e += 256) out[e] = 0.0;
    return;
  }
  for (int e = tid; e < T.width * S; e += 256) {
    ew[e] = T.elW[e];
    ev[e] = T.elV[e];
  }
  const double* hf = hBlocks + static_cast<size_t>(f) * B * B;
  for (int v = tid; v < G; v += 256) {
    double* row = T1 + static_cast<size_t>(v) * S;
    for (int s = 0; s < S; ++s) row[s] = 0.0;
    const double lv = lam[static_cast<size_t>(f) * B + 7 + v];
    for (int vp = 0; vp < G; ++vp) {
      // (symmetric block: entry (v, v') read as (v', v), coalesced over v)
      const double hv = hf[static_cast<size_t>(7 + vp) * B + 7 + v] + (vp == v ? lv : 0.0);
      const float4 w = T.vW[vp];
      const unsigned int idx = T.vIdx[vp];
      row[idx & 255u] += static_cast<double>(w.x) * hv;
      row[(idx >> 8) & 255u] += static_cast<double>(w.y) * hv;
      row[(idx >> 16) & 255u] += static_cast<double>(w.z) * hv;
      row[idx >> 24] += static_cast<double>(w.w) * hv;
    }
  }
  __syncthreads();
  for (int e = tid; e < S * S; e += 256) {
    const int s = e / S, sp = e - s * S;
    double a = 0.0;
    for (int k = 0; k < T.width; ++k) a += static_cast<double>(ew[k * S + s]) * T1[static_cast<size_t>(ev[k * S + s]) * S + sp];
    out[e] = a;
  }
}

// ---- pair part: E_item = sum_c rho'_c (dr2/dtheta_fa)(dr2/dtheta_fb)^T projected on the hats, rows = hats of fa --------------
// The residual chain of k_coarse_edges_fast (cvd_coarse.h) cut down to what the depth-depth coupling needs: the disparity row.
// Scope of the fast kernels with a bilinear one-parameter depth grid (KD = 4, N = 1).
template <bool DENSE>
inline __global__ __launch_bounds__(256) void k_tl_edges(Layout L, Table T, Items it, const double* __restrict__ x,
                                                  const FrameConst* __restrict__ fc, const double* __restrict__ mask,
                                                  TlTables Tb, double* __restrict__ Eout) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr double eps = 1e-6;
  const int B = L.B, S = Tb.S;
  double* xa = sm;
  double* xb = sm + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(sm + 2 * B);
  double* E = reinterpret_cast<double*>(fcs + 2);                // [S][S]
  float* hx = reinterpret_cast<float*>(E + S * S);               // 2 gx
  float* hy = hx + 2 * L.gx;                                     // 2 gy
  int* bx = reinterpret_cast<int*>(hy + 2 * L.gy);
  int* by = bx + L.gx;
  const int item = blockIdx.x, tid = threadIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  double* out = Eout + static_cast<size_t>(item) * S * S;
  if (mask[static_cast<size_t>(fa) * B + 7] == 0.0 || mask[static_cast<size_t>(fb) * B + 7] == 0.0) {
    for (int e = tid; e < S * S; e += 256) out[e] = 0.0;
    return;
  }
  for (int i = tid; i < B; i += 256) {
    xa[i] = x[static_cast<size_t>(fa) * B + i];
    xb[i] = x[static_cast<size_t>(fb) * B + i];
  }
  constexpr int FCW = sizeof(FrameConst) / 8;
  if (tid < 2 * FCW) {
    const int which = tid / FCW, k = tid % FCW;
    reinterpret_cast<double*>(fcs + which)[k] = reinterpret_cast<const double*>(fc + (which ? fb : fa))[k];
  }
  for (int e = tid; e < S * S; e += 256) E[e] = 0.0;
  for (int i = tid; i < 2 * L.gx; i += 256) hx[i] = Tb.hx[i];
  for (int i = tid; i < 2 * L.gy; i += 256) hy[i] = Tb.hy[i];
  for (int i = tid; i < L.gx; i += 256) bx[i] = Tb.bx[i];
  for (int i = tid; i < L.gy; i += 256) by[i] = Tb.by[i];
  __syncthreads();
  const double A = L.aspect;
  for (int dir = 0; dir < 2; ++dir) {
    const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
    const FrameConst& Fa = fcs[dir];      // source frame of this direction
    const FrameConst& Fb = fcs[dir ^ 1];  // target frame
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    const double fya = Fa.fy, fxa = Fa.fy * A;
    const double fyb = Fb.fy;
    const double ifyb = 1.0 / fyb, ifxb = 1.0 / (fyb * A);
    const int fsrc = dir ? fb : fa, ftgt = dir ? fa : fb;
    const long long pixBase = DENSE ? (cb / (static_cast<long long>(T.W) * T.H)) * (static_cast<long long>(T.W) * T.H) : 0;
    // (deterministic build: the first wave alone walks the constraints -- its LDS atomics land in program order)
    for (long long c = cb + tid; c < ce && tid < kAtomicWalkers256; c += kAtomicWalkers256) {
      float4 nd;
      float2 d;
      if (!loadConstraint<DENSE>(T, c, pixBase, fsrc, ftgt, nd, d)) continue;
      const double da = static_cast<double>(d.x), db = static_cast<double>(d.y);
      // fine cells and fractions of both sides (gridCell: the taps of bilinearTaps)
      int ixa, iya, ixb, iyb;
      double rxa, rya, rxb, ryb;
      gridCell(nd.x, L.gx, L.maxcx, ixa, rxa);
      gridCell(nd.y, L.gy, L.maxcy, iya, rya);
      gridCell(nd.z, L.gx, L.maxcx, ixb, rxb);
      gridCell(nd.w, L.gy, L.maxcy, iyb, ryb);
      const int ia = ixa + iya * L.gx, ib = ixb + iyb * L.gx;
      const double sa = (1.0 - rxa) * (1.0 - rya) * xs[7 + ia] + rxa * (1.0 - rya) * xs[7 + ia + 1] +
                        (1.0 - rxa) * rya * xs[7 + ia + L.gx] + rxa * rya * xs[7 + ia + L.gx + 1];
      const double sb = (1.0 - rxb) * (1.0 - ryb) * xt[7 + ib] + rxb * (1.0 - ryb) * xt[7 + ib + 1] +
                        (1.0 - rxb) * ryb * xt[7 + ib + L.gx] + rxb * ryb * xt[7 + ib + L.gx + 1];
      const double Da = da * sa, Db = db * sb;
      const double pax = static_cast<double>(nd.x), pay = static_cast<double>(nd.y);
      const double pbx = static_cast<double>(nd.z), pby = static_cast<double>(nd.w);
      const double ca[3] = {pax * fxa, pay * fya, -1.0};
      const double Rca[3] = {dot3(Fa.R, ca), dot3(Fa.R + 3, ca), dot3(Fa.R + 6, ca)};
      const double v[3] = {Fa.t[0] + Rca[0] * Da - Fb.t[0], Fa.t[1] + Rca[1] * Da - Fb.t[1], Fa.t[2] + Rca[2] * Da - Fb.t[2]};
      const double q0 = Fb.R[0] * v[0] + Fb.R[3] * v[1] + Fb.R[6] * v[2];
      const double q1 = Fb.R[1] * v[0] + Fb.R[4] * v[1] + Fb.R[7] * v[2];
      const double q2 = Fb.R[2] * v[0] + Fb.R[5] * v[1] + Fb.R[8] * v[2];
      const double zz = -q2;
      const double iz = 1.0 / zz;
      const double u = q0 * iz * ifxb;
      const double vv = q1 * iz * ifyb;
      const double r0 = (u - pbx) * L.ws;
      const double r1 = (vv - pby) * L.ws;
      double r2, dr2dA, dr2dDb;
      if (L.lossType == kLossDisparity) {
        const bool zo = !(zz < eps), bo = !(Db < eps);
        const double izc = zo ? iz : 1.0 / eps, ibc = 1.0 / (bo ? Db : eps);
        r2 = (izc - ibc) * L.wd;
        dr2dA = zo ? (-L.wd * izc * izc) : 0.0;
        dr2dDb = bo ? (L.wd * ibc * ibc) : 0.0;
      } else {
        const bool zIsMax = !(zz < Db), zIsMin = !(Db < zz);
        const double mx = zIsMax ? zz : Db, mn = zIsMin ? zz : Db;
        if (L.lossType == kLossRatio) {
          r2 = (mx / mn - 1.0) * L.wd;
          const double dmx = 1.0 / mn, dmn = -mx / (mn * mn);
          dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
          dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
        } else {
          r2 = log(mn / mx) * L.wd;
          const double dmn = 1.0 / mn, dmx = -1.0 / mx;
          dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
          dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
        }
      }
      const double w = robustRho1(L, r0 * r0 + r1 * r1 + r2 * r2);
      // d r2 / d (source depth) = m22 (R_b^T R_a c_a)_z with m22 = -dr2dA;  per vertex: x d_src x tap weight
      const double g2 = -dr2dA * (Fb.R[2] * Rca[0] + Fb.R[5] * Rca[1] + Fb.R[8] * Rca[2]) * da;
      const double coef = w * g2 * (dr2dDb * db);
      if (coef == 0.0) continue;
      int jxs, jys, jxt, jyt;
      double wxs[3], wys[3], wxt[3], wyt[3];
      tlAxis(hx, bx, ixa, rxa, jxs, wxs);
      tlAxis(hy, by, iya, rya, jys, wys);
      tlAxis(hx, bx, ixb, rxb, jxt, wxt);
      tlAxis(hy, by, iyb, ryb, jyt, wyt);
      // rows = hats of fa: the source side in direction 0, the target side in direction 1
#pragma unroll
      for (int ys = 0; ys < 3; ++ys) {
#pragma unroll
        for (int xs3 = 0; xs3 < 3; ++xs3) {
          const double ws = wxs[xs3] * wys[ys];
          if (ws == 0.0) continue;
          const int hs = (jxs + xs3) + (jys + ys) * Tb.Sx;
#pragma unroll
          for (int yt = 0; yt < 3; ++yt) {
#pragma unroll
            for (int xt3 = 0; xt3 < 3; ++xt3) {
              const double wt = wxt[xt3] * wyt[yt];
              if (wt == 0.0) continue;
              const int ht = (jxt + xt3) + (jyt + yt) * Tb.Sx;
              atomicAdd(&E[dir == 0 ? hs * S + ht : ht * S + hs], coef * ws * wt);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < S * S; e += 256) out[e] = E[e];
}

// Dense mode with explicit cross blocks (cvd_cross.h): the pair part is a projection of the pair's B x B block X_ab (rows = fa's
// unknowns) -- E = Hs^T X|grid Hs in two gather stages through the transposed vertex table, no walk over the pixels.
inline __global__ __launch_bounds__(256) void k_tl_edges_cross(Layout L, const int* __restrict__ pairFa, const int* __restrict__ pairFb,
                                                        const double* __restrict__ X, const double* __restrict__ mask, TlTables T,
                                                        double* __restrict__ Eout) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B, G = L.nD, S = T.S, pair = blockIdx.x, tid = threadIdx.x;
  double* T1 = sm;                                                        // [G][S]
  float* ew = reinterpret_cast<float*>(T1 + static_cast<size_t>(G) * S);  // [width][S]
  unsigned char* ev = reinterpret_cast<unsigned char*>(ew + T.width * S);
  double* out = Eout + static_cast<size_t>(pair) * S * S;
  const int fa = pairFa[pair], fb = pairFb[pair];
  if (mask[static_cast<size_t>(fa) * B + 7] == 0.0 || mask[static_cast<size_t>(fb) * B + 7] == 0.0) {
    for (int e = tid; e < S * S; e += 256) out[e] = 0.0;
    return;
  }
  for (int e = tid; e < T.width * S; e += 256) {
    ew[e] = T.elW[e];
    ev[e] = T.elV[e];
  }
  __syncthreads();
  const double* Xp = X + static_cast<size_t>(pair) * B * B;
  for (int e = tid; e < G * S; e += 256) {  // T1[v][s'] = sum_k w_k X[7 + v][7 + vertex k of hat s']
    const int v = e / S, sp = e - v * S;
    const double* row = Xp + static_cast<size_t>(7 + v) * B + 7;
    double a = 0.0;
    for (int k = 0; k < T.width; ++k) a += static_cast<double>(ew[k * S + sp]) * row[ev[k * S + sp]];
    T1[e] = a;
  }
  __syncthreads();
  for (int e = tid; e < S * S; e += 256) {
    const int s = e / S, sp = e - s * S;
    double a = 0.0;
    for (int k = 0; k < T.width; ++k) a += static_cast<double>(ew[k * S + s]) * T1[static_cast<size_t>(ev[k * S + s]) * S + sp];
    out[e] = a;
  }
}

// ---- temporal reduction of the pair parts ---------------------------------------------------------------------------------
// Group g = items [gOff[g], gOff[g + 1]) of gItems whose frames lie in the same pair of node intervals (i, j) = (fa / step,
// fb / step); part[g][k], k = 2 ka + kb: sum of E_item weighted with w_{i + ka}(fa) w_{j + kb}(fb).
constexpr int kTlEntriesPerThread = 16;  // S^2 <= 256 * 16 = 4096: S <= 64
inline __global__ __launch_bounds__(256) void k_tl_reduce(int S, int step, const int* __restrict__ gOff, const int* __restrict__ gItems,
                                                   const int* __restrict__ itemFa, const int* __restrict__ itemFb,
                                                   const double* __restrict__ E, double* __restrict__ part) {
  const int g = blockIdx.x, tid = threadIdx.x, SS = S * S;
  double acc[kTlEntriesPerThread][4];
#pragma unroll
  for (int u = 0; u < kTlEntriesPerThread; ++u) acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.0;
  const double inv = 1.0 / static_cast<double>(step);
  for (int k = gOff[g]; k < gOff[g + 1]; ++k) {
    const int item = gItems[k];
    const int fa = itemFa[item], fb = itemFb[item];
    const double ta = static_cast<double>(fa % step) * inv, tb = static_cast<double>(fb % step) * inv;
    const double w00 = (1.0 - ta) * (1.0 - tb), w01 = (1.0 - ta) * tb, w10 = ta * (1.0 - tb), w11 = ta * tb;
    const double* src = E + static_cast<size_t>(item) * SS;
#pragma unroll
    for (int u = 0; u < kTlEntriesPerThread; ++u) {
      const int e = tid + u * 256;
      const double v = e < SS ? src[e] : 0.0;
      acc[u][0] += w00 * v;
      acc[u][1] += w01 * v;
      acc[u][2] += w10 * v;
      acc[u][3] += w11 * v;
    }
  }
#pragma unroll
  for (int u = 0; u < kTlEntriesPerThread; ++u) {
    const int e = tid + u * 256;
    if (e >= SS) continue;
#pragma unroll
    for (int k = 0; k < 4; ++k) part[(static_cast<size_t>(g) * 4 + k) * SS + e] = acc[u][k];
  }
}

// ---- (node, node) blocks of A_T --------------------------------------------------------------------------------------------
// Workgroup = block (a, b), a <= b <= a + 2 (blkA / blkB).  Its entries: the frames' C_f with w_a(f) w_b(f) (b <= a + 1) and the
// group sums listed in gather[gPtr[blk] .. gPtr[blk + 1]): entry = 2 * part index + (1: transposed).  Written to both triangles
// of the dense matrix (unknown e = s * nn + a, leading dimension ld; zeroed by the host).  k_tl_shift_diag follows (after the
// ranks' matrices have been summed in a pair-sharded run): relative shift on the diagonal, empty diagonal entries (no active
// frame under the node) become identity rows.
inline __global__ __launch_bounds__(256) void k_tl_assemble(int S, int nn, int step, int F, int ld, const int* __restrict__ blkA,
                                                     const int* __restrict__ blkB, const int* __restrict__ gPtr,
                                                     const int* __restrict__ gather, const double* __restrict__ Cf,
                                                     const double* __restrict__ part, double* __restrict__ Aout) {
  // grid (blocks, ceil(S^2 / 256)): one entry per thread, four independent sums in flight
  const int blk = blockIdx.x, SS = S * S;
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= SS) return;
  const int a = blkA[blk], b = blkB[blk];
  const double inv = 1.0 / static_cast<double>(step);
  const int s = e / S, sp = e - s * S;
  if (a == b && s > sp) return;  // (a diagonal block is symmetric: its upper triangle is computed and mirrored, bit for bit)
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  if (b <= a + 1) {
    const int fLo = max(0, (b - 1) * step + 1), fHi = min(F - 1, (a + 1) * step - 1);
    auto term = [&](int f) -> double {
      const int fc = min(f, fHi);
      const double wa = 1.0 - fabs(static_cast<double>(fc - a * step)) * inv, wb = 1.0 - fabs(static_cast<double>(fc - b * step)) * inv;
      const double c = Cf[static_cast<size_t>(fc) * SS + e];
      return f <= fHi ? wa * wb * c : 0.0;
    };
    for (int f = fLo; f <= fHi; f += 4) {
      v0 += term(f);
      v1 += term(f + 1);
      v2 += term(f + 2);
      v3 += term(f + 3);
    }
  }
  const int k1 = gPtr[blk + 1];
  auto gterm = [&](int k) -> double {
    const int en = gather[min(k, k1 - 1)];
    const double* src = part + static_cast<size_t>(en >> 1) * SS;
    const double g = (en & 1) ? src[sp * S + s] : src[e];
    return k < k1 ? g : 0.0;
  };
  for (int k = gPtr[blk]; k < k1; k += 4) {
    v0 += gterm(k);
    v1 += gterm(k + 1);
    v2 += gterm(k + 2);
    v3 += gterm(k + 3);
  }
  const double v = (v0 + v1) + (v2 + v3);
  const size_t r = static_cast<size_t>(s) * nn + a, c = static_cast<size_t>(sp) * nn + b;
  Aout[r * ld + c] = v;
  if (r != c) Aout[c * ld + r] = v;
}
inline __global__ void k_tl_shift_diag(int n, int ld, double* __restrict__ A, double shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = A[static_cast<size_t>(i) * ld + i];
  A[static_cast<size_t>(i) * ld + i] = v > 0.0 ? v * (1.0 + shift) : 1.0;
}

// ---- TEMPORAL POSE LEVEL (cvd_solver_options::coarse_level = 3) -----------------------------------------------------------------
// The pose-graph level restricted to temporally smooth corrections: the frame's 8 modes (cvd_coarse.h: 7 pose-like unknowns + the
// uniform depth-scale mode) x temporal hats with a node every `step` frames (default 8: 39 nodes, 312 unknowns at 300 frames,
// against 2400 for the exact level).  Its Galerkin matrix is a weighted sum of the very 8 x 8 blocks the exact level is made of
// (k_coarse_diag, k_coarse_edges*): block (a, b), a <= b, of node pairs = sum over frames under both hats of w_a w_b D_f + sum
// over the listed edges (entry = 2 * edge + (1: the edge's block transposed, i.e. node a is the edge's SECOND frame)) of
// w_a w_b E_e.  One wave per node pair, lane = (i, j); fixed order (the ranks of a sharded run build the same bits).  Unknown
// e = mode * nn + node, as the third level's (tlLevelRows serves both).  Inactive modes contribute nothing; k_tl_shift_diag turns
// their empty diagonal into identity rows.
inline __global__ __launch_bounds__(256) void k_pt_assemble(int nn, int step, int F, int ld, const int* __restrict__ blkA,
                                                     const int* __restrict__ blkB, const int* __restrict__ ptr,
                                                     const int* __restrict__ list, const double* __restrict__ diag,
                                                     const double* __restrict__ edges, const int* __restrict__ edgeFa,
                                                     const int* __restrict__ edgeFb, const unsigned char* __restrict__ modeActive,
                                                     double* __restrict__ A) {
  // four waves per node pair: wave w takes every fourth frame / list entry (lane = (i, j)), the four partial blocks are added in
  // wave order -- the walk is a chain of dependent loads, ~200 list entries for a diagonal node pair
  __shared__ double part[4][kCBB];
  const int blk = blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l >> 3, j = l & 7;
  const int a = blkA[blk], b = blkB[blk];
  const double inv = 1.0 / static_cast<double>(step);
  auto hat = [&](int node, int f) -> double { return fmax(0.0, 1.0 - fabs(static_cast<double>(f - node * step)) * inv); };
  double v = 0.0;
  if (b <= a + 1) {
    const int fLo = max(0, (b - 1) * step + 1), fHi = min(F - 1, (a + 1) * step - 1);
    for (int f = fLo + w; f <= fHi; f += 4)
      if (modeActive[f * kCB + i] && modeActive[f * kCB + j]) v += hat(a, f) * hat(b, f) * diag[static_cast<size_t>(f) * kCBB + l];
  }
  for (int k = ptr[blk] + w; k < ptr[blk + 1]; k += 4) {
    const int en = list[k], e = en >> 1;
    const int fa = edgeFa[e], fb = edgeFb[e];  // block stored rows = fa, columns = fb
    if (en & 1) {
      if (modeActive[fb * kCB + i] && modeActive[fa * kCB + j]) v += hat(a, fb) * hat(b, fa) * edges[static_cast<size_t>(e) * kCBB + j * kCB + i];
    } else {
      if (modeActive[fa * kCB + i] && modeActive[fb * kCB + j]) v += hat(a, fa) * hat(b, fb) * edges[static_cast<size_t>(e) * kCBB + l];
    }
  }
  part[w][l] = v;
  __syncthreads();
  if (w != 0 || (a == b && i > j)) return;  // (a diagonal block is symmetric: its upper triangle is computed and mirrored)
  v = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
  const size_t r = static_cast<size_t>(i) * nn + a, c = static_cast<size_t>(j) * nn + b;
  A[r * ld + c] = v;
  if (r != c) A[c * ld + r] = v;
}

// ---- first residual of a PCG solve -------------------------------------------------------------------------------------------
// sq[f][s] = spatial restriction of the (masked) residual of frame f
inline __global__ __launch_bounds__(256) void k_tl_restrict(Layout L, const double* __restrict__ r, const TlStep* __restrict__ tsp) {
  const TlStep ts = *tsp;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B, f = blockIdx.x, tid = threadIdx.x;
  double* rf = sm;
  double* prod = sm + B;
  for (int i = tid; i < B; i += 256) rf[i] = r[static_cast<size_t>(f) * B + i];
  __syncthreads();
  const int nE = ts.S * ts.width;
  for (int e = tid; e < nE; e += 256) prod[e] = static_cast<double>(ts.elW[e]) * rf[7 + ts.elV[e]];
  __syncthreads();
  if (tid < ts.S) {
    double a = 0.0;
    for (int k = 0; k < ts.width; ++k) a += prod[k * ts.S + tid];
    ts.sq[static_cast<size_t>(f) * ts.S + tid] = a;
  }
}
// t = A_T^-1 P^T r, r_T = P^T r, tl, and the level's part of r^T z added to S_RZPART (k_cg_update(init) left the block-Jacobi part
// there; the pose-graph level's kernel closes the scalars afterwards)
inline __global__ __launch_bounds__(1024) void k_tl_rows_init(const TlStep* __restrict__ tsp, int F, double* __restrict__ scal,
                                                       unsigned int* __restrict__ counter, int closeScalars, double tol2,
                                                       double* __restrict__ hostMirror) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int flag;
  NoMid mid;
  double alpha = 0.0;
  (void)tlLevelRows<false>(tsp, blockIdx.x, F, alpha, 1, 0.0, scal, sm, mid);
  if (!lastBlockArrivesLite(counter, gridDim.x, &flag)) return;
  if (threadIdx.x == 0) {
    double d = 0.0;
    for (int s = 0; s < tsp->S * tsp->parts; ++s) d += readPartial(tsp->dotPart + s);
    if (closeScalars) pcgFinishScalars(scal, 1, scal[S_RZPART] + d, scal[S_RR], tol2, hostMirror);
    else scal[S_RZPART] += d;
  }
}

}  // namespace cvd
