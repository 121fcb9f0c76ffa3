// robust_cvd_amd/csrc/cvd_cross.h
//
// DENSE MODE with explicit cross blocks (reference lib/FlowConstraints.cpp:315-329,381-465 with
// matchSeparation = 0: every masked in-bounds pixel of a frame pair is a constraint).
//
// The matrix-free product re-derives every constraint's Jacobian row in every PCG iteration.  That is the right trade for
// the SAMPLED constraint lists (~600 constraints per frame pair against B^2 = 31 k block entries), and the wrong one in
// dense mode: ~81 k pixel constraints per directed pair, 83 products per LM iteration.  Here the off-diagonal blocks
//     X_ab = sum over the constraints of {a -> b, b -> a} of rho' J_a^T J_b          (B x B doubles, a < b)
// are assembled ONCE per Jacobian evaluation (k_cross_assemble) and a product streams them (k_cross_matvec: 2 B^2 flop
// per 8 B^2 bytes -- HBM-bound); the frame-diagonal part of J^T J p comes from H_ff, which the assembly kernels form anyway
// (k_matvec_finish, Hdiag).  Scope: the fast kernels' (identity spatial transform, reprojection losses, one value
// parameter per vertex, bilinear grids, Fixed / PerFrame intrinsics), single GPU, no triplets.
//
// A constraint's Jacobian row on either side is [ pose part Jp (3 x 7) | JD (3) x tap factors ]: the grid columns are
// rank one in (residual, tap).  Per constraint the block receives
//     pose x pose   7 x 7   w Jp_a^T Jp_b                      -> registers, folded per workgroup
//     pose x grid   7 x 4   (w Jp_a^T JD_b) fac_b[k]           -> LDS f64 atomics
//     grid x pose   4 x 7   fac_a[k] (w JD_a^T Jp_b)           -> LDS f64 atomics
//     grid x grid   4 x 4   (w JD_a . JD_b) fac_a[k] fac_b[l]  -> LDS f64 atomics
// The grid x grid part (G^2 doubles: 231 KB at the 17x10 grid) exceeds the LDS, so it is accumulated in column PANELS by a
// kernel of its own (two panels at G = 170; see k_cross_assemble).
#pragma once

#include "cvd_kernels.h"

namespace cvd {

// Undirected frame pairs of the explicit-block mode: ranges of both directions' pixel slots (either may be empty) and the
// two rows of the partial-product buffer.
struct CrossPairs {
  const int* fa;            // fa < fb
  const int* fb;
  const long long* range;   // 4 per pair: [a -> b begin, end, b -> a begin, end) pixel slots
  const int* slot;          // 2 per pair: rows of the partial buffer (frame-major)
  int count;
};

constexpr int kCrossThreads = 512;
constexpr int kCrossRun = 16;  // consecutive pixels per lane (see kDenseRun: lanes of a wave then touch different cells;
                               // 32 -> 16 measured 12 % off the three assembly kernels, 8 the same, 4 and 64 worse)

// Everything of one constraint that the block needs: both sides' pose rows, depth-row factors, taps.
template <int KD>
struct CrossRows {
  double JpS[3][7];  // d r / d (t, w, fy) of the SOURCE frame
  double JpT[3][7];  // ... of the TARGET frame
  double JDS[3];     // d r / d D_source
  double JDT2;       // d r_2 / d D_target (rows 0, 1 are zero)
  double w;          // rho'
  FastTaps<KD> ts, tt;
  double ds, dt;     // source depths at the two end points
};

// Source frame S -> target frame T.  Follows k_assemble_fast's two branches (side 0 = source, side 1 = target).
template <int KD>
__device__ __forceinline__ void crossRows(const Layout& L, const FrameConst& Fs, const FrameConst& Ft,
                                          const double* __restrict__ xs, const double* __restrict__ xt, const float4& nd,
                                          const float2& d, CrossRows<KD>& o) {
  constexpr double eps = 1e-6;
  const double A = L.aspect;
  const double da = static_cast<double>(d.x), db = static_cast<double>(d.y);
  o.ds = da;
  o.dt = db;
  fastGather<KD>(L, nd.x, nd.y, o.ts);
  fastGather<KD>(L, nd.z, nd.w, o.tt);
  double Da = 0.0, Db = 0.0;
#pragma unroll
  for (int k = 0; k < KD; ++k) {
    Da += da * xs[7 + o.ts.I(k)] * o.ts.Wt(k);
    Db += db * xt[7 + o.tt.I(k)] * o.tt.Wt(k);
  }
  const double fys = Fs.fy, fxs = Fs.fy * A;
  const double fyt = Ft.fy;
  const double ifyt = 1.0 / fyt, ifxt = 1.0 / (fyt * A);
  const double pax = static_cast<double>(nd.x), pay = static_cast<double>(nd.y);
  const double pbx = static_cast<double>(nd.z), pby = static_cast<double>(nd.w);
  const double ca[3] = {pax * fxs, pay * fys, -1.0};
  const double Rca[3] = {dot3(Fs.R, ca), dot3(Fs.R + 3, ca), dot3(Fs.R + 6, ca)};
  const double v[3] = {Fs.t[0] + Rca[0] * Da - Ft.t[0], Fs.t[1] + Rca[1] * Da - Ft.t[1], Fs.t[2] + Rca[2] * Da - Ft.t[2]};
  const double q0 = Ft.R[0] * v[0] + Ft.R[3] * v[1] + Ft.R[6] * v[2];
  const double q1 = Ft.R[1] * v[0] + Ft.R[4] * v[1] + Ft.R[7] * v[2];
  const double q2 = Ft.R[2] * v[0] + Ft.R[5] * v[1] + Ft.R[8] * v[2];
  const double zz = -q2;
  const double iz = 1.0 / zz;
  const double u = q0 * iz * ifxt;
  const double vv = q1 * iz * ifyt;
  double r[3];
  r[0] = (u - pbx) * L.ws;
  r[1] = (vv - pby) * L.ws;
  double dr2dA, dr2dDb;
  if (L.lossType == kLossDisparity) {
    const bool zo = !(zz < eps), bo = !(Db < eps);
    const double izc = zo ? iz : 1.0 / eps, ibc = 1.0 / (bo ? Db : eps);
    r[2] = (izc - ibc) * L.wd;
    dr2dA = zo ? (-L.wd * izc * izc) : 0.0;
    dr2dDb = bo ? (L.wd * ibc * ibc) : 0.0;
  } else {
    const bool zIsMax = !(zz < Db), zIsMin = !(Db < zz);
    const double mx = zIsMax ? zz : Db, mn = zIsMin ? zz : Db;
    if (L.lossType == kLossRatio) {
      r[2] = (mx / mn - 1.0) * L.wd;
      const double dmx = 1.0 / mn, dmn = -mx / (mn * mn);
      dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
      dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
    } else {
      r[2] = log(mn / mx) * L.wd;
      const double dmn = 1.0 / mn, dmx = -1.0 / mx;
      dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
      dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
    }
  }
  double rho0;
  robustRho(L, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho0, o.w);
  // d r / d q (rows): M0 = (m00, 0, m02), M1 = (0, m11, m12), M2 = (0, 0, m22);  G = M R_t^T
  const double wiz = L.ws * iz;
  const double m00 = wiz * ifxt, m11 = wiz * ifyt, m02 = wiz * u, m12 = wiz * vv, m22 = -dr2dA;
  double G[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    G[0][i] = m00 * Ft.R[i * 3 + 0] + m02 * Ft.R[i * 3 + 2];
    G[1][i] = m11 * Ft.R[i * 3 + 1] + m12 * Ft.R[i * 3 + 2];
    G[2][i] = m22 * Ft.R[i * 3 + 2];
  }
  const double cf[3] = {pax * A, pay, 0.0};
  const double dXdf[3] = {Da * (Fs.R[0] * cf[0] + Fs.R[1] * cf[1]), Da * (Fs.R[3] * cf[0] + Fs.R[4] * cf[1]),
                          Da * (Fs.R[6] * cf[0] + Fs.R[7] * cf[1])};
#pragma unroll
  for (int rr = 0; rr < 3; ++rr) {
    o.JpS[rr][0] = G[rr][0];
    o.JpS[rr][1] = G[rr][1];
    o.JpS[rr][2] = G[rr][2];
    o.JpS[rr][6] = dot3(G[rr], dXdf);
    o.JDS[rr] = dot3(G[rr], Rca);
    o.JpT[rr][0] = -G[rr][0];
    o.JpT[rr][1] = -G[rr][1];
    o.JpT[rr][2] = -G[rr][2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double dX[3] = {Da * dot3(Fs.dR[i], ca), Da * dot3(Fs.dR[i] + 3, ca), Da * dot3(Fs.dR[i] + 6, ca)};
    o.JpS[0][3 + i] = dot3(G[0], dX);
    o.JpS[1][3 + i] = dot3(G[1], dX);
    o.JpS[2][3 + i] = dot3(G[2], dX);
    const double* D = Ft.dR[i];  // d q / d w_t,i = dR_t,i^T v
    const double dq0 = D[0] * v[0] + D[3] * v[1] + D[6] * v[2];
    const double dq1 = D[1] * v[0] + D[4] * v[1] + D[7] * v[2];
    const double dq2 = D[2] * v[0] + D[5] * v[1] + D[8] * v[2];
    o.JpT[0][3 + i] = m00 * dq0 + m02 * dq2;
    o.JpT[1][3 + i] = m11 * dq1 + m12 * dq2;
    o.JpT[2][3 + i] = m22 * dq2;
  }
  o.JpT[0][6] = -L.ws * u * ifyt;
  o.JpT[1][6] = -L.ws * vv * ifyt;
  o.JpT[2][6] = 0.0;
  o.JDT2 = dr2dDb;
}

// Two kernels share the walk over a pair's pixels (GRID template flag):
//   GRID = 0, one workgroup per pair: the pose rows / columns -- PP (49 per-lane accumulators, folded at the end), GP
//            (G x 7) and PG (7 x G) in LDS.  This is the register-heavy half (both sides' 3 x 7 pose Jacobians).
//   GRID = 1, one workgroup per (pair, panel of the block's grid COLUMNS): the G x panel part of the grid x grid block.
//            Needs only the two depth rows and the taps of a constraint -- the pose Jacobians are dead code here -- so it
//            runs at twice the occupancy, which is what the second walk over the pixels costs.
// LDS: x of both frames, 2 frame constants, then PPs (56) + GP + PG, or GG (G x panelW).
template <int KD, bool GRID>
inline __global__ __launch_bounds__(kCrossThreads) void k_cross_assemble(Layout L, Table T, CrossPairs cp, const double* __restrict__ x,
                                                                  const FrameConst* __restrict__ fc, int panelW,
                                                                  double* __restrict__ X) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B, G = L.nD;  // (one value parameter per vertex: nD = vertices)
  const int pair = blockIdx.x, panel = GRID ? blockIdx.y : 0;
  const int v0 = GRID ? panel * panelW : 0, v1 = GRID ? ((v0 + panelW < G) ? v0 + panelW : G) : G, pw = v1 - v0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kCrossThreads / 64;
  double* xa = sm;
  double* xb = xa + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(xb + B);
  double* acc0 = reinterpret_cast<double*>(fcs + 2);
  double* PPs = acc0;                                  // GRID = 0: 49 (+ pad)
  double* GP = PPs + 56;                               //           G x 7
  double* PG = GP + static_cast<size_t>(G) * 7;        //           7 x G
  double* GG = acc0;                                   // GRID = 1: G x panelW
  const int fa = cp.fa[pair], fb = cp.fb[pair];
  for (int i = tid; i < B; i += kCrossThreads) {
    xa[i] = x[static_cast<size_t>(fa) * B + i];
    xb[i] = x[static_cast<size_t>(fb) * B + i];
  }
  constexpr int FCW = sizeof(FrameConst) / 8;
  for (int i = tid; i < 2 * FCW; i += kCrossThreads)
    reinterpret_cast<double*>(fcs)[i] = reinterpret_cast<const double*>(fc + (i < FCW ? fa : fb))[i % FCW];
  const int nLds = GRID ? G * panelW : 56 + G * 14;
  for (int i = tid; i < nLds; i += kCrossThreads) acc0[i] = 0.0;
  __syncthreads();

  double PP[GRID ? 1 : 49];
#pragma unroll
  for (int i = 0; i < (GRID ? 1 : 49); ++i) PP[i] = 0.0;
  for (int dir = 0; dir < 2; ++dir) {
    const long long cb = cp.range[pair * 4 + dir * 2], ce = cp.range[pair * 4 + dir * 2 + 1];
    if (cb >= ce) continue;
    const int fs = dir ? fb : fa, ft = dir ? fa : fb;
    const FrameConst& Fs = fcs[dir];
    const FrameConst& Ft = fcs[dir ^ 1];
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    // units of 64 x kCrossRun pixels per wave; a lane walks its own run of consecutive pixels
    constexpr long long kUnit = 64LL * kCrossRun;
    for (long long u0 = cb + wave * kUnit; u0 < ce; u0 += NW * kUnit) {
      const long long cFirst = u0 + static_cast<long long>(lane) * kCrossRun;
      const long long cStop = cFirst + kCrossRun < ce ? cFirst + kCrossRun : ce;
      // (GRID = 1: mask and flow of the lane's next pixel in flight; the pose half sits at its 256-register budget and
      // loses with them: 14.4 -> 16.0 ms)
      RecordStream<true> rs;
      const int iStop = static_cast<int>(cStop - u0);
      if constexpr (GRID) rs.prime(T, u0, static_cast<int>(cFirst - u0), iStop);
      for (long long c = cFirst; c < cStop; ++c) {
        float4 nd;
        float2 d;
        if constexpr (GRID) {
          if (!rs.take(T, u0, static_cast<int>(c - u0), 1, iStop, cb, fs, ft, nd, d)) continue;
        } else {
          if (!loadConstraint<true>(T, c, cb, fs, ft, nd, d)) continue;
        }
        CrossRows<KD> R;
        crossRows<KD>(L, Fs, Ft, xs, xt, nd, d, R);
        // row side = frame fa, column side = frame fb
        const double JDr[3] = {dir ? 0.0 : R.JDS[0], dir ? 0.0 : R.JDS[1], dir ? R.JDT2 : R.JDS[2]};
        const double JDc[3] = {dir ? R.JDS[0] : 0.0, dir ? R.JDS[1] : 0.0, dir ? R.JDS[2] : R.JDT2};
        const FastTaps<KD>& tr = dir ? R.tt : R.ts;
        const FastTaps<KD>& tc = dir ? R.ts : R.tt;
        const double dr = dir ? R.dt : R.ds, dc = dir ? R.ds : R.dt;
        const double w = R.w;
        if constexpr (GRID) {
          const double sDD = w * (JDr[0] * JDc[0] + JDr[1] * JDc[1] + JDr[2] * JDc[2]);
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            const int ir = tr.I(k);
            const double fr = sDD * tr.Wt(k) * dr;
#pragma unroll
            for (int l = 0; l < KD; ++l) {
              const int jc = tc.I(l) - v0;
              if (jc >= 0 && jc < pw) atomicAdd(&GG[ir * panelW + jc], fr * (tc.Wt(l) * dc));
            }
          }
        } else {
          const double(*Jr)[7] = dir ? R.JpT : R.JpS;  // pose rows of fa
          const double(*Jc)[7] = dir ? R.JpS : R.JpT;  // pose rows of fb
          double vA[7], vB[7];
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            const double a0 = w * Jr[0][i], a1 = w * Jr[1][i], a2 = w * Jr[2][i];
#pragma unroll
            for (int j = 0; j < 7; ++j) PP[i * 7 + j] += a0 * Jc[0][j] + a1 * Jc[1][j] + a2 * Jc[2][j];
            vA[i] = a0 * JDc[0] + a1 * JDc[1] + a2 * JDc[2];                        // pose_a x (depth of b)
            vB[i] = w * (JDr[0] * Jc[0][i] + JDr[1] * Jc[1][i] + JDr[2] * Jc[2][i]);  // (depth of a) x pose_b
          }
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            const int ir = tr.I(k), ic = tc.I(k);
            const double fr = tr.Wt(k) * dr, fcl = tc.Wt(k) * dc;
#pragma unroll
            for (int j = 0; j < 7; ++j) atomicAdd(&GP[ir * 7 + j], fr * vB[j]);
#pragma unroll
            for (int i = 0; i < 7; ++i) atomicAdd(&PG[i * G + ic], vA[i] * fcl);
          }
        }
      }
    }
  }
  if constexpr (!GRID) {
#pragma unroll
    for (int i = 0; i < 49; ++i) {
      const double s = waveSum(PP[i]);
      if (lane == 0) atomicAdd(&PPs[i], s);
    }
  }
  __syncthreads();
  // ---- flush into the pair's B x B block (row-major, rows = fa's unknowns)
  double* Xp = X + static_cast<size_t>(pair) * B * B;
  if constexpr (GRID) {
    for (int i = tid; i < G * pw; i += kCrossThreads) {
      const int r = i / pw, cidx = i - r * pw;
      Xp[static_cast<size_t>(7 + r) * B + 7 + v0 + cidx] = GG[r * panelW + cidx];
    }
  } else {
    for (int i = tid; i < 49; i += kCrossThreads) Xp[static_cast<size_t>(i / 7) * B + (i % 7)] = PPs[i];
    for (int i = tid; i < G * 7; i += kCrossThreads) Xp[static_cast<size_t>(7 + i / 7) * B + (i % 7)] = GP[i];
    for (int i = tid; i < 7 * G; i += kCrossThreads) Xp[static_cast<size_t>(i / G) * B + 7 + (i % G)] = PG[i];
  }
}

// q rows of one undirected pair from its block: y_a = X p_b, y_b = X^T p_a with p = (z + Z c + beta p_old) * mask (the
// search direction, formed here exactly as the matrix-free product forms it).  Each wave streams its rows once, fully
// coalesced: a row's dot product with p_b gives y_a[row], the same loads scaled by p_a[row] accumulate y_b per column.
inline __global__ __launch_bounds__(kCrossThreads) void k_cross_matvec(Layout L, CrossPairs cp, const double* __restrict__ X,
                                                                const double* __restrict__ mask, const double* __restrict__ z,
                                                                const double* __restrict__ pOld, const double* __restrict__ scal,
                                                                int useBeta, double* __restrict__ qPart, CoarseView V) {
  const double sDone = scal[S_DONE];
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kCrossThreads / 64;
  double* pa = sm;
  double* pb = pa + B;
  double* ya = pb + B;
  double* ybw = ya + B;          // NW x B partial column sums
  double* cl = ybw + NW * B;     // 2 x kCB coarse corrections
  double* tls = cl + 2 * kCB;    // third level (CoarseView::tl): the two frames' coefficients, 2 x tlS
  const int fa = cp.fa[pair], fb = cp.fb[pair];
  const double beta = useBeta ? scal[S_BETA] : 0.0;
  const bool tlOn = V.tl != nullptr;
  if (tid < 2 * kCB) cl[tid] = (V.cF != nullptr) ? V.cF[(tid < kCB ? fa : fb) * kCB + (tid & (kCB - 1))] : 0.0;
  if (tlOn)
    for (int i = tid; i < 2 * V.tlS; i += kCrossThreads) {
      const int which = i >= V.tlS ? 1 : 0;
      tls[i] = V.tl[static_cast<size_t>(which ? fb : fa) * V.tlS + (i - which * V.tlS)];
    }
  __syncthreads();
  if (sDone != 0.0) return;
  for (int i = tid; i < B; i += kCrossThreads) {
    const size_t ia = static_cast<size_t>(fa) * B + i, ib = static_cast<size_t>(fb) * B + i;
    double ca = coarseAtLds(cl, L, i), cb2 = coarseAtLds(cl + kCB, L, i);
    if (tlOn) {
      TlTaps tt;
      tlLoadTaps(V, i, L.nD, tt);
      ca += tlAt(tls, tt);
      cb2 += tlAt(tls + V.tlS, tt);
    }
    pa[i] = (z[ia] + ca + (useBeta ? beta * pOld[ia] : 0.0)) * mask[ia];
    pb[i] = (z[ib] + cb2 + (useBeta ? beta * pOld[ib] : 0.0)) * mask[ib];
  }
  __syncthreads();
  const double* Xp = X + static_cast<size_t>(pair) * B * B;
  constexpr int NC = 4;  // column chunks of 64 per lane: B <= 256
  double cb[NC] = {0.0, 0.0, 0.0, 0.0};
  double pbv[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) pbv[k] = (lane + 64 * k < B) ? pb[lane + 64 * k] : 0.0;
  for (int r0 = wave; r0 < B; r0 += 2 * NW) {
    // two rows per trip: their loads are in flight together
    const int r1 = r0 + NW;
    double xv0[NC], xv1[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int j = lane + 64 * k;
      xv0[k] = (j < B) ? Xp[static_cast<size_t>(r0) * B + j] : 0.0;
      xv1[k] = (j < B && r1 < B) ? Xp[static_cast<size_t>(r1) * B + j] : 0.0;
    }
    const double a0 = pa[r0], a1 = (r1 < B) ? pa[r1] : 0.0;
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      d0 += xv0[k] * pbv[k];
      d1 += xv1[k] * pbv[k];
      cb[k] += xv0[k] * a0 + xv1[k] * a1;
    }
    d0 = waveSum(d0);
    d1 = waveSum(d1);
    if (lane == 0) {
      ya[r0] = d0;
      if (r1 < B) ya[r1] = d1;
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k)
    if (lane + 64 * k < B) ybw[wave * B + lane + 64 * k] = cb[k];
  __syncthreads();
  double* outA = qPart + static_cast<size_t>(cp.slot[pair * 2]) * B;
  double* outB = qPart + static_cast<size_t>(cp.slot[pair * 2 + 1]) * B;
  for (int i = tid; i < B; i += kCrossThreads) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += ybw[w * B + i];
    outA[i] = ya[i];
    outB[i] = s;
  }
}

// Coarse edge blocks of the two-level preconditioner from the explicit blocks: E_ab = Z_a^T X_ab Z_b with Z = [I_7 0; 0 1]
// (the 8th mode moves every depth-scale vertex together), i.e. the pose x pose corner, row / column sums over the grid part
// and its total -- a reduction of the 250 KB block instead of a third walk over the pair's pixels (k_coarse_edges_fast:
// 9.9 ms per rebuild at 300 frames).  One workgroup per pair; rows coalesced, one wave per row.  Stored rows = fa, columns = fb
// like k_coarse_edges; `pairEdge` < 0: the pair has no edge block.
inline __global__ __launch_bounds__(256) void k_coarse_edges_cross(Layout L, CrossPairs cp, const double* __restrict__ X,
                                                            const int* __restrict__ pairEdge, double* __restrict__ edgeOut) {
  __shared__ double rowG[256];      // sum over the grid columns of row r
  __shared__ double colP[4][8];     // per wave: sum over the grid rows of pose column j
  const int B = L.B;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int edge = pairEdge[pair];
  if (edge < 0) return;
  const double* Xp = X + static_cast<size_t>(pair) * B * B;
  double cp0 = 0.0;  // lanes 0..6: column sums over the rows >= 7 handled by this wave
  for (int r = wave; r < B; r += 4) {
    double s = 0.0;
    for (int j = lane; j < B; j += 64) {
      const double v = Xp[static_cast<size_t>(r) * B + j];
      if (j >= 7) s += v;
      else if (r >= 7) cp0 += v;  // (j == lane < 7 here)
    }
    s = waveSum(s);
    if (lane == 0) rowG[r] = s;
  }
  if (lane < 8) colP[wave][lane] = lane < 7 ? cp0 : 0.0;
  __syncthreads();
  if (tid < kCBB) {
    const int i = tid >> 3, j = tid & 7;
    double v;
    if (i < 7 && j < 7) v = Xp[static_cast<size_t>(i) * B + j];
    else if (i < 7) v = rowG[i];
    else if (j < 7) v = (colP[0][j] + colP[1][j]) + (colP[2][j] + colP[3][j]);
    else {
      v = 0.0;
      for (int r = 7; r < B; ++r) v += rowG[r];
    }
    edgeOut[static_cast<size_t>(edge) * kCBB + tid] = v;
  }
}

}  // namespace cvd
