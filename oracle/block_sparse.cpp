// oracle/block_sparse.cpp
//
// *** TEST INFRASTRUCTURE ONLY (part of the CPU oracle) -- never used by the product path. ***
//
// Block-sparse symmetric storage + exact block-sparse Cholesky on the frame graph (see block_sparse.h).
// Compiled on its own with -O3 -ffp-contract=fast (the residual arithmetic in cvd_oracle.cpp keeps
// -ffp-contract=off); the dense block kernels are multi-versioned (AVX-512 / AVX2+FMA / baseline) and picked
// at load time, so the library built here runs on whatever host the GPU box has.
#include "block_sparse.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>
#include <stdexcept>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace cvdo {

#define CVDO_CLONES __attribute__((target_clones("avx512f", "arch=haswell", "default")))

typedef double v8d __attribute__((vector_size(64)));

static inline __attribute__((always_inline)) v8d loadu(const double* p) {
  v8d v;
  std::memcpy(&v, p, sizeof(v));
  return v;
}
static inline __attribute__((always_inline)) void storeu(double* p, v8d v) { std::memcpy(p, &v, sizeof(v)); }

// C[MR x 16] -= A[MR x k] * Bt[k x 16]
template <int MR>
static inline __attribute__((always_inline)) void microKernel(double* C, int ldc, const double* A, int lda,
                                                              const double* Bt, int ldb, int k) {
  v8d acc[MR][2];
  for (int r = 0; r < MR; ++r) { acc[r][0] = v8d{}; acc[r][1] = v8d{}; }
  for (int p = 0; p < k; ++p) {
    const v8d b0 = loadu(Bt + static_cast<size_t>(p) * ldb);
    const v8d b1 = loadu(Bt + static_cast<size_t>(p) * ldb + 8);
    for (int r = 0; r < MR; ++r) {
      const double a = A[static_cast<size_t>(r) * lda + p];
      acc[r][0] += a * b0;
      acc[r][1] += a * b1;
    }
  }
  for (int r = 0; r < MR; ++r) {
    double* c = C + static_cast<size_t>(r) * ldc;
    storeu(c, loadu(c) - acc[r][0]);
    storeu(c + 8, loadu(c + 8) - acc[r][1]);
  }
}

// C[m x n] -= A[m x k] * Bt[k x n]   (all row-major)
CVDO_CLONES
static void gemmSub(double* C, int ldc, const double* A, int lda, const double* Bt, int ldb, int m, int n, int k) {
  const int n16 = n & ~15;
  for (int j = 0; j < n16; j += 16) {
    int i = 0;
    for (; i + 6 <= m; i += 6) microKernel<6>(C + static_cast<size_t>(i) * ldc + j, ldc, A + static_cast<size_t>(i) * lda, lda, Bt + j, ldb, k);
    switch (m - i) {
      case 5: microKernel<5>(C + static_cast<size_t>(i) * ldc + j, ldc, A + static_cast<size_t>(i) * lda, lda, Bt + j, ldb, k); break;
      case 4: microKernel<4>(C + static_cast<size_t>(i) * ldc + j, ldc, A + static_cast<size_t>(i) * lda, lda, Bt + j, ldb, k); break;
      case 3: microKernel<3>(C + static_cast<size_t>(i) * ldc + j, ldc, A + static_cast<size_t>(i) * lda, lda, Bt + j, ldb, k); break;
      case 2: microKernel<2>(C + static_cast<size_t>(i) * ldc + j, ldc, A + static_cast<size_t>(i) * lda, lda, Bt + j, ldb, k); break;
      case 1: microKernel<1>(C + static_cast<size_t>(i) * ldc + j, ldc, A + static_cast<size_t>(i) * lda, lda, Bt + j, ldb, k); break;
      default: break;
    }
  }
  if (n16 < n) {  // ragged columns
    for (int i = 0; i < m; ++i) {
      const double* a = A + static_cast<size_t>(i) * lda;
      double* c = C + static_cast<size_t>(i) * ldc;
      for (int j = n16; j < n; ++j) {
        double s = 0.0;
        for (int p = 0; p < k; ++p) s += a[p] * Bt[static_cast<size_t>(p) * ldb + j];
        c[j] -= s;
      }
    }
  }
}

// In-place lower Cholesky of a dense row-major matrix (the strict upper triangle is ignored and left as is).
CVDO_CLONES
static bool potrfLower(double* A, int n, int lda) {
  for (int j = 0; j < n; ++j) {
    double* Aj = A + static_cast<size_t>(j) * lda;
    double d = Aj[j];
    for (int p = 0; p < j; ++p) d -= Aj[p] * Aj[p];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    d = std::sqrt(d);
    Aj[j] = d;
    const double inv = 1.0 / d;
    for (int i = j + 1; i < n; ++i) {
      double* Ai = A + static_cast<size_t>(i) * lda;
      double s = Ai[j];
      for (int p = 0; p < j; ++p) s -= Ai[p] * Aj[p];
      Ai[j] = s * inv;
    }
  }
  return true;
}
bool denseCholeskyInPlace(double* A, int n, int lda) { return potrfLower(A, n, lda); }

// rows of X (m x n, row-major) <- solve  x L^T = row   with L lower n x n:  x[c] = (row[c] - sum_{p<c} x[p] L[c][p]) / L[c][c]
CVDO_CLONES
static void trsmRows(double* X, int ldx, const double* L, int ldl, int m, int n) {
  for (int r = 0; r < m; ++r) {
    double* x = X + static_cast<size_t>(r) * ldx;
    for (int c = 0; c < n; ++c) {
      const double* Lc = L + static_cast<size_t>(c) * ldl;
      double s = x[c];
      for (int p = 0; p < c; ++p) s -= x[p] * Lc[p];
      x[c] = s / Lc[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// BlockSym
// ---------------------------------------------------------------------------------------------------
int BlockSym::find(int I, int J) const {
  const int* b = rowCol.data() + rowPtr[I];
  const int* e = rowCol.data() + rowPtr[I + 1];
  const int* it = std::lower_bound(b, e, J);
  return (it != e && *it == J) ? static_cast<int>(it - rowCol.data()) : -1;
}

void BlockSym::build(const std::vector<int>& blockSizes, const std::vector<std::pair<int, int>>& pairs) {
  nb = static_cast<int>(blockSizes.size());
  off.assign(nb + 1, 0);
  for (int i = 0; i < nb; ++i) off[i + 1] = off[i] + blockSizes[i];
  std::vector<std::set<int>> rows(nb);
  for (int i = 0; i < nb; ++i) rows[i].insert(i);
  for (const auto& p : pairs) {
    const int I = std::max(p.first, p.second), J = std::min(p.first, p.second);
    if (I < 0 || I >= nb || J < 0) throw std::runtime_error("BlockSym::build: pair out of range");
    rows[I].insert(J);
  }
  rowPtr.assign(nb + 1, 0);
  rowCol.clear();
  blkOff.clear();
  size_t total = 0;
  for (int I = 0; I < nb; ++I) {
    for (int J : rows[I]) {
      rowCol.push_back(J);
      blkOff.push_back(total);
      total += static_cast<size_t>(size(I)) * size(J);
    }
    rowPtr[I + 1] = static_cast<int>(rowCol.size());
  }
  val.assign(total, 0.0);
}

void BlockSym::zero() { std::fill(val.begin(), val.end(), 0.0); }

void BlockSym::multiply(const double* x, double* y) const {
  const int N = n();
  std::fill(y, y + N, 0.0);
  for (int I = 0; I < nb; ++I) {
    const int ni = size(I);
    for (int e = rowPtr[I]; e < rowPtr[I + 1]; ++e) {
      const int J = rowCol[e];
      const int nj = size(J);
      const double* B = val.data() + blkOff[e];
      const double* xj = x + off[J];
      double* yi = y + off[I];
      for (int a = 0; a < ni; ++a) {
        double s = 0.0;
        const double* Ba = B + static_cast<size_t>(a) * nj;
        for (int b = 0; b < nj; ++b) s += Ba[b] * xj[b];
        yi[a] += s;
      }
      if (J != I) {  // transpose part
        const double* xi = x + off[I];
        double* yj = y + off[J];
        for (int a = 0; a < ni; ++a) {
          const double xa = xi[a];
          const double* Ba = B + static_cast<size_t>(a) * nj;
          for (int b = 0; b < nj; ++b) yj[b] += Ba[b] * xa;
        }
      }
    }
  }
}

void BlockSym::diagonal(double* d) const {
  for (int I = 0; I < nb; ++I) {
    const int ni = size(I);
    const int e = rowPtr[I + 1] - 1;  // the diagonal block is the last of its row
    const double* B = val.data() + blkOff[e];
    for (int a = 0; a < ni; ++a) d[off[I] + a] = B[static_cast<size_t>(a) * ni + a];
  }
}

// ---------------------------------------------------------------------------------------------------
// BlockCholesky
// ---------------------------------------------------------------------------------------------------
int BlockCholesky::findL(int i, int k) const {
  const int* b = lRow_.data() + colPtr_[k];
  const int* e = lRow_.data() + colPtr_[k + 1];
  const int* it = std::lower_bound(b, e, i);
  return (it != e && *it == i) ? static_cast<int>(it - lRow_.data()) : -1;
}

double BlockCholesky::analyze(const BlockSym& A) {
  nb_ = A.nb;
  off_ = A.off;
  // frame graph (blocks of size 0 -- frames without free unknowns -- take no part)
  std::vector<std::set<int>> g(nb_);
  for (int I = 0; I < nb_; ++I)
    for (int e = A.rowPtr[I]; e < A.rowPtr[I + 1]; ++e) {
      const int J = A.rowCol[e];
      if (J == I || A.size(I) == 0 || A.size(J) == 0) continue;
      g[I].insert(J);
      g[J].insert(I);
    }
  // greedy minimum degree (weighted by nothing: the blocks have nearly equal sizes), ties by frame index
  perm_.clear();
  pos_.assign(nb_, -1);
  std::vector<std::vector<int>> structOf(nb_);  // by position: neighbours (original ids) alive at elimination
  std::vector<char> alive(nb_, 1);
  std::set<std::pair<int, int>> queue;  // (degree, frame)
  for (int v = 0; v < nb_; ++v) queue.insert({static_cast<int>(g[v].size()), v});
  while (!queue.empty()) {
    const int v = queue.begin()->second;
    queue.erase(queue.begin());
    pos_[v] = static_cast<int>(perm_.size());
    perm_.push_back(v);
    std::vector<int> nbv(g[v].begin(), g[v].end());
    structOf[pos_[v]] = nbv;
    for (int a : nbv) queue.erase({static_cast<int>(g[a].size()), a});
    for (int a : nbv) {
      g[a].erase(v);
      for (int b : nbv)
        if (a != b) g[a].insert(b);
    }
    for (int a : nbv) queue.insert({static_cast<int>(g[a].size()), a});
    alive[v] = 0;
    g[v].clear();
  }
  psize_.assign(nb_, 0);
  poff_.assign(nb_ + 1, 0);
  for (int k = 0; k < nb_; ++k) {
    psize_[k] = A.size(perm_[k]);
    poff_[k + 1] = poff_[k] + psize_[k];
  }
  colPtr_.assign(nb_ + 1, 0);
  lRow_.clear();
  lOff_.clear();
  size_t total = 0;
  flops_ = 0.0;
  for (int k = 0; k < nb_; ++k) {
    std::vector<int> rows;
    for (int fr : structOf[k]) rows.push_back(pos_[fr]);
    std::sort(rows.begin(), rows.end());
    lRow_.push_back(k);
    lOff_.push_back(total);
    total += static_cast<size_t>(psize_[k]) * psize_[k];
    const double nk = psize_[k];
    flops_ += nk * nk * nk / 3.0;
    double below = 0.0, belowSq = 0.0;
    for (int i : rows) {
      lRow_.push_back(i);
      lOff_.push_back(total);
      total += static_cast<size_t>(psize_[i]) * psize_[k];
      below += psize_[i];
      belowSq += static_cast<double>(psize_[i]) * psize_[i];
    }
    flops_ += below * nk * nk;                          // triangular solves
    flops_ += nk * (below * below + belowSq);           // trailing update: 2 nk sum_{i >= j} n_i n_j
    colPtr_[k + 1] = static_cast<int>(lRow_.size());
  }
  lbuf_.reset(new double[std::max<size_t>(total, 1)]);
  lsize_ = total;
  return flops_;
}

bool BlockCholesky::factor(const BlockSym& A, const double* scale, const double* extraDiag, int numThreads) {
  if (A.nb != nb_) throw std::runtime_error("BlockCholesky::factor: analyze() was run for another structure");
  double* const lval = lbuf_.get();
  const int T = std::max(1, numThreads);
  {
    const long long nChunks = static_cast<long long>((lsize_ + (1u << 20) - 1) >> 20);
#pragma omp parallel for schedule(static) num_threads(T)
    for (long long c = 0; c < nChunks; ++c) {
      const size_t b0 = static_cast<size_t>(c) << 20, b1 = std::min(lsize_, b0 + (size_t(1) << 20));
      std::memset(lval + b0, 0, (b1 - b0) * sizeof(double));
    }
  }
  // scatter S A S (+ diag) into L's storage, permuted (a block whose position order is reversed is transposed)
  for (int I = 0; I < nb_; ++I) {
    const int ni = A.size(I);
    for (int e = A.rowPtr[I]; e < A.rowPtr[I + 1]; ++e) {
      const int J = A.rowCol[e];
      const int nj = A.size(J);
      if (ni == 0 || nj == 0) continue;
      const double* B = A.val.data() + A.blkOff[e];
      const int pi = pos_[I], pj = pos_[J];
      const double* si = scale ? scale + A.off[I] : nullptr;
      const double* sj = scale ? scale + A.off[J] : nullptr;
      if (pi >= pj) {
        const int lb = findL(pi, pj);
        double* D = lval + lOff_[lb];
        for (int a = 0; a < ni; ++a)
          for (int b = 0; b < nj; ++b)
            D[static_cast<size_t>(a) * nj + b] = B[static_cast<size_t>(a) * nj + b] * (si ? si[a] * sj[b] : 1.0);
        if (I == J && extraDiag)
          for (int a = 0; a < ni; ++a) D[static_cast<size_t>(a) * ni + a] += extraDiag[A.off[I] + a];
      } else {
        const int lb = findL(pj, pi);
        double* D = lval + lOff_[lb];
        for (int a = 0; a < ni; ++a)
          for (int b = 0; b < nj; ++b)
            D[static_cast<size_t>(b) * ni + a] = B[static_cast<size_t>(a) * nj + b] * (si ? si[a] * sj[b] : 1.0);
      }
    }
  }
  std::vector<double> Lt;  // transposes of column k's blocks below the diagonal: for block j, [nk x nj]
  std::vector<size_t> ltOff;
  bool ok = true;
  for (int k = 0; k < nb_ && ok; ++k) {
    const int nk = psize_[k];
    if (nk == 0) continue;
    const int c0 = colPtr_[k], c1 = colPtr_[k + 1];
    double* Lkk = lval + lOff_[c0];
    if (!potrfLower(Lkk, nk, nk)) { ok = false; break; }
    const int nBelow = c1 - c0 - 1;
    if (nBelow == 0) continue;
    // L_ik = A_ik L_kk^-T, parallel over (block, row chunk)
    ltOff.assign(nBelow + 1, 0);
    for (int b = 0; b < nBelow; ++b) ltOff[b + 1] = ltOff[b] + static_cast<size_t>(nk) * psize_[lRow_[c0 + 1 + b]];
    Lt.resize(ltOff[nBelow]);
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
    for (int b = 0; b < nBelow; ++b) {
      const int i = lRow_[c0 + 1 + b];
      const int ni = psize_[i];
      double* X = lval + lOff_[c0 + 1 + b];
      trsmRows(X, nk, Lkk, nk, ni, nk);
      double* Xt = Lt.data() + ltOff[b];
      for (int a = 0; a < ni; ++a)
        for (int p = 0; p < nk; ++p) Xt[static_cast<size_t>(p) * ni + a] = X[static_cast<size_t>(a) * nk + p];
    }
    // trailing update: A_ij -= L_ik L_jk^T for all i >= j in the column's structure (distinct targets)
    const long long nPairs = static_cast<long long>(nBelow) * (nBelow + 1) / 2;
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
    for (long long q = 0; q < nPairs; ++q) {
      // q -> (bi >= bj) by triangular index
      int bi = static_cast<int>((std::sqrt(8.0 * static_cast<double>(q) + 1.0) - 1.0) / 2.0);
      while (static_cast<long long>(bi) * (bi + 1) / 2 > q) --bi;
      while (static_cast<long long>(bi + 1) * (bi + 2) / 2 <= q) ++bi;
      const int bj = static_cast<int>(q - static_cast<long long>(bi) * (bi + 1) / 2);
      const int i = lRow_[c0 + 1 + bi], j = lRow_[c0 + 1 + bj];
      const int ni = psize_[i], nj = psize_[j];
      if (ni == 0 || nj == 0) continue;
      const int tb = findL(i, j);
      double* Cij = lval + lOff_[tb];
      gemmSub(Cij, nj, lval + lOff_[c0 + 1 + bi], nk, Lt.data() + ltOff[bj], nj, ni, nj, nk);
    }
  }
  return ok;
}

void BlockCholesky::solve(double* b) const {
  const double* const lval = lbuf_.get();
  const int N = poff_[nb_];
  std::vector<double> y(N);
  for (int k = 0; k < nb_; ++k)
    for (int a = 0; a < psize_[k]; ++a) y[poff_[k] + a] = b[off_[perm_[k]] + a];
  // forward: L y = b
  for (int k = 0; k < nb_; ++k) {
    const int nk = psize_[k];
    if (nk == 0) continue;
    const int c0 = colPtr_[k], c1 = colPtr_[k + 1];
    const double* Lkk = lval + lOff_[c0];
    double* yk = y.data() + poff_[k];
    for (int c = 0; c < nk; ++c) {
      double s = yk[c];
      const double* Lc = Lkk + static_cast<size_t>(c) * nk;
      for (int p = 0; p < c; ++p) s -= Lc[p] * yk[p];
      yk[c] = s / Lc[c];
    }
    for (int e = c0 + 1; e < c1; ++e) {
      const int i = lRow_[e];
      const int ni = psize_[i];
      const double* Lik = lval + lOff_[e];
      double* yi = y.data() + poff_[i];
      for (int a = 0; a < ni; ++a) {
        double s = 0.0;
        const double* La = Lik + static_cast<size_t>(a) * nk;
        for (int p = 0; p < nk; ++p) s += La[p] * yk[p];
        yi[a] -= s;
      }
    }
  }
  // backward: L^T x = y
  for (int k = nb_ - 1; k >= 0; --k) {
    const int nk = psize_[k];
    if (nk == 0) continue;
    const int c0 = colPtr_[k], c1 = colPtr_[k + 1];
    double* yk = y.data() + poff_[k];
    for (int e = c0 + 1; e < c1; ++e) {
      const int i = lRow_[e];
      const int ni = psize_[i];
      const double* Lik = lval + lOff_[e];
      const double* yi = y.data() + poff_[i];
      for (int a = 0; a < ni; ++a) {
        const double ya = yi[a];
        const double* La = Lik + static_cast<size_t>(a) * nk;
        for (int p = 0; p < nk; ++p) yk[p] -= La[p] * ya;
      }
    }
    const double* Lkk = lval + lOff_[c0];
    for (int c = nk - 1; c >= 0; --c) {
      double s = yk[c];
      for (int p = c + 1; p < nk; ++p) s -= Lkk[static_cast<size_t>(p) * nk + c] * yk[p];
      yk[c] = s / Lkk[static_cast<size_t>(c) * nk + c];
    }
  }
  for (int k = 0; k < nb_; ++k)
    for (int a = 0; a < psize_[k]; ++a) b[off_[perm_[k]] + a] = y[poff_[k] + a];
}

}  // namespace cvdo
