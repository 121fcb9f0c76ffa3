// oracle/block_sparse.h
//
// *** TEST INFRASTRUCTURE ONLY (part of the CPU oracle) -- never used by the product path. ***
//
// Block-sparse symmetric matrices on the FRAME GRAPH and an exact block-sparse Cholesky factorisation: the oracle's
// stand-in for the SPARSE_NORMAL_CHOLESKY solve the reference asks Ceres for (reference lib/PoseOptimizer.cpp:956).
// The normal equations J^T J of the optimizer couple two unknowns only when their frames share a residual block
// (a frame pair of the flow list, a smoothness triplet, a position-regulariser triple, frame 0 with shared
// intrinsics), so J^T J is stored as one dense block per coupled frame pair -- never as a dense n x n matrix -- and
// factorised block column by block column under a minimum-degree ordering of the frames.  Independent code: nothing
// here is shared with the device solver (which uses PCG).
#pragma once

#include <cstddef>
#include <memory>
#include <utility>
#include <vector>

namespace cvdo {

// Lower block triangle (block row >= block column) of a symmetric matrix whose unknowns are grouped by frame.
// Block (I, J), I >= J, is dense row-major size(I) x size(J); the diagonal blocks hold BOTH triangles.
struct BlockSym {
  int nb = 0;                      // block rows / columns (frames)
  std::vector<int> off;            // [nb + 1] scalar offset of every block row (size(I) = off[I+1] - off[I], may be 0)
  std::vector<int> rowPtr;         // [nb + 1] blocks of block ROW I: cols rowCol[rowPtr[I] .. rowPtr[I+1]) ascending, last = I
  std::vector<int> rowCol;
  std::vector<size_t> blkOff;      // value offset of every block (same index as rowCol)
  std::vector<double> val;

  int size(int I) const { return off[I + 1] - off[I]; }
  int n() const { return off.empty() ? 0 : off[nb]; }
  // index of block (I, J), I >= J, or -1
  int find(int I, int J) const;
  // structure from a list of coupled frame pairs (any order, duplicates allowed); values zeroed
  void build(const std::vector<int>& blockSizes, const std::vector<std::pair<int, int>>& pairs);
  void zero();
  // y = A x (full symmetric product)
  void multiply(const double* x, double* y) const;
  void diagonal(double* d) const;
};

// Exact Cholesky A = L L^T of a BlockSym under a fill-reducing (minimum degree) block ordering.
class BlockCholesky {
 public:
  // Symbolic phase: ordering, fill, storage.  Returns the number of floating-point operations of one numeric
  // factorisation (exact count for this structure).
  double analyze(const BlockSym& A);
  // Numeric phase on  S A S + diag(extraDiag)  (S = diag(scale); scale / extraDiag may be null).
  // false: a non-positive pivot was met (matrix not positive definite).
  bool factor(const BlockSym& A, const double* scale, const double* extraDiag, int numThreads);
  // b <- (L L^T)^-1 b
  void solve(double* b) const;
  size_t numBlocks() const { return lRow_.size(); }
  size_t numValues() const { return lsize_; }
  double flops() const { return flops_; }

 private:
  int nb_ = 0;
  std::vector<int> off_;           // scalar offsets in the ORIGINAL order
  std::vector<int> perm_, pos_;    // perm_[k] = original block at position k; pos_ = inverse
  std::vector<int> psize_, poff_;  // sizes / scalar offsets by position
  std::vector<int> colPtr_, lRow_; // L block column k: diagonal first, then rows ascending (positions)
  std::vector<size_t> lOff_;
  std::unique_ptr<double[]> lbuf_;  // L's values (uninitialised until factor(): first touch happens in parallel)
  size_t lsize_ = 0;
  double flops_ = 0.0;
  int findL(int i, int k) const;   // block index of L(i, k), i >= k, or -1
};

// dense helpers shared with the tests (row-major)
bool denseCholeskyInPlace(double* A, int n, int lda);   // lower factor in place; false if not positive definite

}  // namespace cvdo
