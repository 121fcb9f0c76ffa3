// cvd_hip.hip -- host side of libcvd_hip.so: device memory, the Levenberg-Marquardt driver, the
// block-Jacobi PCG linear solve, the coarse-to-fine schedule and the C ABI of include/cvd_hip.h.
//
// Mirrors (file:line relative to the reference root, facebookresearch/robust_cvd):
//   DepthVideoPoseOptimizer ctor / pose write-back   lib/PoseOptimizer.cpp:748-783, 964-987
//   poseOptimization (CTF schedule, deferred spatial) lib/PoseOptimizer.cpp:788-888
//   poseOptimizationStep (which terms, which blocks constant)  lib/PoseOptimizer.cpp:890-952
//   normalizeDepth                                      lib/PoseOptimizer.cpp:992-1147
//   gridXformSplit / resetPoses                         lib/Processor.cpp:888-1003
//   trust-region loop = Ceres defaults (the reference only sets solver type / max iterations / threads,
//   lib/PoseOptimizer.cpp:955-961); the sparse Cholesky is replaced by PCG.
//
// gfx950 only. There is NO CPU path: every entry point that computes fails if no HIP device is usable.

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cvd_hip.h"
#include "cvd_kernels.h"
#include "cvd_coarse.h"
#include "cvd_cross.h"
#include "cvd_triplets.h"
#include "cvd_dense.h"
#include "cvd_sampling.h"
#include "cvd_imageops.h"
#include "cvd_filter.h"

#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace cvd {

static std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return buf;
}

#define HIP_CHECK(expr)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      throw std::runtime_error(fmt("HIP error %s at %s:%d: %s", hipGetErrorName(e_), __FILE__,    \
                                   __LINE__, hipGetErrorString(e_)));                            \
  } while (0)

#define NCCL_CHECK(expr)                                                                         \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess)                                                                       \
      throw std::runtime_error(fmt("RCCL error %s at %s:%d", ncclGetErrorString(r_), __FILE__, __LINE__)); \
  } while (0)

static double nowSeconds() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void ensure(size_t count) {
    if (count <= n) return;
    release();
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)));
    n = count;
  }
  void upload(const T* src, size_t count, hipStream_t s) {
    ensure(count);
    if (count) HIP_CHECK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T* dst, size_t count, hipStream_t s) const {
    if (count) HIP_CHECK(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
  }
};

// ---- transform bookkeeping on the host (Xform::params_ layout, reference lib/DepthMapTransform.cpp:526-534,
// 702-707, 1102, 1176, 1356-1357) ---------------------------------------------------------------------
static int valueNumParams(int t) {
  if (t == CVD_VALUE_SCALE) return 1;
  if (t == CVD_VALUE_SCALE_SHIFT) return 2;
  throw std::runtime_error("Invalid value transform.");
}
static int xformBlockSize(const cvd_xform_desc& d) {
  if (d.type == CVD_XFORM_DEPTH) return d.depth_type == CVD_DEPTH_IDENTITY ? 0 : valueNumParams(d.value_xform);
  return d.spatial_type == CVD_SPATIAL_IDENTITY ? 0 : 2;
}
static int xformNumBlocks(const cvd_xform_desc& d) {
  if (d.type == CVD_XFORM_DEPTH) {
    switch (d.depth_type) {
      case CVD_DEPTH_IDENTITY: return 0;
      case CVD_DEPTH_GLOBAL: return 1;
      case CVD_DEPTH_GRID: {
        const int gx = d.grid_size[0], gy = d.grid_size[1], gz = d.grid_size[2];
        if (gx > 1 || gy > 1)
          if (gx < 2 || gy < 2)
            throw std::runtime_error(
                "Spatial grid transforms must have at least two rows and columns, respectively.");
        if (valueNumParams(d.value_xform) * gx * gy * gz <= 1)
          throw std::runtime_error("Grid transform cannot have an empty grid.");
        return gx * gy * gz;
      }
      default: throw std::runtime_error("Invalid depth transform type.");
    }
  } else if (d.type == CVD_XFORM_SPATIAL) {
    switch (d.spatial_type) {
      case CVD_SPATIAL_IDENTITY: return 0;
      case CVD_SPATIAL_VERTICAL_LINEAR: return 2;
      case CVD_SPATIAL_CORNERS_BILINEAR: return 4;
      case CVD_SPATIAL_BILINEAR_GRID:
      case CVD_SPATIAL_BICUBIC_GRID:
        if (d.grid_size[1] < 2 || d.grid_size[0] < 2)
          throw std::logic_error("Need at least two rows and columns in depth transform grid.");
        return d.grid_size[0] * d.grid_size[1];
      default: throw std::runtime_error("Invalid spatial transform type.");
    }
  }
  throw std::runtime_error("Invalid transform type.");
}

// ---- rotation conversions on the host (ceres/rotation.h + Eigen semantics) -------------
static void quatToMatrix(const double q[4] /*x,y,z,w*/, double R[3][3]) {
  // columns = q * e_x, q * e_y, q * e_z with Eigen's v + w*uv + qv x uv, uv = 2 qv x v
  for (int c = 0; c < 3; ++c) {
    double v[3] = {0, 0, 0};
    v[c] = 1.0;
    const double uv[3] = {2.0 * (q[1] * v[2] - q[2] * v[1]), 2.0 * (q[2] * v[0] - q[0] * v[2]),
                          2.0 * (q[0] * v[1] - q[1] * v[0])};
    R[0][c] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    R[1][c] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    R[2][c] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
  }
}
static void matrixToAngleAxis(const double R[3][3], double aa[3]) {
  double q0, q1, q2, q3;  // w, x, y, z
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr >= 0.0) {
    double t = std::sqrt(tr + 1.0);
    q0 = 0.5 * t;
    t = 0.5 / t;
    q1 = (R[2][1] - R[1][2]) * t;
    q2 = (R[0][2] - R[2][0]) * t;
    q3 = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    double qq[4];
    qq[i + 1] = 0.5 * t;
    t = 0.5 / t;
    qq[0] = (R[k][j] - R[j][k]) * t;
    qq[j + 1] = (R[j][i] + R[i][j]) * t;
    qq[k + 1] = (R[k][i] + R[i][k]) * t;
    q0 = qq[0]; q1 = qq[1]; q2 = qq[2]; q3 = qq[3];
  }
  const double s2 = q1 * q1 + q2 * q2 + q3 * q3;
  if (s2 > 0.0) {
    const double s = std::sqrt(s2);
    const double two = 2.0 * ((q0 < 0.0) ? std::atan2(-s, -q0) : std::atan2(s, q0));
    const double k = two / s;
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  } else {
    aa[0] = q1 * 2.0; aa[1] = q2 * 2.0; aa[2] = q3 * 2.0;
  }
}
static void angleAxisToMatrix(const double aa[3], double R[3][3]) {
  const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > std::numeric_limits<double>::epsilon()) {
    const double th = std::sqrt(th2);
    const double wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
    const double c = std::cos(th), s = std::sin(th);
    R[0][0] = c + wx * wx * (1 - c);       R[1][0] = wz * s + wx * wy * (1 - c);  R[2][0] = -wy * s + wx * wz * (1 - c);
    R[0][1] = wx * wy * (1 - c) - wz * s;  R[1][1] = c + wy * wy * (1 - c);       R[2][1] = wx * s + wy * wz * (1 - c);
    R[0][2] = wy * s + wx * wz * (1 - c);  R[1][2] = -wx * s + wy * wz * (1 - c); R[2][2] = c + wz * wz * (1 - c);
  } else {
    R[0][0] = 1;      R[1][0] = aa[2];  R[2][0] = -aa[1];
    R[0][1] = -aa[2]; R[1][1] = 1;      R[2][1] = aa[0];
    R[0][2] = aa[1];  R[1][2] = -aa[0]; R[2][2] = 1;
  }
}
static void matrixToQuat(const double R[3][3], double q[4] /*x,y,z,w*/) {
  double t = R[0][0] + R[1][1] + R[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t;
    q[1] = (R[0][2] - R[2][0]) * t;
    q[2] = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t;
    q[j] = (R[j][i] + R[i][j]) * t;
    q[k] = (R[k][i] + R[i][k]) * t;
  }
}

enum KernelClass { KC_ASSEMBLE = 0, KC_MATVEC_PAIRS, KC_MATVEC_FINISH, KC_CG_UPDATE, KC_INVERSE, KC_COST, KC_COUNT,
                   // exchange steps of the pair-sharded multi-GPU mode (cvd_get_comm_times): timed whenever any class is
                   KC_COMM_EVAL = KC_COUNT, KC_COMM_PRODUCT, KC_COMM_COARSE, KC_TOTAL };

// A helper host thread for work that is many small enqueues on the SIDE stream (the dense coarse level's rocSOLVER
// inversion is ~250 kernel launches, ~2.3 ms of host time): submitted there, the main thread goes on enqueuing the
// PCG and the device never idles between the block inverse and the first product.  One job at a time.
class SideWorker {
 public:
  ~SideWorker() {
    {
      std::lock_guard<std::mutex> g(m_);
      quit_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  void submit(std::function<void()> job) {
    wait();
    std::lock_guard<std::mutex> g(m_);
    if (!th_.joinable()) th_ = std::thread([this]() { run(); });
    job_ = std::move(job);
    busy_ = true;
    cv_.notify_all();
  }
  // returns once the submitted job has finished ENQUEUING; rethrows what it threw
  void wait() {
    std::unique_lock<std::mutex> g(m_);
    cv_.wait(g, [this]() { return !busy_; });
    if (err_) {
      std::exception_ptr e = err_;
      err_ = nullptr;
      std::rethrow_exception(e);
    }
  }
  void waitNoThrow() noexcept {
    try { wait(); } catch (...) {}
  }

 private:
  void run() {
    std::unique_lock<std::mutex> g(m_);
    while (true) {
      cv_.wait(g, [this]() { return quit_ || (busy_ && job_); });
      if (quit_) return;
      std::function<void()> job = std::move(job_);
      job_ = nullptr;
      g.unlock();
      std::exception_ptr e;
      try { job(); } catch (...) { e = std::current_exception(); }
      g.lock();
      err_ = e;
      busy_ = false;
      cv_.notify_all();
    }
  }
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool busy_ = false, quit_ = false;
  std::exception_ptr err_;
};

struct Ceres {  // ceres::Solver::Options defaults used on this path
  static constexpr double initial_radius = 1e4, max_radius = 1e16, min_radius = 1e-32;
  static constexpr double min_relative_decrease = 1e-3;
  static constexpr double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  static constexpr int max_consecutive_invalid = 5;
};

enum ProblemKind { PK_POSE_STEP = 0, PK_NORMALIZE = 1 };

}  // namespace cvd

using namespace cvd;

struct cvd_handle_t {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  cvd_solver_options opt{};

  // video
  int F = 0, W = 0, H = 0;
  float aspect = 1.f, invAspect = 1.f;
  DevBuf<float> dDepth;
  std::vector<float> median;
  DevBuf<float> dMedian;
  bool medianDirty = true;

  // constraints
  int P = 0;
  long long C = 0;
  std::vector<int> pairA, pairB;
  std::vector<long long> pairOff;
  DevBuf<int> dPairA, dPairB, dCPair;
  DevBuf<long long> dPairOff;
  DevBuf<float4> dLoc, dNdc;
  DevBuf<float2> dDsrc;
  DevBuf<unsigned char> dStatic, dInRange, dRegOwner;
  // dense mode (cvd_set_pair_flows): flow / mask images of every directed pair instead of a constraint list
  bool dense = false;
  DevBuf<float2> dFlow;
  DevBuf<unsigned char> dFMask;

  // multi-GPU (pair-sharded): one RCCL communicator, this rank owns the regularisers of frames f % world == rank
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  bool distForced = false;  // test hook (CVD_FORCE_DIST with a 1-rank communicator): run the multi-rank code path
  bool dist() const { return world > 1 || distForced; }
  // Frame ownership of the sharded mode: rank r owns the contiguous chunk [r Fc, (r + 1) Fc), Fc = ceil(F / world): it
  // receives the reduced H_ff of those frames (reduce-scatter), inverts them and all-gathers the f32 inverses.
  int ownChunk() const { return (F + world - 1) / world; }
  int ownFirst() const { return std::min(F, rank * ownChunk()); }
  int ownCount() const { return std::min(F, (rank + 1) * ownChunk()) - ownFirst(); }
  int framesPadded() const { return dist() ? ownChunk() * world : F; }
  bool haveTriplets = false;
  // scene-flow smoothness triplets (cvd_triplets.h): groups keyed by the centre frame
  std::vector<int> tripCenter;
  std::vector<long long> tripOff;
  long long tripC = 0;
  DevBuf<float> dTLoc, dTDsrc;
  DevBuf<float2> dTNdc;
  DevBuf<unsigned char> dTStatic;
  DevBuf<int> dTGroupOfC, dTCenterAll, dTCenter, dTSlot, dFtOff, dFtList;
  DevBuf<long long> dTOff;
  DevBuf<double> dCostTrip;
  std::vector<int> tripActive;      // groups this rank evaluates for the compiled range
  bool tableWithTriplets = false;   // the compiled work decomposition includes the triplet rows
  long long numValidTrip = 0;
  int qRows = 0;                    // rows of the partial-product buffer (2 per pair item + 3 per triplet group)

  // work decomposition
  std::vector<int> itemFa, itemFb;
  std::vector<long long> itemRange;  // 4 per item
  DevBuf<int> dItemFa, dItemFb, dItemSlot, dFiOff, dFiList, dFpOff, dFpList;
  // k_assemble_fast work list (AsmWork): parts sorted longest first, units, partial-block slots of split frames
  DevBuf<AsmPart> dAsmParts;
  DevBuf<int2> dAsmUnits;
  DevBuf<double> dAsmScratch;
  DevBuf<unsigned int> dAsmCount;
  int nAsmParts = 0, nAsmSlots = 0;
  int numCU = 256;
  hipStream_t stream2 = nullptr;                       // side stream of the asynchronous coarse rebuild
  hipStream_t streamCapture = nullptr;                 // capture-only stream of its hipGraph (never executes anything)
  SideWorker sideWorker;                               // host thread that enqueues the dense rebuild there
  rocblas_handle rbMain = nullptr;                     // main-stream handle (batched block inverses beyond B = 256)
  DevBuf<double> dInvScratch;
  DevBuf<int> dInvInfo;
  hipEvent_t evCoarseIn = nullptr, evCoarseDone = nullptr, evCoarseRead = nullptr;  // (evCoarseRead: the rebuild has consumed H, lam, x)
  DevBuf<FrameConst> dFc2;                             // its own frame constants (the main stream rewrites dFc)
  DevBuf<long long> dItemRange;
  // explicit cross blocks of the dense mode (cvd_cross.h): undirected pairs, their rows, the blocks
  std::vector<int> xFa, xFb;
  DevBuf<int> dXFa, dXFb, dXSlot, dXFiOff, dXPairEdge;
  DevBuf<long long> dXRange;
  DevBuf<double> dXBlocks;
  DevBuf<unsigned int> dCounters;  // [0] k_matvec_finish, [1] k_cg_update (last-workgroup tickets)
  std::vector<unsigned char> tableRange;  // range the table / items were compiled for
  bool tableValid = false;
  bool tableIgnoresStatic = false;  // compiled for normalizeDepth's pair loop (every constraint, dynamic ones included)
  long long numValid = 0;

  // state
  std::vector<cvd_frame_pose> poses;
  cvd_xform_desc ddesc{}, sdesc{};
  std::vector<double> dparams, sparams;  // F x nD, F x nS
  std::vector<std::array<double, 7>> poseParams;
  bool poseParamsValid = false;

  // solver buffers
  DevBuf<double> dX, dXc, dG, dLam, dMask, dScale, dDx, dR, dR1, dZ, dP0, dP1, dQ, dH, dQPart;
  DevBuf<float> dMinv;
  DevBuf<double> dFdot, dCostItem, dCostFrame, dScal, dHd, dFocal;
  DevBuf<double> dStatPart;  // per-workgroup partials of k_step_stats
  DevBuf<double> dDense;     // output of the dense consumer kernels (cvd_dense.h)
  DevBuf<float> dImgIn, dImgGray, dImgCov, dImgOut;  // cvd_imageops.h staging
  DevBuf<unsigned char> dImgMask;
  DevBuf<unsigned int> dImgTmp;
  // AdaptiveDeformationCost: dynamic masks of all frames (cvd_set_dynamic_masks) and the vertex weights of the
  // current depth grid
  DevBuf<unsigned char> dDynMask;
  DevBuf<double> dAdaptW;
  int dynW = 0, dynH = 0, adaptGx = 0, adaptGy = 0;
  bool haveDynMasks = false;
  DevBuf<float> dFltDepth, dFltOut, dFltFlowF, dFltFlowB;  // cvd_filter.h staging
  DevBuf<unsigned char> dFltMaskF, dFltMaskB;
  DevBuf<FilterCam> dFltCams;
  // constraint sampling (cvd_sampling.h): result of the last cvd_sample_pair_constraints
  DevBuf<float2> dSampledLoc, dSampledTrip;  // 2 resp. 3 float2 per constraint
  std::vector<long long> sampledOff, sampledTripOff;

  // coarse (pose-graph) level of the two-level preconditioner (cvd_coarse.h)
  struct CoarseHost {
    bool valid = false;   // plan built for the current table
    int nEdges = 0, nBlocks = 0, nLevels = 0;
    std::vector<int> itemEdge;
    DevBuf<int> order, pos, levelPtr, levelCols, lvlBlkPtr, lvlBlks, blkCol, blkRow, colPtr, rowPtr, rowBlk, updPtr, updBlk,
        updA, updB, edgeBlk, edgeFa, edgeFb, wPtr, wRow, wtPtr, wtBlk, wtCol, wtFrame, wuPtr, wuL, wuW, itemEdgeDev, wSlot;
    DevBuf<double> edges, diag, Lb, Linv, Wb, rc, qc, y, c, dotPart, fdotY, wq, dropDiag;
    bool sparsified = false;  // some frame pairs were left out of the coarse graph (sparsifyCoarseGraph)
    // dense variant (cvd_coarse.h "DENSE coarse level"): A_c^-1 as a full f32 matrix, two buffers for the side-stream rebuild
    bool denseMode = false;
    hipGraphExec_t denseGraph = nullptr;  // the side stream's assemble + potrf + potri sequence (launchCoarseSetup)
    std::array<const void*, 8> denseGraphKey{};
    int denseGraphState = 0;              // 0 first direct call still to come, 1 capture allowed, -1 capture unsupported
    int denseForB = 0;        // frame-block size of the problem that inverse was built for (a coarse-to-fine level)
    bool denseReady = false;  // denseInv holds an inverse for this plan (possibly of an earlier solve: a usable, stale preconditioner)
    DevBuf<double> denseA;
    DevBuf<float> denseInv, denseInv2;
    DevBuf<int> denseInfo;
    rocblas_handle rb[2] = {nullptr, nullptr};  // [0] main stream, [1] side stream
    int nW = 0;
    DevBuf<unsigned char> modeActive;
    DevBuf<int> fail;
    DevBuf<unsigned int> barrier;  // grid barrier of k_coarse_factor_mw
    // second set of the factor's outputs: a rebuild runs on a side stream while the PCG of the same LM iteration
    // still uses the previous factor (launchCoarseSetup / the LM loop)
    DevBuf<double> Wb2;
    DevBuf<int> fail2;
    CoarsePlan plan{};
  } coarse;
  bool coarseOn = false;  // this solve uses the coarse level
  // frame-pair graph of the WHOLE problem (cvd_set_pair_graph): in the pair-sharded multi-GPU mode every rank must
  // build the same elimination plan although it only holds its own pairs
  std::vector<std::pair<int, int>> globalEdges;
  bool haveGlobalEdges = false;
  DevBuf<double> dRegJac;  // regulariser Jacobian rows of the current linearisation point (RegCache)
  DevBuf<unsigned short> dRegCol;
  DevBuf<unsigned char> dRegCnt;
  RegCache regCache{};
  DevBuf<FrameConst> dFc;
  DevBuf<int> dFail;
  DevBuf<unsigned long long> dCount;
  double* hScal = nullptr;  // pinned
  double* hStage[2] = {nullptr, nullptr};  // pinned staging for the per-solve state / mask transfers
  size_t hStageN[2] = {0, 0};
  double* hPcg = nullptr;   // pinned, device-written PCG progress mirror: [0] iterations + 1, [1..8] done-flag ring
  hipEvent_t pcgEvent[2] = {nullptr, nullptr};

  // results
  cvd_solve_summary summary{};
  std::vector<cvd_iteration_record> records;

  bool forceGeneric = false;  // test hook: route the products through the generic (all-variants) kernel

  // kernel timing
  int timing = 0;  // bit mask of KernelClass values to time with HIP events
  int timingStride = 1;        // hipExtLaunchKernelGGL event pairs (tReserve) on every timingStride-th launch only
  long long timingCounter = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evPool;
  std::vector<int> evClass;
  std::vector<int> evIter;  // PCG iteration the launch belongs to (-1 outside PCG): launches enqueued past
  int curPcgIter = -1;      // convergence are no-ops and are dropped from the statistics (tDropFrom)
  size_t evUsed = 0;
  double kcMs[KC_TOTAL] = {0};
  long long kcN[KC_TOTAL] = {0};

  ~cvd_handle_t() {
    sideWorker.waitNoThrow();
    for (auto& e : evPool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (coarse.denseGraph) (void)hipGraphExecDestroy(coarse.denseGraph);
    if (rbMain) (void)rocblas_destroy_handle(rbMain);
    for (auto& rbh : coarse.rb) if (rbh) (void)rocblas_destroy_handle(rbh);
    if (comm) (void)ncclCommDestroy(comm);
    if (hScal) (void)hipHostFree(hScal);
    for (auto& p : hStage) if (p) (void)hipHostFree(p);
    if (hPcg) (void)hipHostFree(hPcg);
    for (auto& e : pcgEvent) if (e) (void)hipEventDestroy(e);
    if (evCoarseIn) (void)hipEventDestroy(evCoarseIn);
    if (evCoarseDone) (void)hipEventDestroy(evCoarseDone);
    if (evCoarseRead) (void)hipEventDestroy(evCoarseRead);
    if (streamCapture) (void)hipStreamDestroy(streamCapture);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream) (void)hipStreamDestroy(stream);
  }

  int nD() const { return xformNumBlocks(ddesc) * xformBlockSize(ddesc); }
  int nS() const { return xformNumBlocks(sdesc) * xformBlockSize(sdesc); }
  int Bsz() const { return 7 + nD() + nS(); }

  // ---- timing helpers --------------------------------------------------------------------------------
  int tBegin(int kc) {
    if (kc >= KC_COUNT ? timing == 0 : !(timing & (1 << kc))) return -1;
    if (evUsed == evPool.size()) {
      hipEvent_t a, b;
      HIP_CHECK(hipEventCreate(&a));
      HIP_CHECK(hipEventCreate(&b));
      evPool.emplace_back(a, b);
      evClass.push_back(kc);
      evIter.push_back(-1);
    }
    evClass[evUsed] = kc;
    evIter[evUsed] = curPcgIter;
    HIP_CHECK(hipEventRecord(evPool[evUsed].first, stream));
    return static_cast<int>(evUsed++);
  }
  void tEnd(int slot) {
    if (slot >= 0) HIP_CHECK(hipEventRecord(evPool[slot].second, stream));
  }
  // Event pair for hipExtLaunchKernelGGL(start, stop): the events take the kernel's own begin / end time stamps
  // (what rocprofv3 --kernel-trace reports), without the dispatch gap a record-before / record-after pair includes.
  int tReserve(int kc, hipEvent_t& start, hipEvent_t& stop) {
    start = nullptr;
    stop = nullptr;
    if (!(timing & (1 << kc))) return -1;
    if (timingStride > 1 && (timingCounter++ % timingStride) != 0) return -1;  // uniform sample of the launches
    if (evUsed == evPool.size()) {
      hipEvent_t a, b;
      HIP_CHECK(hipEventCreate(&a));
      HIP_CHECK(hipEventCreate(&b));
      evPool.emplace_back(a, b);
      evClass.push_back(kc);
      evIter.push_back(-1);
    }
    evClass[evUsed] = kc;
    evIter[evUsed] = curPcgIter;
    start = evPool[evUsed].first;
    stop = evPool[evUsed].second;
    return static_cast<int>(evUsed++);
  }
  void tDropFrom(size_t firstSlot, int firstDeadIter) {
    for (size_t i = firstSlot; i < evUsed; ++i)
      if (evIter[i] >= firstDeadIter) evClass[i] = -1;
  }
  void tCollect() {
    if (!timing || evUsed == 0) return;
    HIP_CHECK(hipStreamSynchronize(stream));
    for (size_t i = 0; i < evUsed; ++i) {
      if (evClass[i] < 0) continue;
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, evPool[i].first, evPool[i].second));
      kcMs[evClass[i]] += ms;
      kcN[evClass[i]] += 1;
    }
    evUsed = 0;
  }
};

namespace cvd {

static std::vector<int> rangeOf(const cvd_opt_params& p, int F) {
  std::vector<int> r;
  if (!p.frame_range || p.num_range_frames <= 0) {
    for (int i = 0; i < F; ++i) r.push_back(i);
  } else {
    r.assign(p.frame_range, p.frame_range + p.num_range_frames);
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
    for (int f : r)
      if (f < 0 || f >= F) throw std::runtime_error("frame range out of bounds");
  }
  return r;
}

// DepthVideoPoseOptimizer ctor, reference lib/PoseOptimizer.cpp:753-782
static void posesToParams(cvd_handle* h) {
  h->poseParams.resize(h->F);
  for (int f = 0; f < h->F; ++f) {
    const cvd_frame_pose& p = h->poses[f];
    auto& pose = h->poseParams[f];
    pose[0] = p.position[0];
    pose[1] = p.position[1];
    pose[2] = p.position[2];
    const double q[4] = {p.orientation[0], p.orientation[1], p.orientation[2], p.orientation[3]};
    double R[3][3];
    quatToMatrix(q, R);  // columns right, up, -front == q*ex, q*ey, q*ez
    matrixToAngleAxis(R, &pose[3]);
    pose[6] = std::tan(p.vfov / 2.0);
  }
  h->poseParamsValid = true;
}

// pose write-back, reference lib/PoseOptimizer.cpp:964-987
static void paramsToPoses(cvd_handle* h, const cvd_opt_params& params) {
  for (int f : rangeOf(params, h->F)) {
    const auto& pose = h->poseParams[f];
    cvd_frame_pose& p = h->poses[f];
    p.position[0] = static_cast<float>(pose[0]);
    p.position[1] = static_cast<float>(pose[1]);
    p.position[2] = static_cast<float>(pose[2]);
    double R[3][3], q[4];
    angleAxisToMatrix(&pose[3], R);
    matrixToQuat(R, q);
    for (int i = 0; i < 4; ++i) p.orientation[i] = static_cast<float>(q[i]);
    const double fsrc = (params.intr_opt == CVD_INTR_SHARED) ? h->poseParams[0][6] : pose[6];
    p.vfov = static_cast<float>(std::atan(fsrc) * 2.f);
    p.hfov = static_cast<float>(std::atan(fsrc * h->aspect) * 2.f);
  }
}

static void resetXforms(cvd_handle* h, const cvd_xform_desc& d, bool spatial) {
  const int nb = xformNumBlocks(d), bs = xformBlockSize(d);
  if (!spatial) {
    if (d.type != CVD_XFORM_DEPTH) throw std::runtime_error("Transform has the wrong type.");
    if (d.depth_type == CVD_DEPTH_GRID && d.grid_size[2] > 1) {
      // GridDepthXform ctor, reference lib/DepthMapTransform.cpp:709-717
      if (d.depth_min_max[0] <= 0.0 || d.depth_min_max[1] <= 0.0) throw std::runtime_error("Depth values must be positive.");
      if (d.depth_min_max[1] - d.depth_min_max[0] <= 0.0) throw std::runtime_error("Depth range must be positive.");
      if (d.cubic_interpolation)
        throw std::runtime_error("Cubic interpolation of depth-wise grids is not defined (reference lib/DepthMapTransform.cpp:944 "
                                 "never applies the depth-wise weights).");
    }
    if (d.depth_type == CVD_DEPTH_GRID && !d.cubic_interpolation && bs != 1 && (d.grid_size[0] > 1 || d.grid_size[2] > 1))
      throw std::runtime_error(
          "Linear grid gather is only defined for 1-parameter value transforms (reference "
          "lib/DepthMapTransform.cpp:829 aliases the blocks otherwise).");
    h->ddesc = d;
    h->dparams.assign(static_cast<size_t>(h->F) * nb * bs, 1.0);
  } else {
    if (d.type != CVD_XFORM_SPATIAL) throw std::runtime_error("Transform has the wrong type.");
    h->sdesc = d;
    h->sparams.assign(static_cast<size_t>(h->F) * nb * bs, 0.0);
  }
}

// DepthVideoProcessor::gridXformSplit, reference lib/Processor.cpp:888-985
static void gridXformSplit(cvd_handle* h, const cvd_xform_desc& nd) {
  if (nd.depth_type != CVD_DEPTH_GRID) throw std::runtime_error("Transform type must be a grid type.");
  const cvd_xform_desc prev = h->ddesc;
  if (prev.depth_type != CVD_DEPTH_GLOBAL && prev.depth_type != CVD_DEPTH_GRID)
    throw std::runtime_error("Can only split global or grid type transforms.");
  if (nd.value_xform != prev.value_xform)
    throw std::runtime_error("Old and new transforms must use same value transform.");
  if (prev.depth_type != CVD_DEPTH_GLOBAL &&
      (prev.grid_size[0] > nd.grid_size[0] || prev.grid_size[1] > nd.grid_size[1]))
    throw std::runtime_error(
        "New transform must have at least the same number of rows and columns as the old transform.");
  const std::vector<double> old = h->dparams;
  const int oldN = h->nD();
  resetXforms(h, nd, false);
  const int N = xformBlockSize(nd);
  const int newCols = nd.grid_size[0], newRows = nd.grid_size[1];
  const int newN = h->nD();
  for (int f = 0; f < h->F; ++f) {
    const double* po = &old[static_cast<size_t>(f) * oldN];
    double* pn = &h->dparams[static_cast<size_t>(f) * newN];
    for (int row = 0; row < newRows; ++row) {
      for (int col = 0; col < newCols; ++col) {
        double* dst = pn + static_cast<size_t>(col + row * newCols) * N;
        if (prev.depth_type == CVD_DEPTH_GLOBAL) {
          for (int i = 0; i < N; ++i) dst[i] = po[i];
        } else {
          const int prevRows = prev.grid_size[1], prevCols = prev.grid_size[0];
          const double maxx = std::nextafter(static_cast<double>(prevCols - 1), 0.0);
          const double maxy = std::nextafter(static_cast<double>(prevRows - 1), 0.0);
          const double sx = std::min(col / double(newCols - 1) * (prevCols - 1), maxx);
          const double sy = std::min(row / double(newRows - 1) * (prevRows - 1), maxy);
          const int ix = static_cast<int>(sx), iy = static_cast<int>(sy);
          const double rx = sx - ix, ry = sy - iy;
          const double* b0 = po + static_cast<size_t>(ix + iy * prevCols) * N;
          const double* b1 = po + static_cast<size_t>((ix + 1) + iy * prevCols) * N;
          const double* b2 = po + static_cast<size_t>(ix + (iy + 1) * prevCols) * N;
          const double* b3 = po + static_cast<size_t>((ix + 1) + (iy + 1) * prevCols) * N;
          const double w0 = (1.f - rx) * (1.f - ry), w1 = rx * (1.f - ry), w2 = (1.f - rx) * ry, w3 = rx * ry;
          for (int i = 0; i < N; ++i) dst[i] = b0[i] * w0 + b1[i] * w1 + b2[i] * w2 + b3[i] * w3;
        }
      }
    }
  }
}

// ---- problem -> device layout -------------------------------------------------------------------------
static Layout makeLayout(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, ProblemKind kind) {
  if ((p.smooth_static_weight > 0.0 || p.smooth_dynamic_weight > 0.0) && kind == PK_POSE_STEP) {
    if (p.smooth_loss_type < CVD_SMOOTH_EUCLIDEAN_LAPLACIAN || p.smooth_loss_type > CVD_SMOOTH_REPRO_LOG_DEPTH_CONSISTENCY)
      throw std::runtime_error("Invalid loss type.");
    if (!h->haveTriplets) throw std::runtime_error("Missing triplet constraints.");
  }
  Layout L{};
  L.F = h->F;
  L.B = h->Bsz();
  L.depthType = h->ddesc.depth_type;
  L.N = xformBlockSize(h->ddesc);
  L.cubic = h->ddesc.cubic_interpolation ? 1 : 0;
  L.gx = h->ddesc.depth_type == CVD_DEPTH_GRID ? h->ddesc.grid_size[0] : 1;
  L.gy = h->ddesc.depth_type == CVD_DEPTH_GRID ? h->ddesc.grid_size[1] : 1;
  L.maxcx = std::nextafter(static_cast<double>(L.gx - 1), 0.0);
  L.maxcy = std::nextafter(static_cast<double>(L.gy - 1), 0.0);
  L.gz = h->ddesc.depth_type == CVD_DEPTH_GRID ? std::max(1, h->ddesc.grid_size[2]) : 1;
  L.maxcz = std::nextafter(static_cast<double>(L.gz - 1), 0.0);
  L.dispMin = 0.0;
  L.dispInterval = 1.0;
  if (L.gz > 1) {  // reference lib/DepthMapTransform.cpp:719-729
    const double dmin = 1.0 / h->ddesc.depth_min_max[1], dmax = 1.0 / h->ddesc.depth_min_max[0];
    L.dispMin = dmin;
    L.dispInterval = (dmax - dmin) / (L.gz - 1);
  }
  L.nD = h->nD();
  L.spatialType = h->sdesc.spatial_type;
  L.sgx = h->sdesc.grid_size[0];
  L.sgy = h->sdesc.grid_size[1];
  L.smaxcx = std::nextafter(static_cast<double>(L.sgx - 1), 0.0);
  L.smaxcy = std::nextafter(static_cast<double>(L.sgy - 1), 0.0);
  L.nS = h->nS();
  L.aspect = h->aspect;
  L.vFocal = (h->aspect >= 1.f ? p.focal_long / static_cast<double>(h->aspect) : p.focal_long);
  L.intrOpt = p.intr_opt;
  L.lossType = p.static_loss_type;
  L.ws = p.static_spatial_weight;
  L.wd = p.static_depth_weight;
  L.cauchyB = p.robustness * p.robustness;
  L.cauchyC = 1.0 / L.cauchyB;
  L.robustA = p.robustness;
  if (h->opt.robust_loss != 0 && h->opt.robust_loss != 1) throw std::runtime_error("robust_loss must be 0 (Cauchy) or 1 (Huber)");
  L.robustKind = h->opt.robust_loss;
  // scale regulariser sample grid, reference lib/PoseOptimizer.cpp:1347-1351
  int gX = p.scale_reg_grid_size;
  int gY = static_cast<int>(std::round(static_cast<float>(gX) * h->invAspect));
  if (h->aspect <= 1.f) std::swap(gX, gY);
  L.sregX = gX;
  L.sregY = gY;
  if (kind == PK_POSE_STEP) {
    L.includeStatic = 1;
    L.scaleRegSqrt = (!p.fix_depth_xforms && p.scale_reg > 0.0) ? std::sqrt(p.scale_reg) : 0.0;
    L.focalRegSqrt = (p.focal_reg > 0.0 && p.intr_opt != CVD_INTR_FIXED) ? std::sqrt(p.focal_reg) : 0.0;
    L.depthDeformW = depthDeformReg > 0.0 ? depthDeformReg : 0.0;
    L.spatialDeformW = p.spatial_deform_reg > 0.0 ? p.spatial_deform_reg : 0.0;
  } else {
    // normalizeDepthFromFirstFrame (default): no pairs at all (reference lib/PoseOptimizer.cpp:1014-1018); otherwise the
    // pair loop of :1020-1105: one DisparityDissimilarityCost with CauchyLoss(robustness) per constraint
    L.includeStatic = p.normalize_depth_from_first_frame ? 0 : 1;
    if (L.includeStatic) {
      L.lossType = kLossNormalizeDisparity;
      L.robustKind = kRobustCauchy;  // (the reference hard-wires CauchyLoss here, :1080)
    }
    L.scaleRegSqrt = p.scale_reg > 0.0 ? std::sqrt(p.scale_reg) : 0.0;
    L.focalRegSqrt = 0.0;
    L.depthDeformW = p.depth_deform_reg_initial > 0.0 ? p.depth_deform_reg_initial : 0.0;
    L.spatialDeformW = 0.0;
  }
  {
    const std::vector<int> rg = rangeOf(p, h->F);
    L.firstFrame = rg.empty() ? 0 : rg.front();
    L.lastFrame = rg.empty() ? 0 : rg.back();
    L.positionRegSqrt = (kind == PK_POSE_STEP && p.position_reg > 0.0) ? std::sqrt(p.position_reg) : 0.0;
    L.rank = h->rank;
    L.world = h->world;
  }
  if (L.scaleRegSqrt > 0.0 && (L.sregX < 2 || L.sregY < 2))
    throw std::runtime_error("scaleRegGridSize too small for this aspect ratio.");
  // AdaptiveDeformationCost replaces DeformationCost for the depth transforms that have deformation residuals, i.e.
  // grids (reference lib/PoseOptimizer.cpp:1465-1484: the others are skipped before the cost is constructed)
  L.adaptW = nullptr;
  L.adaptive = 0.0;
  if (p.adaptive_deformation_cost > 0.0 && L.depthDeformW > 0.0 && L.depthType == CVD_DEPTH_GRID) {
    if (!h->haveDynMasks) throw std::runtime_error("Adaptive smoothness requires a dynamic mask stream.");
    if (L.gz > 1) throw std::runtime_error("AdaptiveDeformationCost with a depth-wise grid is not implemented on the device path.");
    if (L.gx < 2 || L.gy < 2) throw std::runtime_error("Adaptive deformation cost needs a grid of at least 2 x 2 vertices.");
    if (h->adaptGx != L.gx || h->adaptGy != L.gy) {
      const size_t G = static_cast<size_t>(L.gx) * L.gy;
      h->dAdaptW.ensure(G * h->F);
      hipLaunchKernelGGL(k_adaptive_weights, dim3(h->F), dim3(256), 2 * G * sizeof(double), h->stream, h->dDynMask.p, h->dynW,
                         h->dynH, L.gx, L.gy, h->dAdaptW.p);
      HIP_CHECK(hipGetLastError());
      h->adaptGx = L.gx;
      h->adaptGy = L.gy;
    }
    L.adaptW = h->dAdaptW.p;
    L.adaptive = p.adaptive_deformation_cost;
  }
  return L;
}

// The per-frame kernels of the solve (k_matvec_finish, k_cg_update, the block inverse, the fast pairs product) hold one
// frame block per workgroup with at most 256 unknowns.  Checked BEFORE any work or state mutation (a coarse-to-fine
// schedule would otherwise fail at its last level with the transforms already refined).
constexpr int kMaxFrameBlock = 512;
static void checkFrameBlock(size_t B, const char* what) {
  if (B > static_cast<size_t>(kMaxFrameBlock))
    throw std::runtime_error(fmt("%s: %zu unknowns per frame (7 + depth-transform + spatial-transform parameters) exceed the "
                                 "%d this build supports", what, B, kMaxFrameBlock));
}

static void tapCounts(const Layout& L, int& KD, int& KS) {
  KD = (L.depthType == CVD_DEPTH_GRID) ? (L.cubic ? 16 : 4) : 1;
  // depth-wise grids: 8 taps (spatial x depth-wise) run in the 16-slot instantiation, 2 taps (depth-wise only) in the 4-slot
  if (L.depthType == CVD_DEPTH_GRID && L.gz > 1) KD = L.gx > 1 ? 16 : 4;
  switch (L.spatialType) {
    case CVD_SPATIAL_IDENTITY: KS = 0; break;
    case CVD_SPATIAL_BICUBIC_GRID: KS = 16; break;
    default: KS = 4;
  }
}

// Scope of the specialised fast kernels: identity spatial transform and the three reprojection losses.
static bool fastLoss(const Layout& L) {
  if (L.gz > 1) return false;  // (the fast kernels gather 2-D grids only)
  if (L.B > 256) return false;  // (and hold one element of a frame block per thread)
  return L.lossType == CVD_STATIC_REPRO_DISPARITY || L.lossType == CVD_STATIC_REPRO_DEPTH_RATIO || L.lossType == CVD_STATIC_REPRO_LOG_DEPTH;
}

constexpr long long kListChunk = 768;    // constraints per direction and work item (list mode)
constexpr long long kDenseChunk = 8192;  // pixel slots per direction and work item (dense mode)

static Table makeTable(cvd_handle* h) {
  Table T{};
  T.ndc = h->dense ? nullptr : h->dNdc.p;
  T.dsrc = h->dense ? nullptr : h->dDsrc.p;
  T.pairA = h->dPairA.p;
  T.pairB = h->dPairB.p;
  T.pairOff = h->dPairOff.p;
  T.flow = h->dense ? h->dFlow.p : nullptr;
  T.fmask = h->dense ? h->dFMask.p : nullptr;
  T.depth = h->dDepth.p;
  T.W = h->W;
  T.H = h->H;
  T.sx = 1.f / static_cast<float>(h->W);                 // reference lib/FlowConstraints.cpp:371: Vector2f scale(1.f / w, invAspect / h)
  T.sy = h->invAspect / static_cast<float>(h->H);
  T.invAspect = h->invAspect;
  return T;
}
// Dense mode runs on the specialised fast kernels only (the default residual configuration of the reference pipeline).
static void checkDenseScope(cvd_handle* h, const Layout& L, int KS, bool trip) {
  if (!h->dense) return;
  if (KS != 0 || !fastLoss(L) || L.N != 1 || trip || L.intrOpt == CVD_INTR_SHARED || h->forceGeneric || h->dist() || L.cubic)
    throw std::runtime_error("dense mode (cvd_set_pair_flows) supports the fast kernels' residual configurations only: identity "
                             "spatial transform, a reprojection loss (ReproDisparity / ReproDepthRatio / ReproLogDepth), Scale "
                             "value transform, Global or bilinear grid, per-frame or fixed intrinsics, no smoothness triplets, "
                             "one GPU");
}

#define CVD_DISPATCH(KDv, KSv, ...)                                             \
  do {                                                                          \
    if (KDv == 1 && KSv == 0) { constexpr int KD = 1, KS = 0; __VA_ARGS__; }     \
    else if (KDv == 4 && KSv == 0) { constexpr int KD = 4, KS = 0; __VA_ARGS__; } \
    else if (KDv == 16 && KSv == 0) { constexpr int KD = 16, KS = 0; __VA_ARGS__; } \
    else if (KDv == 1 && KSv == 4) { constexpr int KD = 1, KS = 4; __VA_ARGS__; } \
    else if (KDv == 4 && KSv == 4) { constexpr int KD = 4, KS = 4; __VA_ARGS__; } \
    else if (KDv == 16 && KSv == 4) { constexpr int KD = 16, KS = 4; __VA_ARGS__; } \
    else if (KDv == 1 && KSv == 16) { constexpr int KD = 1, KS = 16; __VA_ARGS__; } \
    else if (KDv == 4 && KSv == 16) { constexpr int KD = 4, KS = 16; __VA_ARGS__; } \
    else { constexpr int KD = 16, KS = 16; __VA_ARGS__; }                        \
  } while (0)

#define CVD_DISPATCH_KD(KDv, ...)                                  \
  do {                                                             \
    if (KDv == 1) { constexpr int KD = 1; __VA_ARGS__; }            \
    else if (KDv == 4) { constexpr int KD = 4; __VA_ARGS__; }       \
    else { constexpr int KD = 16; __VA_ARGS__; }                    \
  } while (0)

constexpr size_t kMaxLds = 160 * 1024;

// Row panels of the packed lower triangle of a B x B frame block that fit `capDoubles` of LDS each (AsmPanels,
// cvd_kernels.h): one panel up to B = 199, two at the reference's default deferred-spatial block B = 201.
static AsmPanels makePanels(int B, size_t capDoubles, int& panelCap) {
  AsmPanels P{};
  P.n = 0;
  P.row[0] = 0;
  size_t biggest = 0;
  int r0 = 0;
  while (r0 < B) {
    if (P.n >= 8) throw std::runtime_error(fmt("frame block of %d unknowns is too large for the assembly kernels", B));
    const size_t base = static_cast<size_t>(r0) * (r0 + 1) / 2;
    int r1 = r0;
    while (r1 < B && static_cast<size_t>(r1 + 1) * (r1 + 2) / 2 - base <= capDoubles) ++r1;
    if (r1 < std::max(r0 + 1, 7) && r1 < B) throw std::runtime_error("LDS panel too small for the assembly kernels");
    biggest = std::max(biggest, static_cast<size_t>(r1) * (r1 + 1) / 2 - base);
    P.row[++P.n] = r1;
    r0 = r1;
  }
  panelCap = static_cast<int>(biggest);
  return P;
}

template <typename K>
static void allowLds(K kernel, size_t bytes) {
  if (bytes > kMaxLds)
    throw std::runtime_error(fmt("per-frame block needs %zu B of LDS (> 160 KiB): frame block too large", bytes));
  if (bytes > 48 * 1024) {
    // one driver call per (device, kernel) and high-water mark, not per launch; handles on several GPUs / host threads
    // share this cache
    static std::map<std::pair<int, const void*>, size_t> granted;
    static std::mutex grantedMutex;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(grantedMutex);
    size_t& g = granted[{dev, reinterpret_cast<const void*>(kernel)}];
    if (bytes > g) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(bytes)));
      g = bytes;
    }
  }
}

// ---- coarse level: symbolic block-sparse Cholesky of the frame graph (cvd_coarse.h) --------------------------
// Greedy minimum-degree ordering on the frame graph, column structures with fill, left-looking update lists and
// the level schedule (level of a column = 1 + the highest level among the columns that update it).
static void buildCoarsePlan(cvd_handle* h, const std::vector<std::pair<int, int>>& edgeList,
                            const std::vector<int>& itemEdge) {
  auto& C = h->coarse;
  const int F = h->F;
  hipStream_t s = h->stream;
  std::vector<std::set<int>> adj(F);
  for (const auto& e : edgeList) {
    adj[e.first].insert(e.second);
    adj[e.second].insert(e.first);
  }
  // Multilevel independent-set ordering: every round eliminates a maximal independent set of low-degree frames
  // (degree <= 2 * current minimum + 2), which become one level of the factorisation; the elimination graph
  // receives the fill.  Far fewer levels than plain minimum degree on these near-chain graphs (34 vs 73 for the
  // 300-frame hierarchical pair set) at ~15% more fill.
  std::vector<int> order, pos(F, -1);
  std::vector<std::vector<int>> structFrames(F);  // by position: neighbours (frames) alive at elimination
  {
    std::vector<std::set<int>> g = adj;
    std::vector<char> alive(F, 1);
    int remaining = F;
    while (remaining > 0) {
      size_t minDeg = std::numeric_limits<size_t>::max();
      for (int v = 0; v < F; ++v)
        if (alive[v]) minDeg = std::min(minDeg, g[v].size());
      const size_t cap = 2 * minDeg + 2;
      std::vector<int> cand;
      for (int v = 0; v < F; ++v)
        if (alive[v] && g[v].size() <= cap) cand.push_back(v);
      std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return g[a].size() < g[b].size(); });
      std::vector<char> blocked(F, 0);
      std::vector<int> chosen;
      for (int v : cand) {
        if (blocked[v]) continue;
        chosen.push_back(v);
        blocked[v] = 1;
        for (int u : g[v]) blocked[u] = 1;
      }
      for (int v : chosen) {
        pos[v] = static_cast<int>(order.size());
        order.push_back(v);
        std::vector<int> nb(g[v].begin(), g[v].end());
        structFrames[pos[v]] = nb;
        for (int a2 : nb) {
          g[a2].erase(v);
          for (int b2 : nb)
            if (a2 != b2) g[a2].insert(b2);
        }
        alive[v] = 0;
        g[v].clear();
        --remaining;
      }
    }
  }
  // column structures by position (sorted), block ids
  std::vector<int> colPtr(F + 1, 0), blkCol, blkRow;
  std::vector<std::vector<int>> colRows(F);
  for (int j = 0; j < F; ++j) {
    for (int fr : structFrames[j]) colRows[j].push_back(pos[fr]);
    std::sort(colRows[j].begin(), colRows[j].end());
    colPtr[j + 1] = colPtr[j] + static_cast<int>(colRows[j].size());
  }
  const int nnz = colPtr[F];
  const int nBlocks = F + nnz;
  blkCol.assign(nBlocks, 0);
  blkRow.assign(nBlocks, 0);
  std::map<std::pair<int, int>, int> blockOf;  // (row position, column position) -> block id
  for (int j = 0; j < F; ++j) {
    blkCol[j] = j;
    blkRow[j] = j;
    blockOf[{j, j}] = j;
    for (size_t e = 0; e < colRows[j].size(); ++e) {
      const int b = F + colPtr[j] + static_cast<int>(e);
      blkCol[b] = j;
      blkRow[b] = colRows[j][e];
      blockOf[{colRows[j][e], j}] = b;
    }
  }
  // row structures and levels
  std::vector<std::vector<int>> rowBlks(F);
  for (int b = F; b < nBlocks; ++b) rowBlks[blkRow[b]].push_back(b);
  std::vector<int> level(F, 0);
  for (int j = 0; j < F; ++j)
    for (int b : rowBlks[j]) level[j] = std::max(level[j], level[blkCol[b]] + 1);
  const int nLevels = F ? *std::max_element(level.begin(), level.end()) + 1 : 0;
  std::vector<int> rowPtr(F + 1, 0), rowBlk;
  for (int j = 0; j < F; ++j) {
    std::sort(rowBlks[j].begin(), rowBlks[j].end(), [&](int a, int b) { return blkCol[a] < blkCol[b]; });
    rowPtr[j + 1] = rowPtr[j] + static_cast<int>(rowBlks[j].size());
    rowBlk.insert(rowBlk.end(), rowBlks[j].begin(), rowBlks[j].end());
  }
  // update lists: column k contributes L(i,k) L(j,k)^T to block (i, j) for every i >= j in struct(k)
  std::vector<std::vector<std::pair<int, int>>> upd(nBlocks);
  for (int k = 0; k < F; ++k) {
    const auto& rows = colRows[k];
    for (size_t a = 0; a < rows.size(); ++a)
      for (size_t b = a; b < rows.size(); ++b) {
        const int j = rows[a], i = rows[b];  // i >= j
        const int target = blockOf.at({i, j});
        upd[target].push_back({F + colPtr[k] + static_cast<int>(b), F + colPtr[k] + static_cast<int>(a)});
      }
  }
  std::vector<int> updPtr(nBlocks + 1, 0), updA, updB, updBlk;
  for (int b = 0; b < nBlocks; ++b) {
    updPtr[b + 1] = updPtr[b] + static_cast<int>(upd[b].size());
    for (const auto& u : upd[b]) { updA.push_back(u.first); updB.push_back(u.second); updBlk.push_back(b); }
  }
  std::vector<int> levelPtr(nLevels + 1, 0), levelCols, lvlBlkPtr(nLevels + 1, 0), lvlBlks;
  for (int lv = 0; lv < nLevels; ++lv) {
    for (int j = 0; j < F; ++j)
      if (level[j] == lv) {
        levelCols.push_back(j);
        lvlBlks.push_back(j);
        for (int e = colPtr[j]; e < colPtr[j + 1]; ++e) lvlBlks.push_back(F + e);
      }
    levelPtr[lv + 1] = static_cast<int>(levelCols.size());
    lvlBlkPtr[lv + 1] = static_cast<int>(lvlBlks.size());
  }
  // W = L^-1: column j is non-zero on the elimination-tree path j -> root (parent = first row below the diagonal)
  std::vector<int> wPtr(F + 1, 0), wRow;
  for (int j = 0; j < F; ++j) {
    for (int i = j; i >= 0; i = colRows[i].empty() ? -1 : colRows[i][0]) wRow.push_back(i);
    wPtr[j + 1] = static_cast<int>(wRow.size());
  }
  const int nW = static_cast<int>(wRow.size());
  std::vector<std::vector<std::pair<int, int>>> wt(F);  // row -> (column, W block id), columns ascending
  for (int j = 0; j < F; ++j)
    for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) wt[wRow[t]].push_back({j, t});
  std::vector<int> wtPtr(F + 1, 0), wtBlk, wtCol, wtFrame;
  std::vector<int> wSlot(nW, 0);  // W block -> its slot in the row lists (coarseColumnProducts writes there)
  for (int i = 0; i < F; ++i) {
    wtPtr[i + 1] = wtPtr[i] + static_cast<int>(wt[i].size());
    for (const auto& e : wt[i]) {
      wSlot[e.second] = static_cast<int>(wtBlk.size());
      wtCol.push_back(e.first); wtBlk.push_back(e.second); wtFrame.push_back(order[e.first]);
    }
  }
  std::vector<int> wuPtr(nW + 1, 0), wuL, wuW;
  {
    std::vector<int> mark(F, -1);
    for (int j = 0; j < F; ++j) {
      for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) mark[wRow[t]] = t;
      for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) {
        const int i = wRow[t];
        if (t > wPtr[j])
          for (int b : rowBlks[i]) {
            const int wk = mark[blkCol[b]];
            if (wk >= 0 && wk < t) { wuL.push_back(b); wuW.push_back(wk); }
          }
        wuPtr[t + 1] = static_cast<int>(wuL.size());
      }
      for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) mark[wRow[t]] = -1;
    }
  }
  std::vector<int> edgeBlk, edgeFa, edgeFb;
  for (const auto& e : edgeList) {
    const int pa = pos[e.first], pb = pos[e.second];
    // stored rows = fa, columns = fb; block (i, j), i > j, has rows = frame of position i
    const int b = blockOf.at({std::max(pa, pb), std::min(pa, pb)});
    edgeBlk.push_back((b << 1) | (pa > pb ? 0 : 1));
    edgeFa.push_back(e.first);
    edgeFb.push_back(e.second);
  }
  C.nEdges = static_cast<int>(edgeList.size());
  C.nBlocks = nBlocks;
  C.nLevels = nLevels;
  C.itemEdge = itemEdge;
  auto up = [&](DevBuf<int>& d, const std::vector<int>& v) { d.upload(v.data(), v.size(), s); };
  if (std::getenv("CVD_COARSE_PLAN_STATS")) {  // development aid: shape of the elimination levels
    for (int lv = 0; lv < nLevels; ++lv) {
      long long nupd = 0, maxCol = 0, nblk = 0, maxChain = 0;
      for (int q = levelPtr[lv]; q < levelPtr[lv + 1]; ++q) {
        const int j = levelCols[q];
        long long colUpd = 0, chain = 0;
        const int nOff = colPtr[j + 1] - colPtr[j];
        for (int k = 0; k <= nOff; ++k) {
          const int b = (k == 0) ? j : F + colPtr[j] + k - 1;
          const long long u = updPtr[b + 1] - updPtr[b];
          colUpd += u;
          chain += u ? (u + 15) / 16 + 1 : 0;
        }
        nblk += nOff + 1;
        nupd += colUpd;
        maxCol = std::max(maxCol, colUpd);
        maxChain = std::max(maxChain, chain);
      }
      std::printf("level %2d: cols %3d blocks %5lld updates %6lld  max updates/col %5lld  max serial steps/col %4lld\n", lv,
                  levelPtr[lv + 1] - levelPtr[lv], nblk, nupd, maxCol, maxChain);
    }
  }
  up(C.order, order); up(C.pos, pos); up(C.levelPtr, levelPtr); up(C.levelCols, levelCols);
  up(C.lvlBlkPtr, lvlBlkPtr); up(C.lvlBlks, lvlBlks); up(C.blkCol, blkCol); up(C.blkRow, blkRow);
  up(C.colPtr, colPtr); up(C.rowPtr, rowPtr); up(C.rowBlk, rowBlk); up(C.updPtr, updPtr); up(C.updA, updA);
  up(C.updB, updB); up(C.updBlk, updBlk); up(C.edgeBlk, edgeBlk); up(C.edgeFa, edgeFa); up(C.edgeFb, edgeFb);
  up(C.wPtr, wPtr); up(C.wRow, wRow); up(C.wtPtr, wtPtr); up(C.wtBlk, wtBlk); up(C.wtCol, wtCol); up(C.wtFrame, wtFrame);
  up(C.wuPtr, wuPtr); up(C.wuL, wuL); up(C.wuW, wuW); up(C.wSlot, wSlot);
  C.nW = nW;
  up(C.itemEdgeDev, itemEdge);
  const size_t n = static_cast<size_t>(F) * kCB;
  C.edges.ensure(static_cast<size_t>(std::max(C.nEdges, 1)) * kCBB);
  C.diag.ensure(static_cast<size_t>(h->framesPadded()) * kCBB);
  C.dropDiag.ensure(static_cast<size_t>(F) * kCBB);
  C.Lb.ensure(static_cast<size_t>(nBlocks) * kCBB);
  C.Linv.ensure(static_cast<size_t>(F) * kCBB);
  C.Wb.ensure(static_cast<size_t>(nW) * kCBB);
  C.Wb2.ensure(static_cast<size_t>(nW) * kCBB);
  C.fail2.ensure(1);
  C.rc.ensure(n);
  C.qc.ensure(n);
  C.wq.ensure(static_cast<size_t>(nW) * kCB);
  C.fdotY.ensure(F);
  C.y.ensure(n);
  C.c.ensure(n);
  C.dotPart.ensure(static_cast<size_t>(F));
  C.modeActive.ensure(n);
  C.fail.ensure(1);
  HIP_CHECK(hipStreamSynchronize(s));
  C.plan = CoarsePlan{F, nBlocks, nLevels, C.nEdges, C.order.p, C.pos.p, C.levelPtr.p, C.levelCols.p, C.lvlBlkPtr.p,
                      C.lvlBlks.p, C.blkCol.p, C.blkRow.p, C.colPtr.p, C.rowPtr.p, C.rowBlk.p, C.updPtr.p, C.updA.p,
                      C.updB.p, C.edgeBlk.p, C.edgeFa.p, C.edgeFb.p, C.wPtr.p, C.wRow.p, C.wtPtr.p, C.wtBlk.p, C.wtCol.p, C.wtFrame.p,
                      C.wuPtr.p, C.wuL.p, C.wuW.p, nW, C.updBlk.p};
  C.valid = true;
  C.denseReady = false;
}

// ---- coarse graph sparsification ---------------------------------------------------------------------------
// The coarse factorisation is sparse-direct on the frame graph: its cost follows the FILL of that graph.  The reference
// sampler's hierarchical list (utils/frame_sampling.py:77-120: distance 2^l from every 2^(l-1)-th frame) eliminates with
// ~10 k block updates at 300 frames; the densified "~4k pairs" list of BASELINE.json (long-range pairs from nearly every
// frame) needs ~10^6 and a 44 ms factorisation per rebuild.  The coarse level is only a preconditioner, so it may be built
// on a SUBGRAPH: when the full graph's elimination exceeds a budget, a pair {a, b} at distance d stays in the coarse graph
// iff both frames are multiples of s(d) = the largest power of two <= d / 8 (>= 1) -- a nested, multi-scale subgraph,
// sparse for any flow list.  Dropped pairs are removed from the coarse operator altogether (k_coarse_edges: dropDiag),
// which keeps it the Galerkin operator of a sub-problem: SPD and consistent on the smooth drift modes.  Measured on the
// 4140-pair list (ms per LM iteration at the final level / PCG iterations per LM iteration): full graph 30.0 / 36
// (44 ms per factorisation), s(d) <= d/2 (the reference sampler's own 883-edge skeleton) 16.9 / 123, d/4 12.9 / 88,
// d/8 10.8 / 64, d/16 11.7 / 47.
static long long coarseEliminationUpdates(int F, const std::vector<std::pair<int, int>>& edgeList) {
  std::vector<std::set<int>> g(F);
  for (const auto& e : edgeList) { g[e.first].insert(e.second); g[e.second].insert(e.first); }
  std::set<std::pair<int, int>> queue;
  for (int v = 0; v < F; ++v) queue.insert({static_cast<int>(g[v].size()), v});
  long long updates = 0;
  while (!queue.empty()) {
    const int v = queue.begin()->second;
    queue.erase(queue.begin());
    const std::vector<int> nb(g[v].begin(), g[v].end());
    const long long sN = static_cast<long long>(nb.size());
    updates += sN * (sN + 1) / 2;
    if (updates > (1ll << 40)) break;
    for (int a : nb) queue.erase({static_cast<int>(g[a].size()), a});
    for (int a : nb) {
      g[a].erase(v);
      for (int b : nb)
        if (a != b) g[a].insert(b);
    }
    for (int a : nb) queue.insert({static_cast<int>(g[a].size()), a});
    g[v].clear();
  }
  return updates;
}
static bool coarseKeepsPair(int a, int b) {
  static const int shift = []() { const char* e = std::getenv("CVD_COARSE_KEEP_SHIFT"); return e ? std::atoi(e) : 3; }();  // development knob
  const int d = std::abs(a - b);
  int s2 = 1;
  while (s2 * 2 <= (d >> shift)) s2 *= 2;
  return (a % s2) == 0 && (b % s2) == 0;
}

// ---- compile the constraint table + work decomposition for a frame range -------------------------------
static void compileTable(cvd_handle* h, const std::vector<int>& range, bool withTriplets = false, bool ignoreStatic = false) {
  std::vector<unsigned char> inRange(h->F, 0);
  for (int f : range) inRange[f] = 1;
  if (h->tableValid && inRange == h->tableRange && withTriplets == h->tableWithTriplets && ignoreStatic == h->tableIgnoresStatic) return;
  h->tableIgnoresStatic = ignoreStatic;
  hipStream_t s = h->stream;
  h->dInRange.upload(inRange.data(), inRange.size(), s);
  {
    std::vector<unsigned char> owner(h->F, 0);
    for (int f = 0; f < h->F; ++f) owner[f] = inRange[f] && (f % h->world == h->rank);
    h->dRegOwner.upload(owner.data(), owner.size(), s);
  }
  h->dCount.ensure(1);
  HIP_CHECK(hipMemsetAsync(h->dCount.p, 0, sizeof(unsigned long long), s));
  if (h->dense) {
    // no table: the kernels read the images; only the number of valid constraints is needed here
    if (h->C > 0) {
      const unsigned grid = static_cast<unsigned>((h->C + 255) / 256);
      hipLaunchKernelGGL(k_dense_count, dim3(grid), dim3(256), 0, s, makeTable(h), h->P, h->dInRange.p, h->dCount.p);
      HIP_CHECK(hipGetLastError());
    }
  } else {
    h->dNdc.ensure(std::max<long long>(h->C, 1));
    h->dDsrc.ensure(std::max<long long>(h->C, 1));
  }
  if (h->C > 0 && !h->dense) {
    const int bs = 256;
    const unsigned grid = static_cast<unsigned>((h->C + bs - 1) / bs);
    hipLaunchKernelGGL(k_build_table, dim3(grid), dim3(bs), 0, s, h->W, h->H, h->invAspect, h->C, h->dLoc.p,
                       h->dStatic.p, h->dCPair.p, h->dPairA.p, h->dPairB.p, h->dInRange.p, h->dDepth.p,
                       h->dNdc.p, h->dDsrc.p, h->dCount.p, ignoreStatic ? 1 : 0);
    HIP_CHECK(hipGetLastError());
  }
  if (h->dist()) NCCL_CHECK(ncclAllReduce(h->dCount.p, h->dCount.p, 1, ncclUint64, ncclSum, h->comm, s));
  unsigned long long nv = 0;
  HIP_CHECK(hipMemcpyAsync(&nv, h->dCount.p, sizeof(nv), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  h->numValid = static_cast<long long>(nv);

  // work items: one per UNDIRECTED pair {a < b} and chunk; each carries a slice of a->b and of b->a
  h->itemFa.clear();
  h->itemFb.clear();
  h->itemRange.clear();
  std::vector<std::vector<int>> frameItems(h->F), framePairs(h->F);
  std::map<std::pair<int, int>, std::array<int, 2>> edges;  // (min, max) -> {pair min->max, pair max->min}
  for (int p = 0; p < h->P; ++p) {
    const int a = h->pairA[p], b = h->pairB[p];
    if (!inRange[a] || !inRange[b] || a == b) continue;
    const long long n = h->pairOff[p + 1] - h->pairOff[p];
    if (n <= 0) continue;
    framePairs[a].push_back(p * 2 + 0);
    framePairs[b].push_back(p * 2 + 1);
    auto it = edges.find({std::min(a, b), std::max(a, b)});
    if (it == edges.end()) it = edges.insert({{std::min(a, b), std::max(a, b)}, {-1, -1}}).first;
    it->second[a < b ? 0 : 1] = p;
  }
  struct ItemDesc { int fa, fb; long long b0, e0, b1, e1; };
  std::vector<ItemDesc> itemList;
  for (const auto& e : edges) {
    const int fa = e.first.first, fb = e.first.second;
    long long n0 = 0, n1 = 0, o0 = 0, o1 = 0;
    if (e.second[0] >= 0) { o0 = h->pairOff[e.second[0]]; n0 = h->pairOff[e.second[0] + 1] - o0; }
    if (e.second[1] >= 0) { o1 = h->pairOff[e.second[1]]; n1 = h->pairOff[e.second[1] + 1] - o1; }
    const long long chunk = h->dense ? kDenseChunk : kListChunk;
    const long long nItems = std::max<long long>(1, (std::max(n0, n1) + chunk - 1) / chunk);
    const long long c0 = (n0 + nItems - 1) / nItems, c1 = (n1 + nItems - 1) / nItems;
    for (long long k = 0; k < nItems; ++k) {
      const long long b0 = o0 + std::min(n0, k * c0), e0 = o0 + std::min(n0, (k + 1) * c0);
      const long long b1 = o1 + std::min(n1, k * c1), e1 = o1 + std::min(n1, (k + 1) * c1);
      if (b0 >= e0 && b1 >= e1) continue;
      itemList.push_back({fa, fb, b0, e0, b1, e1});
    }
  }
  // Longest items first: the pair-major kernels run one workgroup per item in launch order, ~2.7 rounds of the device at the
  // benchmark's 2070 items -- with the short items last the final, partly filled round is short too.  (Stable: equal sizes
  // keep the frame-pair order.)
  static const bool itemOrderOff = std::getenv("CVD_ITEMS_UNSORTED") != nullptr;  // comparison knob
  if (!itemOrderOff)
    std::stable_sort(itemList.begin(), itemList.end(), [](const ItemDesc& a, const ItemDesc& b) {
      return (a.e0 - a.b0) + (a.e1 - a.b1) > (b.e0 - b.b0) + (b.e1 - b.b1);
    });
  for (const ItemDesc& d : itemList) {
    const int item = static_cast<int>(h->itemFa.size());
    h->itemFa.push_back(d.fa);
    h->itemFb.push_back(d.fb);
    h->itemRange.insert(h->itemRange.end(), {d.b0, d.e0, d.b1, d.e1});
    frameItems[d.fa].push_back(item * 2 + 0);
    frameItems[d.fb].push_back(item * 2 + 1);
  }
  // ---- explicit-block mode of the dense mode (cvd_cross.h): one entry per undirected pair with both directions' whole
  // pixel ranges, two partial rows each, rows grouped by frame
  h->xFa.clear();
  h->xFb.clear();
  if (h->dense) {
    std::vector<long long> xRange;
    std::vector<std::vector<int>> frameRows(h->F);
    for (const auto& e : edges) {
      const int fa = e.first.first, fb = e.first.second;
      long long b0 = 0, e0 = 0, b1 = 0, e1 = 0;
      if (e.second[0] >= 0) { b0 = h->pairOff[e.second[0]]; e0 = h->pairOff[e.second[0] + 1]; }
      if (e.second[1] >= 0) { b1 = h->pairOff[e.second[1]]; e1 = h->pairOff[e.second[1] + 1]; }
      if (b0 >= e0 && b1 >= e1) continue;
      const int k = static_cast<int>(h->xFa.size());
      h->xFa.push_back(fa);
      h->xFb.push_back(fb);
      xRange.insert(xRange.end(), {b0, e0, b1, e1});
      frameRows[fa].push_back(k * 2 + 0);
      frameRows[fb].push_back(k * 2 + 1);
    }
    std::vector<int> xFiOff(h->F + 1, 0), xSlot(std::max<size_t>(1, h->xFa.size() * 2), 0);
    int row = 0;
    for (int f = 0; f < h->F; ++f) {
      for (int code : frameRows[f]) xSlot[code] = row++;
      xFiOff[f + 1] = row;
    }
    h->dXFa.upload(h->xFa.data(), h->xFa.size(), s);
    h->dXFb.upload(h->xFb.data(), h->xFb.size(), s);
    h->dXRange.upload(xRange.data(), xRange.size(), s);
    h->dXSlot.upload(xSlot.data(), xSlot.size(), s);
    h->dXFiOff.upload(xFiOff.data(), xFiOff.size(), s);
  }
  // ---- scene-flow smoothness triplets: table, active groups (all three frames in range; groups are sharded
  // over the ranks like the per-frame regularisers), per-frame (group, role) lists and their partial-product rows
  h->tripActive.clear();
  h->numValidTrip = 0;
  std::vector<std::vector<int>> frameTrips(h->F);
  if (withTriplets) {
    std::map<int, int> groupOf;
    for (size_t g = 0; g < h->tripCenter.size(); ++g) groupOf[h->tripCenter[g]] = static_cast<int>(g);
    if (!range.empty()) {
      // reference lib/PoseOptimizer.cpp:1255-1262: every in-range consecutive triple needs its constraints
      for (int fr = range.front(); fr < range.back() - 1; ++fr) {
        if (!inRange[fr] || !inRange[fr + 1] || !inRange[fr + 2]) continue;
        auto itg = groupOf.find(fr + 1);
        if (itg == groupOf.end()) throw std::runtime_error("Missing triplet constraints.");
        const int g = itg->second;
        h->tripActive.push_back(g);
      }
    }
    h->dTNdc.ensure(static_cast<size_t>(std::max<long long>(h->tripC, 1)) * 3);
    h->dTDsrc.ensure(static_cast<size_t>(std::max<long long>(h->tripC, 1)) * 3);
    HIP_CHECK(hipMemsetAsync(h->dCount.p, 0, sizeof(unsigned long long), s));
    if (h->tripC > 0) {
      const unsigned grid = static_cast<unsigned>((h->tripC + 255) / 256);
      hipLaunchKernelGGL(k_build_triplet_table, dim3(grid), dim3(256), 0, s, h->W, h->H, h->invAspect, h->tripC,
                         h->dTLoc.p, h->dTGroupOfC.p, h->dTCenterAll.p, h->F, h->dInRange.p, h->dDepth.p, h->dTNdc.p,
                         h->dTDsrc.p, h->dCount.p);
      HIP_CHECK(hipGetLastError());
    }
    unsigned long long nvt = 0;
    HIP_CHECK(hipMemcpyAsync(&nvt, h->dCount.p, sizeof(nvt), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    h->numValidTrip = static_cast<long long>(nvt);
    // this rank's share of the groups
    std::vector<int> mine;
    for (size_t k = 0; k < h->tripActive.size(); ++k)
      if (static_cast<int>(k) % h->world == h->rank) mine.push_back(h->tripActive[k]);
    h->tripActive.swap(mine);
    const int pairCodes = static_cast<int>(h->itemFa.size()) * 2;
    for (size_t k = 0; k < h->tripActive.size(); ++k) {
      const int f1 = h->tripCenter[h->tripActive[k]];
      for (int role = 0; role < 3; ++role) {
        frameItems[f1 - 1 + role].push_back(pairCodes + static_cast<int>(k) * 3 + role);
        frameTrips[f1 - 1 + role].push_back((static_cast<int>(k) << 2) | role);
      }
    }
  }
  h->tableWithTriplets = withTriplets;
  h->coarse.valid = false;
  // The coarse level needs the frame graph of the whole problem.  One rank: the local items are the whole problem.
  // Several ranks: only with cvd_set_pair_graph (identical on all ranks); otherwise the level stays off.
  if (static_cast<size_t>(h->F) * kCB <= kCoarseMaxUnknowns &&
      (!h->dist() ? !h->itemFa.empty() : h->haveGlobalEdges)) {  // (rank-independent decision when sharded)
    std::map<std::pair<int, int>, int> edgeId;
    std::vector<std::pair<int, int>> edgeList;
    if (h->haveGlobalEdges) {
      for (const auto& e : h->globalEdges) {
        if (!inRange[e.first] || !inRange[e.second]) continue;
        edgeId.insert({e, static_cast<int>(edgeList.size())});
        edgeList.push_back(e);
      }
    }
    std::vector<int> itemEdge(h->itemFa.size());
    for (size_t i = 0; i < h->itemFa.size(); ++i) {
      const std::pair<int, int> key{h->itemFa[i], h->itemFb[i]};
      auto it = edgeId.find(key);
      if (it == edgeId.end()) {
        if (h->haveGlobalEdges) throw std::runtime_error("cvd_set_pair_graph: a frame pair with constraints is missing from the graph");
        it = edgeId.insert({key, static_cast<int>(edgeList.size())}).first;
        edgeList.push_back(key);
      }
      itemEdge[i] = it->second;
    }
    // sparsify the coarse graph when its elimination is too expensive (a function of the whole problem's pair graph
    // only: identical on all ranks of a sharded run)
    // (read per compile, not cached: tests force the dense / sparsified variants on small problems through it)
    const long long updateBudget = []() { const char* e = std::getenv("CVD_COARSE_UPDATE_BUDGET"); return e ? std::atoll(e) : 40000ll; }();
    h->coarse.sparsified = false;
    h->coarse.denseMode = false;
    const int denseMaxUnknowns = []() { const char* e = std::getenv("CVD_COARSE_DENSE_MAX"); return e ? std::atoi(e) : 4096; }();
    const bool overBudget = coarseEliminationUpdates(h->F, edgeList) > updateBudget;
    if (overBudget && h->F * kCB <= denseMaxUnknowns && !h->dist()) {
      h->coarse.denseMode = true;  // small enough to invert as a dense matrix: keeps every pair (cvd_coarse.h)
    } else if (overBudget) {
      std::vector<int> newId(edgeList.size(), -1);
      std::vector<std::pair<int, int>> kept;
      for (size_t e = 0; e < edgeList.size(); ++e)
        if (coarseKeepsPair(edgeList[e].first, edgeList[e].second)) {
          newId[e] = static_cast<int>(kept.size());
          kept.push_back(edgeList[e]);
        }
      for (auto& ie : itemEdge) ie = newId[ie];
      h->coarse.sparsified = kept.size() != edgeList.size();
      edgeList.swap(kept);
    }
    buildCoarsePlan(h, edgeList, itemEdge);
    if (h->dense) {  // edge block of every cross pair (cvd_cross.h: k_coarse_edges_cross)
      std::map<std::pair<int, int>, int> edgeOfPair;
      for (size_t i = 0; i < h->itemFa.size(); ++i) edgeOfPair[{h->itemFa[i], h->itemFb[i]}] = itemEdge[i];
      std::vector<int> pairEdge(std::max<size_t>(1, h->xFa.size()), -1);
      for (size_t k = 0; k < h->xFa.size(); ++k) {
        auto it = edgeOfPair.find({h->xFa[k], h->xFb[k]});
        if (it != edgeOfPair.end()) pairEdge[k] = it->second;
      }
      h->dXPairEdge.upload(pairEdge.data(), pairEdge.size(), s);
    }
  }
  std::vector<int> fiOff(h->F + 1, 0), fiList, fpOff(h->F + 1, 0), fpList;
  for (int f = 0; f < h->F; ++f) {
    fiOff[f + 1] = fiOff[f] + static_cast<int>(frameItems[f].size());
    fiList.insert(fiList.end(), frameItems[f].begin(), frameItems[f].end());
    fpOff[f + 1] = fpOff[f] + static_cast<int>(framePairs[f].size());
    fpList.insert(fpList.end(), framePairs[f].begin(), framePairs[f].end());
  }
  h->dItemFa.upload(h->itemFa.data(), h->itemFa.size(), s);
  h->dItemFb.upload(h->itemFb.data(), h->itemFb.size(), s);
  h->dItemRange.upload(h->itemRange.data(), h->itemRange.size(), s);
  std::vector<int> itemSlot(fiList.size(), 0);  // [item * 2 + side | pairs * 2 + group * 3 + role] -> row
  for (size_t e = 0; e < fiList.size(); ++e) itemSlot[fiList[e]] = static_cast<int>(e);
  h->qRows = static_cast<int>(fiList.size());
  h->dItemSlot.upload(itemSlot.data(), itemSlot.size(), s);
  if (withTriplets) {
    // compact per-rank group arrays in tripActive order: offsets / centre / rows; per-frame (group, role) lists
    const size_t nG = h->tripActive.size();
    std::vector<long long> tOff(2 * std::max<size_t>(nG, 1), 0);
    std::vector<int> tCen(std::max<size_t>(nG, 1), 0), tSlot(3 * std::max<size_t>(nG, 1), 0);
    const int pairCodes = static_cast<int>(h->itemFa.size()) * 2;
    for (size_t k = 0; k < nG; ++k) {
      const int g = h->tripActive[k];
      tOff[2 * k] = h->tripOff[g];
      tOff[2 * k + 1] = h->tripOff[g + 1];
      tCen[k] = h->tripCenter[g];
      for (int role = 0; role < 3; ++role) tSlot[3 * k + role] = itemSlot[pairCodes + static_cast<int>(k) * 3 + role];
    }
    std::vector<int> ftOff(h->F + 1, 0), ftList;
    for (int f = 0; f < h->F; ++f) {
      ftOff[f + 1] = ftOff[f] + static_cast<int>(frameTrips[f].size());
      ftList.insert(ftList.end(), frameTrips[f].begin(), frameTrips[f].end());
    }
    h->dTOff.upload(tOff.data(), tOff.size(), s);
    h->dTCenter.upload(tCen.data(), tCen.size(), s);
    h->dTSlot.upload(tSlot.data(), tSlot.size(), s);
    h->dFtOff.upload(ftOff.data(), ftOff.size(), s);
    h->dFtList.upload(ftList.data(), ftList.size(), s);
    h->dCostTrip.ensure(std::max<size_t>(nG, 1));
  }
  {
    // k_assemble_fast work list: units of <= kAsmUnit constraints, parts of <= capU units.  capU = the mean units
    // per frame (or per CU when there are fewer frames than CUs), so a frame of average size stays whole and
    // only the long-range hub frames of the hierarchical flow list are split.
    std::vector<int2> units;
    std::vector<int> fuOff(h->F + 1, 0);
    for (int f = 0; f < h->F; ++f) {
      for (const int code : framePairs[f]) {
        const long long n = h->pairOff[(code >> 1) + 1] - h->pairOff[code >> 1];
        for (long long o = 0; o < n; o += (h->dense ? kAsmUnitDense : kAsmUnit)) units.push_back(make_int2(code, static_cast<int>(o)));
      }
      fuOff[f + 1] = static_cast<int>(units.size());
    }
    int activeFrames = 0;
    for (int f = 0; f < h->F; ++f) activeFrames += inRange[f] ? 1 : 0;
    static const double partsPerCU = []() { const char* e = std::getenv("CVD_ASM_PARTS_PER_CU"); return e ? std::atof(e) : 1.0; }();
    const long long denom = std::max<long long>(1, std::max<long long>(activeFrames, static_cast<long long>(partsPerCU * h->numCU)));
    const int capU = static_cast<int>(std::max<long long>(kAsmThreads / 64, (static_cast<long long>(units.size()) + denom - 1) / denom));
    std::vector<AsmPart> parts;
    int slots = 0;
    for (int f = 0; f < h->F; ++f) {
      // (frames outside the range have no entries: their part writes the zero block / gradient / cost)
      const int nu = fuOff[f + 1] - fuOff[f];
      const int np = std::max(1, (nu + capU - 1) / capU);
      const int per = (nu + np - 1) / np;
      for (int q = 0; q < np; ++q) {
        AsmPart a;
        a.frame = f;
        a.u0 = fuOff[f] + std::min(nu, q * per);
        a.u1 = fuOff[f] + std::min(nu, (q + 1) * per);
        a.part = q;
        a.nParts = np;
        a.slot0 = np > 1 ? slots : 0;
        parts.push_back(a);
      }
      if (np > 1) slots += np;
    }
    std::stable_sort(parts.begin(), parts.end(),
                     [](const AsmPart& a, const AsmPart& b) { return a.u1 - a.u0 > b.u1 - b.u0; });
    h->nAsmParts = static_cast<int>(parts.size());
    h->nAsmSlots = slots;
    h->dAsmParts.upload(parts.data(), parts.size(), s);
    h->dAsmUnits.upload(units.data(), units.size(), s);
    if (h->dAsmCount.n < static_cast<size_t>(h->F)) {
      h->dAsmCount.ensure(h->F);
      HIP_CHECK(hipMemsetAsync(h->dAsmCount.p, 0, sizeof(unsigned int) * h->F, s));
    }
  }
  h->dFiOff.upload(fiOff.data(), fiOff.size(), s);
  h->dFiList.upload(fiList.data(), fiList.size(), s);
  h->dFpOff.upload(fpOff.data(), fpOff.size(), s);
  h->dFpList.upload(fpList.data(), fpList.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));
  h->tableRange = inRange;
  h->tableValid = true;
}

// ---- solver context (one solve) --------------------------------------------------------------------------
struct Ctx {
  cvd_handle* h;
  Layout L;
  int KD, KS;
  Table T;
  Items it;
  int nItems;
  size_t n;  // F * B
  int boundDepth0 = 0;
  bool trip = false;  // scene-flow smoothness triplets are part of this problem
  TripletTable TT{};
  bool cross = false;  // dense mode with explicit cross blocks (cvd_cross.h)
};

// Pinned staging buffer `which` with room for n doubles (pageable transfers of the F x B vectors cost ~1 ms each).
static double* pinnedStage(cvd_handle* h, int which, size_t n) {
  if (h->hStageN[which] < n) {
    if (h->hStage[which]) HIP_CHECK(hipHostFree(h->hStage[which]));
    h->hStage[which] = nullptr;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->hStage[which]), std::max<size_t>(n, 1) * sizeof(double)));
    h->hStageN[which] = n;
  }
  return h->hStage[which];
}

static void uploadState(cvd_handle* h, const Layout& L, DevBuf<double>& dst) {
  const size_t nAll = static_cast<size_t>(L.F) * L.B;
  double* x = pinnedStage(h, 1, nAll);
  const int nD = L.nD, nS = L.nS;
  for (int f = 0; f < L.F; ++f) {
    double* xf = &x[static_cast<size_t>(f) * L.B];
    for (int i = 0; i < 7; ++i) xf[i] = h->poseParams[f][i];
    for (int i = 0; i < nD; ++i) xf[7 + i] = h->dparams[static_cast<size_t>(f) * nD + i];
    for (int i = 0; i < nS; ++i) xf[7 + nD + i] = h->sparams[static_cast<size_t>(f) * nS + i];
  }
  dst.upload(x, nAll, h->stream);
  HIP_CHECK(hipStreamSynchronize(h->stream));
}

static void downloadState(cvd_handle* h, const Layout& L, const DevBuf<double>& src) {
  const size_t nAll = static_cast<size_t>(L.F) * L.B;
  double* x = pinnedStage(h, 0, nAll);
  src.download(x, nAll, h->stream);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  const int nD = L.nD, nS = L.nS;
  for (int f = 0; f < L.F; ++f) {
    const double* xf = &x[static_cast<size_t>(f) * L.B];
    for (int i = 0; i < 7; ++i) h->poseParams[f][i] = xf[i];
    for (int i = 0; i < nD; ++i) h->dparams[static_cast<size_t>(f) * nD + i] = xf[7 + i];
    for (int i = 0; i < nS; ++i) h->sparams[static_cast<size_t>(f) * nS + i] = xf[7 + nD + i];
  }
}

static void buildMask(cvd_handle* h, const Layout& L, const cvd_opt_params& p, ProblemKind kind,
                      const std::vector<int>& range) {
  const size_t nAll = static_cast<size_t>(L.F) * L.B;
  HIP_CHECK(hipStreamSynchronize(h->stream));  // an earlier transfer may still read the staging buffer
  double* m = pinnedStage(h, 0, nAll);
  std::fill(m, m + nAll, 0.0);
  for (int f : range) {
    double* mf = &m[static_cast<size_t>(f) * L.B];
    const bool poseFree = (kind == PK_POSE_STEP) && !p.fix_poses;
    for (int i = 0; i < 6; ++i) mf[i] = poseFree ? 1.0 : 0.0;
    mf[6] = (kind == PK_POSE_STEP && p.intr_opt != CVD_INTR_FIXED) ? 1.0 : 0.0;
    const bool depthFree = (kind == PK_NORMALIZE) || !p.fix_depth_xforms;
    for (int i = 0; i < L.nD; ++i) mf[7 + i] = depthFree ? 1.0 : 0.0;
    const bool spatialFree = (kind == PK_POSE_STEP) && !p.fix_spatial_xforms;
    for (int i = 0; i < L.nS; ++i) mf[7 + L.nD + i] = spatialFree ? 1.0 : 0.0;
  }
  h->dMask.upload(m, nAll, h->stream);
}

static void ensureBuffers(Ctx& c) {
  cvd_handle* h = c.h;
  const size_t n = c.n;
  const size_t B = c.L.B;
  h->dX.ensure(n); h->dXc.ensure(n); h->dG.ensure(n); h->dLam.ensure(n); h->dScale.ensure(n);
  h->dDx.ensure(n); h->dR.ensure(n); h->dR1.ensure(n); h->dZ.ensure(n); h->dP0.ensure(n); h->dP1.ensure(n); h->dQ.ensure(n);
  h->dHd.ensure(n);
  {
    // (sharded mode: room for world x chunk frames so that the reduce-scatter / all-gather chunks are equal; the tail
    // frames are zero and stay zero)
    const size_t nPad = static_cast<size_t>(h->framesPadded()) * B;
    const bool grow = h->dH.n < nPad * B;
    h->dH.ensure(nPad * B); h->dMinv.ensure(nPad * B); h->dHd.ensure(nPad);
    if (h->dist() && (grow || nPad > n)) {
      HIP_CHECK(hipMemsetAsync(h->dH.p, 0, nPad * B * sizeof(double), h->stream));
      HIP_CHECK(hipMemsetAsync(h->dMinv.p, 0, nPad * B * sizeof(float), h->stream));
      HIP_CHECK(hipMemsetAsync(h->dHd.p, 0, nPad * sizeof(double), h->stream));
    }
  }
  h->dQPart.ensure(std::max<size_t>(1, static_cast<size_t>(std::max(h->qRows, c.nItems * 2)) * B));
  h->dFdot.ensure(static_cast<size_t>(c.L.F) * 4);
  h->dCostItem.ensure(std::max(1, c.nItems));
  h->dCostFrame.ensure(c.L.F);
  h->dFocal.ensure(static_cast<size_t>(c.L.F) * 2);
  h->dScal.ensure(S_COUNT);
  h->dFc.ensure(c.L.F);
  h->dFail.ensure(1);
  if (!h->dCounters.p) {
    h->dCounters.ensure(8);
    HIP_CHECK(hipMemsetAsync(h->dCounters.p, 0, 8 * sizeof(unsigned int), h->stream));
  }
  if (!h->hScal) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->hScal), S_COUNT * sizeof(double)));
  if (!h->hPcg) {
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->hPcg), 16 * sizeof(double)));
    for (auto& e : h->pcgEvent) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
}

static void launchFrameConsts(Ctx& c, const double* x) {
  hipLaunchKernelGGL(k_frame_consts, dim3((c.L.F + 63) / 64), dim3(64), 0, c.h->stream, c.L, x, c.h->dFc.p);
  HIP_CHECK(hipGetLastError());
}

// The LM loop is latency-bound on its read-backs (a handful of scalars per decision): poll instead of the
// interrupt-driven hipStreamSynchronize / hipEventSynchronize, whose wake-up costs tens of microseconds.
static void spinStream(hipStream_t s) {
  hipError_t e;
  while ((e = hipStreamQuery(s)) == hipErrorNotReady) {}
  HIP_CHECK(e);
}
static void spinEvent(hipEvent_t ev) {
  hipError_t e;
  while ((e = hipEventQuery(ev)) == hipErrorNotReady) {}
  HIP_CHECK(e);
}
static void readScalars(Ctx& c) {
  HIP_CHECK(hipMemcpyAsync(c.h->hScal, c.h->dScal.p, S_COUNT * sizeof(double), hipMemcpyDeviceToHost, c.h->stream));
  spinStream(c.h->stream);
}

// cost only at x: enqueueCost leaves it in S_COST on the device, evalCost also reads it back
static void enqueueCost(Ctx& c, const double* x) {
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
  if (h->dist()) NCCL_CHECK(ncclAllReduce(h->dScal.p + S_COST, h->dScal.p + S_COST, 1, ncclDouble, ncclSum, h->comm, s));
  h->tEnd(slot);
}
static double evalCost(Ctx& c, const double* x) {
  enqueueCost(c, x);
  readScalars(c);
  return c.h->hScal[S_COST];
}

// step statistics (k_step_stats) into the device scalars; the caller reads them back
static void enqueueStats(Ctx& c) {
  cvd_handle* h = c.h;
  const int G = static_cast<int>(std::min<size_t>(128, (c.n + 511) / 512));
  h->dStatPart.ensure(6 * 128);
  hipLaunchKernelGGL(k_step_stats, dim3(G), dim3(256), 0, h->stream, c.n, h->dDx.p, h->dG.p, h->dR.p, h->dLam.p,
                     h->dX.p, h->dHd.p, h->dScal.p, h->dStatPart.p, h->dCounters.p + 2);
  HIP_CHECK(hipGetLastError());
}

// cost + gradient + diagonal blocks at x
// Dense mode with explicit cross blocks (cvd_cross.h) whenever the problem is in its scope.  (Bilinear grids only: at the
// Global level every pixel hits the one vertex -- same-address LDS atomics, 52 ms per assembly measured -- and the 8 x 8
// problem is cheap to solve matrix-free.)
static bool crossScope(cvd_handle* h, const Ctx& c) {
  const bool off = std::getenv("CVD_DENSE_MATRIX_FREE") != nullptr;  // comparison knob (read per solve: the tests toggle it)
  return h->dense && !off && !h->dist() && !h->forceGeneric && c.L.includeStatic && !h->xFa.empty() && c.KS == 0 && fastLoss(c.L) &&
         c.L.N == 1 && c.L.nD > 0 && c.KD == 4 && c.L.intrOpt != CVD_INTR_SHARED && !c.trip && !(c.L.positionRegSqrt > 0.0) &&
         c.L.B <= 256;
}
static CrossPairs crossPairs(cvd_handle* h) {
  return CrossPairs{h->dXFa.p, h->dXFb.p, h->dXRange.p, h->dXSlot.p, static_cast<int>(h->xFa.size())};
}
// X_ab of every undirected pair at the linearisation point x (frame constants in dFc are those of x)
static void launchCrossAssemble(Ctx& c, const double* x) {
  cvd_handle* h = c.h;
  const size_t B = c.L.B, G = c.L.nD;
  h->dXBlocks.ensure(h->xFa.size() * B * B);
  // pose rows / columns: one workgroup per pair; grid x grid: column panels of the largest width that fits the LDS
  const size_t fixedDoubles = 2 * B + 2 * sizeof(FrameConst) / 8;
  int panelW = static_cast<int>(((kMaxLds - 8192) / 8 - fixedDoubles) / G);
  panelW = std::max(1, std::min<int>(panelW, static_cast<int>(G)));
  const int nPanels = static_cast<int>((G + panelW - 1) / panelW);
  panelW = static_cast<int>((G + nPanels - 1) / nPanels);  // (even panels)
  const size_t ldsPose = (fixedDoubles + 56 + 14 * G) * 8;
  const size_t ldsGrid = (fixedDoubles + static_cast<size_t>(panelW) * G) * 8;
  const unsigned nP = static_cast<unsigned>(h->xFa.size());
  CVD_DISPATCH_KD(c.KD, {
    if constexpr (KD == 4) {
      allowLds((k_cross_assemble<KD, false>), ldsPose);
      hipLaunchKernelGGL((k_cross_assemble<KD, false>), dim3(nP), dim3(kCrossThreads), ldsPose, h->stream, c.L, c.T, crossPairs(h),
                         x, h->dFc.p, static_cast<int>(G), h->dXBlocks.p);
      allowLds((k_cross_assemble<KD, true>), ldsGrid);
      hipLaunchKernelGGL((k_cross_assemble<KD, true>), dim3(nP, nPanels), dim3(kCrossThreads), ldsGrid, h->stream, c.L, c.T,
                         crossPairs(h), x, h->dFc.p, panelW, h->dXBlocks.p);
    }
  });
  HIP_CHECK(hipGetLastError());
}

static double evalFull(Ctx& c, const double* x, bool withStats = false) {
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
    CVD_DISPATCH_KD(c.KD, {
      if (h->dense) {
        allowLds((k_assemble_fast<KD, true>), ldsFast);
        hipLaunchKernelGGL((k_assemble_fast<KD, true>), dim3(h->nAsmParts), dim3(kAsmThreads), ldsFast, s, c.L, c.T, x, h->dFc.p,
                           h->dMask.p, h->dMedian.p, h->dRegOwner.p, h->dInRange.p, work, h->dG.p, h->dH.p,
                           h->dCostFrame.p, h->dFocal.p, h->dFocal.p + c.L.F);
      } else {
        allowLds((k_assemble_fast<KD, false>), ldsFast);
        hipLaunchKernelGGL((k_assemble_fast<KD, false>), dim3(h->nAsmParts), dim3(kAsmThreads), ldsFast, s, c.L, c.T, x, h->dFc.p,
                           h->dMask.p, h->dMedian.p, h->dRegOwner.p, h->dInRange.p, work, h->dG.p, h->dH.p,
                           h->dCostFrame.p, h->dFocal.p, h->dFocal.p + c.L.F);
      }
    });
  } else {
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
    NCCL_CHECK(ncclGroupStart());
    NCCL_CHECK(ncclAllReduce(h->dG.p, h->dG.p, c.n, ncclDouble, ncclSum, h->comm, s));
    NCCL_CHECK(ncclAllReduce(h->dCostFrame.p, h->dCostFrame.p, c.L.F, ncclDouble, ncclSum, h->comm, s));
    NCCL_CHECK(ncclReduceScatter(h->dH.p, h->dH.p + static_cast<size_t>(h->rank) * chunkH, chunkH, ncclDouble, ncclSum, h->comm, s));
    NCCL_CHECK(ncclGroupEnd());
    Layout own = c.L;
    own.F = h->ownCount();
    if (own.F > 0)
      hipLaunchKernelGGL(k_extract_diag, dim3((static_cast<size_t>(own.F) * B + 255) / 256), dim3(256), 0, s, own,
                         h->dH.p + static_cast<size_t>(h->ownFirst()) * B * B, h->dHd.p + static_cast<size_t>(h->ownFirst()) * B);
    const size_t chunkD = static_cast<size_t>(h->ownChunk()) * B;
    NCCL_CHECK(ncclAllGather(h->dHd.p + static_cast<size_t>(h->rank) * chunkD, h->dHd.p, chunkD, ncclDouble, h->comm, s));
    h->tEnd(ct);
  } else {
    hipLaunchKernelGGL(k_extract_diag, dim3((c.n + 255) / 256), dim3(256), 0, s, c.L, h->dH.p, h->dHd.p);
  }
  hipLaunchKernelGGL(k_sum2, dim3(1), dim3(256), 0, s, h->dCostFrame.p, c.L.F, h->dCostFrame.p, 0, h->dScal.p, S_COST);
  HIP_CHECK(hipGetLastError());
  if (withStats) {  // |g|_max and |x| of the new point in the same read-back (lam = 0: only those two are used)
    HIP_CHECK(hipMemsetAsync(h->dLam.p, 0, c.n * sizeof(double), s));
    enqueueStats(c);
  }
  readScalars(c);
  return h->hScal[S_COST];
}

static bool coarseFusedConsumers() {
  static const bool v = std::getenv("CVD_COARSE_FUSED") != nullptr;  // experiment: c_f formed inside the consumers
  return v;
}
static bool coarseDenseFused() {
  static const bool v = std::getenv("CVD_COARSE_DENSE_UNFUSED") == nullptr;  // comparison knob: separate k_coarse_dense_apply launch
  return v;
}
static CoarseView coarseView(cvd_handle* h, bool on, bool walk) {
  if (!on) return CoarseView{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  auto& C = h->coarse;
  return CoarseView{C.pos.p, C.wPtr.p, C.wRow.p, walk ? C.Wb.p : nullptr, C.y.p, C.modeActive.p, C.fail.p, C.c.p};
}

// Fills the regulariser Jacobian cache for the products at linearisation point x (before runPcg / the J^T J hook).
static void prepareMatvec(Ctx& c, const double* x) {
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

static void launchMatvec(Ctx& c, const double* x, const double* z, const double* pOld, double* pNew, int useBeta,
                         const double* lam, double* q, bool withCoarse = false) {
  cvd_handle* h = c.h;
  hipStream_t s = h->stream;
  const CoarseView cF = coarseView(h, withCoarse, coarseFusedConsumers());  // z + Z c: the coarse part of the preconditioned residual
  const size_t B = c.L.B;
  if (c.cross) {
    // explicit cross blocks (dense mode): one workgroup per undirected pair streams its B x B block
    hipEvent_t evStart, evStop;
    (void)h->tReserve(KC_MATVEC_PAIRS, evStart, evStop);
    const size_t ldsX = (3 * B + (kCrossThreads / 64) * B + 2 * kCB) * 8;
    const unsigned nP = static_cast<unsigned>(h->xFa.size());
    if (evStart)
      hipExtLaunchKernelGGL(k_cross_matvec, dim3(nP), dim3(kCrossThreads), ldsX, s, evStart, evStop, 0, c.L, crossPairs(h),
                            h->dXBlocks.p, h->dMask.p, z, pOld, h->dScal.p, useBeta, h->dQPart.p, cF);
    else
      hipLaunchKernelGGL(k_cross_matvec, dim3(nP), dim3(kCrossThreads), ldsX, s, c.L, crossPairs(h), h->dXBlocks.p, h->dMask.p, z,
                         pOld, h->dScal.p, useBeta, h->dQPart.p, cF);
    HIP_CHECK(hipGetLastError());
  } else if (c.L.includeStatic && c.nItems > 0) {
    const size_t lds = 6 * B * 8 + 2 * sizeof(FrameConst) + (18 + 4 * 24 + 8 + 2 * kCB) * 8;
    const size_t ldsFast = 6 * B * 8 + 2 * sizeof(FrameConst) + (18 + 32 + static_cast<size_t>(kRedVals) * kRedStride) * 8;
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
      static const int forcedNT = []() { const char* e = std::getenv("CVD_PAIRS_NT"); return e ? std::atoi(e) : 0; }();
      const int nt = forcedNT ? forcedNT : (c.nItems > 2 * h->numCU && c.nItems <= 4 * h->numCU ? 128 : 256);
      // SPEC = 1: the default pipeline's variant (one value parameter, ReproDisparity, Cauchy) fixed at compile time
      const bool spec = c.L.N == 1 && c.L.lossType == CVD_STATIC_REPRO_DISPARITY && c.L.robustKind == 0;
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
#define CVD_LAUNCH_PAIRS_FAST(NTV) do { if (spec) CVD_LAUNCH_PAIRS_FAST_S(NTV, 1); else CVD_LAUNCH_PAIRS_FAST_S(NTV, 0); } while (0)
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
        if (spec) CVD_LAUNCH_PAIRS_DENSE(1);
        else CVD_LAUNCH_PAIRS_DENSE(0);  // (the other reprojection losses / the Huber robustifier: runtime branches)
#undef CVD_LAUNCH_PAIRS_DENSE
      } else if (nt == 128) CVD_LAUNCH_PAIRS_FAST(128);
      else CVD_LAUNCH_PAIRS_FAST(256);
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
  {
    if (B > 512) throw std::runtime_error("frame block larger than 512 unknowns is not supported by k_matvec_finish");
    const size_t lds = 3 * B * 8 + (8 + kCB) * 8;  // xf, pf, qf + red[6] + flag + coarse correction
    // column half of the fused coarse update y <- y - alpha W (Z^T q) (the row half is in k_cg_update)
    const bool fusedCoarse = withCoarse && !h->coarse.denseMode;
    const bool denseFused = withCoarse && h->coarse.denseMode && coarseDenseFused() && !h->dist();  // (needs Z^T q: DenseStep)
    const CoarseColumns cc{h->coarse.pos.p, h->coarse.wPtr.p, h->coarse.wSlot.p, fusedCoarse ? h->coarse.Wb.p : nullptr,
                           h->coarse.wq.p};
    const int slot = h->tBegin(KC_MATVEC_FINISH);
    CVD_DISPATCH_KD(c.KD, {
      hipLaunchKernelGGL((k_matvec_finish<KD>), dim3(c.L.F), dim3(256), lds, s, c.L, x, h->dMask.p, lam,
                         h->dMedian.p, h->dRegOwner.p, h->dInRange.p, c.cross ? h->dXFiOff.p : h->dFiOff.p, h->dFiList.p,
                         h->dQPart.p, z, pOld, pNew, h->dScal.p, h->dCounters.p, useBeta, q, h->dFdot.p,
                         h->dist() ? (h->rank == 0 ? 1 : 2) : 0, c.cross ? static_cast<int>(h->xFa.size()) * 2 : h->qRows,
                         h->regCache, cF, ((fusedCoarse || denseFused) && !h->dist()) ? h->coarse.qc.p : nullptr, cc,
                         c.cross ? h->dH.p : nullptr);
    });
    HIP_CHECK(hipGetLastError());
    if (h->dist()) {
      // per-product exchange: q (F x B doubles) summed over the pair shards, then p.q / alpha on the reduced vector
      const int ct = h->tBegin(KC_COMM_PRODUCT);
      NCCL_CHECK(ncclAllReduce(q, q, c.n, ncclDouble, ncclSum, h->comm, s));
      h->tEnd(ct);
      hipLaunchKernelGGL(k_dot_pq, dim3(c.L.F), dim3(256), 0, s, c.L, pNew, q, h->dScal.p, h->dCounters.p, h->dFdot.p,
                         withCoarse ? h->coarse.qc.p : nullptr, h->coarse.modeActive.p, cc);
      HIP_CHECK(hipGetLastError());
    }
    h->tEnd(slot);
  }
}

// M_f^-1 = (H_ff + diag(lam_f))^-1 for every frame (f32 output).
//   variant 0 (default): blocked sweep on the f64 matrix cores (k_block_inverse_mfma, 16-wide pivot blocks);
//   variant 1: scalar register-resident sweep (4x4 / 6x6 tiles); variant 2: LDS Cholesky (set_generic_kernels).
// The three are kept because they pin each other (tests/test_gpu_block_inverse.py).
static void launchBlockInverseRaw(cvd_handle* h, const Layout& L, const double* dH, const double* dLam, float* dMinv,
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

static void launchBlockInverse(Ctx& c) {
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
static void launchCoarseSetup(Ctx& c, const double* x, int side = 0) {
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

// PCG on (H + diag(lam)) dx = -g with the block-Jacobi preconditioner; returns iterations used.
// Three launches per iteration (pairs product, per-frame finish, per-frame update).  alpha / beta live on
// the device: the last workgroup of k_matvec_finish / k_cg_update reduces the per-frame partial dot products
// (agent-scope release/acquire ticket), so there is neither a scalar kernel nor a host round trip in the loop.
static int runPcg(Ctx& c, const double* x, const std::function<void()>& tail = nullptr) {
  cvd_handle* h = c.h;
  hipStream_t s = h->stream;
  const int F = c.L.F;
  const size_t B = c.L.B;
  if (B > 512) throw std::runtime_error("frame block larger than 512 unknowns is not supported by k_cg_update");
  prepareMatvec(c, x);
  const int nChunks = static_cast<int>((B + 63) / 64);
  const int nThreads = (B > 256 ? 128 : 256) * nChunks;  // (k_cg_update: four segments per row up to B = 256, two beyond)
  double* fd = h->dFdot.p;
  size_t ldsU = (B + nThreads + 48 + 17 * kCB) * 8;
  const double tol2 = c.h->opt.pcg_relative_tolerance * c.h->opt.pcg_relative_tolerance;
  for (int i = 0; i < 9; ++i) h->hPcg[i] = 0.0;  // device progress mirror (pcgFinishScalars): nothing applied yet
  const bool coarse = h->coarseOn;
  double* rc = coarse ? h->coarse.rc.p : nullptr;
  // Coarse level per iteration: y = W Z^T r is kept up to date inside k_cg_update (CoarseStep: y <- y - alpha W Z^T q,
  // |y|^2 closes r^T z), so only c = W^T y remains as a launch; the first residual goes through k_coarse_apply_w.
  static const bool unfusedEnv = std::getenv("CVD_COARSE_UNFUSED_Y") != nullptr;  // comparison: separate y = W Z^T r launch
  const bool denseCoarse = coarse && h->coarse.denseMode;
  const bool unfusedY = unfusedEnv || denseCoarse;  // (the dense level has no W to recur on: Z^T r is restricted every iteration)
  auto coarseC = [&](int init) {
    if (c.L.positionRegSqrt > 0.0 || c.trip || !coarseFusedConsumers())
      hipLaunchKernelGGL(k_coarse_apply_wt, dim3(F), dim3(256), 0, s, coarseView(h, true, true), F, h->coarse.c.p,
                         h->dScal.p, init);
  };
  auto coarseApply = [&](int init) {
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
  const DenseStep dsOff{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // dense level: c <- c - alpha A_c^-1 Z^T q inside k_cg_update (F extra workgroups) instead of a launch of its own
  const bool denseFused = denseCoarse && coarseDenseFused() && !h->dist();
  const DenseStep dsOn = denseFused ? DenseStep{h->coarse.denseInv.p, h->coarse.qc.p, h->coarse.rc.p, h->coarse.c.p,
                                                h->coarse.dotPart.p, h->coarse.modeActive.p, h->coarse.fail.p}
                                    : dsOff;
  if (denseFused) ldsU = std::max(ldsU, (static_cast<size_t>(F) * kCB + nThreads + 16) * 8);  // (its workgroups: Z^T q + partial sums)
  hipLaunchKernelGGL(k_cg_update, dim3(F), dim3(nThreads), ldsU, s, c.L, 1, h->dG.p, h->dMinv.p, h->dP0.p, h->dQ.p,
                     h->dScal.p, h->dCounters.p + 1, h->dDx.p, h->dR.p, h->dZ.p, fd + F, fd + 2 * F, tol2, rc,
                     h->coarse.modeActive.p, h->hPcg, csOff, dsOff);
  if (coarse) coarseApply(1);
  HIP_CHECK(hipGetLastError());
  double* pOld = h->dP0.p;
  double* pNew = h->dP1.p;
  const int maxIt = std::max(1, c.h->opt.pcg_max_iterations);
  const int every = std::max(1, c.h->opt.pcg_check_every);
  // Convergence is decided on the device (S_DONE, set by the last workgroup of k_cg_update); the host enqueues
  // batches of `every` iterations and reads the control scalars of batch b only before enqueuing batch b + 2, so
  // the stream never drains while the host waits.  Iterations enqueued past convergence return immediately.
  // (profiling aid: CVD_PCG_LOCKSTEP=1 checks after every iteration and never runs ahead, so that per-launch
  // counter averages contain no early-exit launches)
  static const bool lockstep = std::getenv("CVD_PCG_LOCKSTEP") != nullptr;
  const size_t firstTimerSlot = h->evUsed;
  int enq = 0;
  auto enqueueIteration = [&](int it, int useBeta) {
    h->curPcgIter = it;
    launchMatvec(c, x, h->dZ.p, pOld, pNew, useBeta, h->dLam.p, h->dQ.p, coarse);
    const int slot = h->tBegin(KC_CG_UPDATE);
    hipLaunchKernelGGL(k_cg_update, dim3(denseFused ? F + (F + kDenseFramesPerGroup - 1) / kDenseFramesPerGroup : F), dim3(nThreads), ldsU, s, c.L, 0, h->dG.p, h->dMinv.p, pNew,
                       h->dQ.p, h->dScal.p, h->dCounters.p + 1, h->dDx.p, h->dR.p, h->dZ.p, fd + F, fd + 2 * F, tol2,
                       (coarse && unfusedY && !denseFused) ? rc : nullptr, h->coarse.modeActive.p, h->hPcg, csOn, dsOn);
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
  static const int runAheadEnv = []() { const char* e = std::getenv("CVD_PCG_RUN_AHEAD"); return e ? std::max(1, std::atoi(e)) : 2; }();
  const int kRunAhead = lockstep ? 1 : runAheadEnv;
  volatile double* prog = h->hPcg;
  while (enq < maxIt) {
    if (enq >= kRunAhead - 1) {
      const int need = enq - kRunAhead + 1;
      for (unsigned long long spins = 0; static_cast<int>(prog[0]) - 1 < need; ++spins) {
        if ((spins & 0xFFFFF) == 0xFFFFF) {
          // never spin forever on a mirror that cannot advance: a faulted stream reports here, and an idle stream
          // whose iterations did not publish progress is a logic error
          const hipError_t e = hipStreamQuery(s);
          if (e != hipErrorNotReady) {
            HIP_CHECK(e);
            if (static_cast<int>(prog[0]) - 1 < need) throw std::runtime_error("PCG progress mirror stalled");
          }
        }
      }
      if (prog[1 + (need & 7)] != 0.0) break;
    }
    enqueueIteration(enq, enq > 0 ? 1 : 0);
    ++enq;
    if (h->opt.verbose >= 2) {  // development trace: per-iteration scalars (synchronises every iteration)
      readScalars(c);
      std::printf("    pcg %3d  rz %.6e  rzpart %.6e  alpha %.6e  beta %.6e  pq %.6e  done %g\n", enq - 1, h->hScal[S_RZ],
                  h->hScal[S_RZPART], h->hScal[S_ALPHA], h->hScal[S_BETA], h->hScal[S_PQ], h->hScal[S_DONE]);
    }
  }
  h->curPcgIter = -1;
  if (tail) tail();  // follow-up work that does not need the host's decision rides on the same read-back
  readScalars(c);    // drains the stream; S_DONE / S_ITERS are final
  if (h->hScal[S_DONE] == 2.0) throw std::runtime_error("PCG produced NaN");
  const int iters = static_cast<int>(h->hScal[S_ITERS]);
  h->tDropFrom(firstTimerSlot, iters);
  return iters;
}

// Scene-flow smoothness triplets of this problem (reference lib/PoseOptimizer.cpp:899): on when a weight is > 0.
static bool wantsTriplets(const cvd_opt_params& p, ProblemKind kind) {
  return kind == PK_POSE_STEP && (p.smooth_static_weight > 0.0 || p.smooth_dynamic_weight > 0.0);
}
static void bindTriplets(Ctx& c, const cvd_opt_params& p, ProblemKind kind) {
  cvd_handle* h = c.h;
  c.trip = wantsTriplets(p, kind) && c.L.includeStatic;
  if (!c.trip) return;
  c.TT = TripletTable{h->dTNdc.p, h->dTDsrc.p, h->dTStatic.p, h->dTOff.p, h->dTCenter.p, h->dTSlot.p,
                      static_cast<int>(h->tripActive.size()),
                      static_cast<int>(p.smooth_loss_type),  // (CVD_SMOOTH_* == kSmooth*)
                      std::sqrt(std::max(0.0, p.smooth_static_weight)), std::sqrt(std::max(0.0, p.smooth_dynamic_weight))};
}

// Median of every frame's source depth (reference lib/PoseOptimizer.cpp:1363-1375: std::nth_element at size / 2), the
// reference value of the scale regulariser.  On the device: one segmented radix sort of the depth maps that are resident
// anyway, element n / 2 of every sorted frame -- the same order statistic, bit for bit.  (The host nth_element this
// replaces cost 0.25 ms per 384x224 frame inside cvd_set_depth: 70 of the 89 ms a 300-frame upload took.)
__global__ void k_pick_median(const float* __restrict__ sorted, size_t n, int first, int count, float* __restrict__ median) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) median[first + i] = sorted[static_cast<size_t>(i) * n + n / 2];
}
static void refreshMedians(cvd_handle* h) {
  if (!h->medianDirty) return;
  hipStream_t s = h->stream;
  const size_t n = static_cast<size_t>(h->W) * h->H;
  const int F = h->F;
  h->dMedian.ensure(F);
  const int PB = static_cast<int>(std::max<size_t>(1, std::min<size_t>(F, (size_t(1) << 28) / (n * sizeof(float)))));  // <= 256 MiB
  if (static_cast<size_t>(PB) * n > 0xFFFFFFFFull) throw std::runtime_error("depth maps too large for the median sort");
  DevBuf<float> sorted;
  DevBuf<unsigned int> seg;
  DevBuf<unsigned char> tmp;
  sorted.ensure(static_cast<size_t>(PB) * n);
  std::vector<unsigned int> segH(PB + 1);
  for (int i = 0; i <= PB; ++i) segH[i] = static_cast<unsigned int>(static_cast<size_t>(i) * n);
  seg.upload(segH.data(), segH.size(), s);
  size_t tmpBytes = 0;
  HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, tmpBytes, h->dDepth.p, sorted.p, static_cast<unsigned int>(PB * n),
                                               static_cast<unsigned int>(PB), seg.p, seg.p + 1, 0, 32, s));
  tmp.ensure(tmpBytes);
  for (int p0 = 0; p0 < F; p0 += PB) {
    const int nb = std::min(PB, F - p0);
    size_t tb = tmpBytes;
    HIP_CHECK(rocprim::segmented_radix_sort_keys(tmp.p, tb, h->dDepth.p + static_cast<size_t>(p0) * n, sorted.p,
                                                 static_cast<unsigned int>(nb * n), static_cast<unsigned int>(nb), seg.p,
                                                 seg.p + 1, 0, 32, s));
    hipLaunchKernelGGL(k_pick_median, dim3((nb + 63) / 64), dim3(64), 0, s, sorted.p, n, p0, nb, h->dMedian.p);
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipStreamSynchronize(s));  // (the temporaries go out of scope)
  h->medianDirty = false;
}

static void solve(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, ProblemKind kind) {
  const double t0 = nowSeconds();
  if (h->F <= 0) throw std::runtime_error("no video set");
  if (!h->poseParamsValid) posesToParams(h);
  const std::vector<int> range = rangeOf(p, h->F);
  struct WorkerGuard {  // (also on the exception paths)
    cvd_handle* h;
    ~WorkerGuard() { h->sideWorker.waitNoThrow(); }
  } workerGuard{h};
  static const bool dbgSetup = std::getenv("CVD_DEBUG_SETUP") != nullptr;  // development: where a solve's fixed cost goes
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
  compileTable(h, range, wantsTriplets(p, kind), kind == PK_NORMALIZE && c.L.includeStatic);
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
  double xCost = evalFull(c, h->dX.p);
  tEval += nowSeconds() - te;
  sum.initial_cost = xCost;
  phase("first eval");

  auto stats = [&]() {
    enqueueStats(c);
    readScalars(c);
  };
  HIP_CHECK(hipMemsetAsync(h->dLam.p, 0, c.n * sizeof(double), s));
  stats();
  double gmax = h->hScal[S_GMAX];
  double xNorm = std::sqrt(h->hScal[S_XX]);
  {
    // number of active unknowns
    std::vector<double> hd(c.n);
    h->dHd.download(hd.data(), c.n, s);
    HIP_CHECK(hipStreamSynchronize(s));
    int na = 0;
    for (double v : hd) na += (v != 0.0);
    sum.num_parameters = na;
  }
  phase("stats");

  double radius = Ceres::initial_radius;
  double decrease = 2.0;
  int invalid = 0, iteration = 0, termination = 1;
  static const int kCoarseRebuildIters = []() {
    const char* e = std::getenv("CVD_COARSE_REBUILD_ITERS");  // development knob
    return e ? std::max(1, std::atoi(e)) : 16;
  }();
  int coarseAge = -1, cgAfterRefresh = 0, cgExcess = 0;  // coarse level: LM iterations since the last rebuild
  static const bool asyncCoarse = std::getenv("CVD_COARSE_SYNC") == nullptr;  // (development knob: rebuild in line)
  bool coarsePending = false;  // a rebuild is running on the side stream
  int factorUses = 0;          // PCG solves done with the factor in use
  double lastRelChange = 1.0;  // relative cost change of the last accepted step
  static const double asyncMaxChange = []() { const char* e = std::getenv("CVD_COARSE_ASYNC_MAX_CHANGE"); return e ? std::atof(e) : 1e-3; }();
  bool freshFactor = false;    // the factor was installed right before this iteration's PCG
  static const bool carryAcrossLevels = std::getenv("CVD_COARSE_CARRY_LEVELS") != nullptr;  // comparison knob
  if (h->coarseOn && h->coarse.denseMode && h->coarse.denseReady && (h->coarse.denseForB == c.L.B || carryAcrossLevels)) {
    // Dense level: the inverse left by the previous solve on this handle (the previous coarse-to-fine level, or the last
    // optimisation of the same video) is a perfectly good SPD preconditioner to start with -- its ~6 ms rocSOLVER rebuild
    // is not paid in line but started beside the first PCG and installed for the second LM iteration.
    coarseAge = 0;
    cgExcess = kCoarseRebuildIters;
  }
  // The dense level's rebuild (~6.5 ms of dependent rocSOLVER kernels) outlasts one PCG solve (~5 ms at 4140 pairs): it is
  // installed after the SECOND solve that runs beside it (a fixed lag, not an event query: the iteration sequence stays
  // a function of the data alone), so that the main stream never waits for it.
  static const int kDenseInstallLag = []() { const char* e = std::getenv("CVD_COARSE_DENSE_LAG"); return e ? std::max(1, std::atoi(e)) : 2; }();
  int pendingSolves = 0;  // PCG solves run since the pending rebuild was started
  auto installPendingCoarse = [&]() {
    if (!coarsePending) return;
    if (h->coarse.denseMode && ++pendingSolves < kDenseInstallLag) return;
    pendingSolves = 0;
    h->sideWorker.wait();  // (the helper thread has finished enqueuing: normally long ago)
    HIP_CHECK(hipStreamWaitEvent(s, h->evCoarseDone, 0));  // (device-side wait: the host does not block)
    std::swap(h->coarse.Wb.p, h->coarse.Wb2.p);
    std::swap(h->coarse.Wb.n, h->coarse.Wb2.n);
    std::swap(h->coarse.fail.p, h->coarse.fail2.p);
    std::swap(h->coarse.fail.n, h->coarse.fail2.n);
    std::swap(h->coarse.denseInv.p, h->coarse.denseInv2.p);
    std::swap(h->coarse.denseInv.n, h->coarse.denseInv2.n);
    coarsePending = false;
    freshFactor = true;
    cgExcess = 0;
    coarseAge = 0;
    factorUses = 0;
    if (h->coarse.denseMode) { h->coarse.denseReady = true; h->coarse.denseForB = c.L.B; }
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

  if (sum.num_parameters == 0 || (gmax <= Ceres::gradient_tolerance && !h->opt.force_iterations)) {
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
      {
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
          // (the dense level's rocSOLVER inversion is a chain of small kernels, ~6 ms at 2400 unknowns, that hardly
          // occupies the device: always beside the PCG once a first inverse exists)
          if ((lastRelChange < asyncMaxChange || (h->coarse.denseMode && coarseAge >= 0)) && asyncCoarse && h->opt.coarse_level != 2 && !h->dist()) {
            h->dFc2.ensure(c.L.F);
            HIP_CHECK(hipEventRecord(h->evCoarseIn, s));
            HIP_CHECK(hipStreamWaitEvent(h->stream2, h->evCoarseIn, 0));
            launchCoarseSetup(c, h->dX.p, 1);
            if (!h->coarse.denseMode) HIP_CHECK(hipEventRecord(h->evCoarseDone, h->stream2));  // (dense: recorded by the job)
            coarsePending = true;
            cgExcess = 0;
          } else {
            const int slot = h->tBegin(KC_INVERSE);  // preconditioner construction, same class as the block inverse
            launchCoarseSetup(c, h->dX.p);
            h->tEnd(slot);
            if (h->coarse.denseMode) { h->coarse.denseReady = true; h->coarse.denseForB = c.L.B; }
            coarseAge = 0;
            cgExcess = 0;
            freshFactor = true;
            factorUses = 0;
          }
        } else {
          ++coarseAge;
        }
      }
      // one read-back for the PCG result, the step statistics and the cost of the candidate point (the
      // candidate is formed speculatively; it is simply not used when the model decrease is invalid)
      const int cgIters = runPcg(c, h->dX.p, [&]() {
        enqueueStats(c);
        hipLaunchKernelGGL(k_apply_step, dim3((c.n + 255) / 256), dim3(256), 0, s, c.L, c.boundDepth0, h->dX.p,
                           h->dDx.p, h->dXc.p);
        HIP_CHECK(hipGetLastError());
        enqueueCost(c, h->dXc.p);
      });
      static const bool dbgCoarse = std::getenv("CVD_DEBUG_COARSE") != nullptr;
      if (dbgCoarse && h->coarseOn && h->coarse.denseMode) {
        HIP_CHECK(hipStreamSynchronize(s));
        int fl[2] = {-1, -1};
        HIP_CHECK(hipMemcpy(&fl[0], h->coarse.fail.p, 4, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(&fl[1], h->coarse.fail2.p, 4, hipMemcpyDeviceToHost));
        const size_t nn = static_cast<size_t>(c.L.F) * kCB;
        std::vector<float> dg(nn);
        HIP_CHECK(hipMemcpy2D(dg.data(), 4, h->coarse.denseInv.p, (nn + 1) * 4, 4, nn, hipMemcpyDeviceToHost));
        double tr = 0.0;
        for (float v : dg) tr += v;
        std::vector<double> cc(nn);
        HIP_CHECK(hipMemcpy(cc.data(), h->coarse.c.p, nn * 8, hipMemcpyDeviceToHost));
        double cn = 0.0;
        for (double v : cc) cn += v * v;
        fprintf(stderr, "[coarse dbg] it %d pcg %d fail %d fail2 %d inv %p trace %.10e |c| %.6e pending %d graph %d/%p\n", iteration, cgIters, fl[0], fl[1],
                (void*)h->coarse.denseInv.p, tr, std::sqrt(cn), coarsePending ? 1 : 0, h->coarse.denseGraphState, (void*)h->coarse.denseGraph);
      }
      // a rebuild still pending past this point (the dense level's fixed lag) must have read its inputs before the next
      // evaluation rewrites them: the side stream competes with a main stream that is never idle, so "enqueued 4 ms ago"
      // is not "executed" (device-side wait, satisfied long ago in the normal case)
      if (coarsePending) HIP_CHECK(hipStreamWaitEvent(s, h->evCoarseRead, 0));
      if (freshFactor) cgAfterRefresh = cgIters;
      else cgExcess += std::max(0, cgIters - cgAfterRefresh);
      freshFactor = false;
      ++factorUses;
      installPendingCoarse();
      tLin += nowSeconds() - tl;
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
      if (h->opt.force_iterations) stop = false;
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
        const double chk = evalFull(c, h->dX.p, true);
        (void)chk;
        tEval += nowSeconds() - te;
        gmax = h->hScal[S_GMAX];
        xNorm = std::sqrt(h->hScal[S_XX]);
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
        if (gmax <= Ceres::gradient_tolerance && !h->opt.force_iterations) { termination = 0; break; }
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
  phase("LM loop");
  h->sideWorker.wait();  // (a rebuild started beside the last PCG: its enqueuing must not outlive this frame)
  downloadState(h, c.L, h->dX);
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

// poseOptimizationStep, reference lib/PoseOptimizer.cpp:890-990
static void poseOptimizationStep(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg) {
  solve(h, p, depthDeformReg, PK_POSE_STEP);
  paramsToPoses(h, p);
}

// poseOptimization, reference lib/PoseOptimizer.cpp:788-888
static void poseOptimization(cvd_handle* h, const cvd_opt_params& p) {
  posesToParams(h);
  h->records.clear();
  int ctfRows = p.ctf_long, ctfCols = p.ctf_short;
  int dsoRows = p.dso_long, dsoCols = p.dso_short;
  if (h->aspect >= 1.f) {
    std::swap(ctfCols, ctfRows);
    std::swap(dsoCols, dsoRows);
  }
  int initGrid[3] = {1, 1, 1};
  if (h->ddesc.depth_type == CVD_DEPTH_GRID)
    for (int i = 0; i < 3; ++i) initGrid[i] = h->ddesc.grid_size[i];
  // largest frame block of the schedule: validated before any state changes, then used to reserve the device buffers
  const int Nval = (h->ddesc.depth_type == CVD_DEPTH_IDENTITY) ? 0 : valueNumParams(h->ddesc.value_xform);
  size_t nDmax = static_cast<size_t>(h->nD());
  if (p.coarse_to_fine && h->ddesc.depth_type != CVD_DEPTH_IDENTITY && p.num_steps > 1)
    nDmax = std::max(nDmax, static_cast<size_t>(ctfCols) * ctfRows * initGrid[2] * Nval);
  size_t nSmax = p.deferred_spatial_opt ? static_cast<size_t>(dsoRows) * dsoCols * 2 : static_cast<size_t>(h->nS());
  const size_t Bmax = 7 + nDmax + nSmax;
  checkFrameBlock(Bmax, "poseOptimization (largest level of the coarse-to-fine / deferred-spatial schedule)");
  if (p.deferred_spatial_opt) {
    cvd_xform_desc sd{};
    sd.type = CVD_XFORM_SPATIAL;
    sd.spatial_type = CVD_SPATIAL_IDENTITY;
    resetXforms(h, sd, true);
  }
  {
    // Reserve the device buffers for the largest block of the schedule up front: growing them level by level
    // costs a hipFree + hipMalloc (milliseconds, with a device synchronisation) per buffer and level.
    const size_t n = static_cast<size_t>(h->F) * Bmax;
    h->dX.ensure(n); h->dXc.ensure(n); h->dG.ensure(n); h->dLam.ensure(n); h->dScale.ensure(n);
    h->dDx.ensure(n); h->dR.ensure(n); h->dR1.ensure(n); h->dZ.ensure(n); h->dP0.ensure(n); h->dP1.ensure(n);
    h->dQ.ensure(n); h->dHd.ensure(n); h->dMask.ensure(n);
    h->dH.ensure(n * Bmax); h->dMinv.ensure(n * Bmax);
    // one undirected work item per ~768 constraints and pair: bounded by pairs + constraints / 768
    h->dQPart.ensure((static_cast<size_t>(h->P) + static_cast<size_t>(h->C / (h->dense ? kDenseChunk : kListChunk)) + 1) * 2 * Bmax);
  }
  cvd_solve_summary total{};
  auto accumulate = [&](const cvd_solve_summary& s, bool first) {
    if (first) total.initial_cost = s.initial_cost;
    total.num_iterations += s.num_iterations;
    total.num_successful_steps += s.num_successful_steps;
    total.total_linear_iterations += s.total_linear_iterations;
    total.total_seconds += s.total_seconds;
    total.evaluate_seconds += s.evaluate_seconds;
    total.linear_solve_seconds += s.linear_solve_seconds;
    total.final_cost = s.final_cost;
    total.termination = s.termination;
    total.num_residual_blocks = s.num_residual_blocks;
    total.num_parameters = s.num_parameters;
  };
  for (int step = 0; step < p.num_steps; ++step) {
    const double stepIter = (p.num_steps > 1 ? step / double(p.num_steps - 1) : 0.0);
    double depthDeformReg = p.depth_deform_reg_final;
    if (p.graduate_depth_deform_reg) {
      const double a = std::log(p.depth_deform_reg_initial), b = std::log(p.depth_deform_reg_final);
      depthDeformReg = std::exp(a + (b - a) * stepIter);
    }
    poseOptimizationStep(h, p, depthDeformReg);
    accumulate(h->summary, step == 0);
    if (p.coarse_to_fine && step < p.num_steps - 1) {
      const double ctfIter = (step + 1) / double(p.num_steps - 1);
      cvd_xform_desc nd = h->ddesc;
      if (nd.depth_type == CVD_DEPTH_GLOBAL) nd.depth_type = CVD_DEPTH_GRID;
      nd.grid_size[0] = static_cast<int>(initGrid[0] + (ctfCols - initGrid[0]) * ctfIter + 0.5);
      nd.grid_size[1] = static_cast<int>(initGrid[1] + (ctfRows - initGrid[1]) * ctfIter + 0.5);
      nd.grid_size[2] = initGrid[2];
      gridXformSplit(h, nd);
    }
  }
  if (p.deferred_spatial_opt) {
    cvd_xform_desc sd{};
    sd.type = CVD_XFORM_SPATIAL;
    sd.spatial_type = CVD_SPATIAL_BICUBIC_GRID;
    sd.grid_size[1] = dsoRows;
    sd.grid_size[0] = dsoCols;
    resetXforms(h, sd, true);
    poseOptimizationStep(h, p, p.depth_deform_reg_final);
    accumulate(h->summary, false);
  }
  h->summary = total;
}

// normalizeDepth, reference lib/PoseOptimizer.cpp:992-1147 (default: from the first frame)
static void normalizeDepth(cvd_handle* h, const cvd_opt_params& p) {
  posesToParams(h);
  h->records.clear();
  solve(h, p, p.depth_deform_reg_initial, PK_NORMALIZE);
  const std::vector<int> range = rangeOf(p, h->F);
  if (p.normalize_depth_from_first_frame && !range.empty()) {
    const int nD = h->nD();
    const int first = range.front();
    for (int f : range)
      if (f != first)
        std::copy(h->dparams.begin() + static_cast<size_t>(first) * nD,
                  h->dparams.begin() + static_cast<size_t>(first + 1) * nD,
                  h->dparams.begin() + static_cast<size_t>(f) * nD);
  }
}

static void evaluate(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, const double* pose7,
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
  compileTable(h, range, wantsTriplets(p, PK_POSE_STEP));
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
      NCCL_CHECK(ncclAllGather(h->dH.p + static_cast<size_t>(h->rank) * chunk, h->dH.p, chunk, ncclDouble, h->comm, s));
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

// ---- constraint sampling (cvd_sampling.h) -----------------------------------------------------------
// triplet == false: keyFrames = 2 x n frames (a, b) of the directed pairs, flow / mask = a -> b.
// triplet == true : keyFrames = n centre frames c, flow / mask = c -> c-1, flow2 / mask2 = c -> c+1.
static void sampleConstraints(cvd_handle* h, bool triplet, int num, const int32_t* keyFrames, const float* corner,
                              const float* flow, const uint8_t* mask, const float* flow2, const uint8_t* mask2,
                              const float* dyn, int dw, int dh, int matchSeparation, float minDynamicDistance,
                              int64_t* offsets) {
  if (h->F <= 0) throw std::runtime_error("no video set");
  if (matchSeparation < 0) throw std::runtime_error("matchSeparation must be >= 0");
  const int W = h->W, H = h->H;
  const size_t npx = static_cast<size_t>(W) * H;
  const int width = triplet ? 3 : 2;  // float2 per constraint
  if ((npx + 31) / 32 * 4 > kMaxLds) throw std::runtime_error("image too large for the LDS-resident sampling mask");
  for (int i = 0; i < (triplet ? 1 : 2) * num; ++i) {
    const int f = keyFrames[i];
    if (f < 0 || f >= h->F || (triplet && (f < 1 || f + 1 >= h->F))) throw std::runtime_error("sampling frame out of range");
  }
  hipStream_t s = h->stream;
  DevBuf<float> dCorner, dDyn;
  DevBuf<float2> dFlow, dFlow2, dSlab;
  DevBuf<unsigned char> dMaskS, dMaskS2, dTmp;
  DevBuf<int> dKeysF;
  DevBuf<unsigned long long> dKeys, dKeysOut;
  DevBuf<unsigned int> dNValid, dCount, dSeg;
  DevBuf<long long> dOff;
  dCorner.upload(corner, static_cast<size_t>(h->F) * npx, s);
  if (dyn) dDyn.upload(dyn, static_cast<size_t>(h->F) * dw * dh, s);
  dKeysF.upload(keyFrames, static_cast<size_t>(num) * (triplet ? 1 : 2), s);
  dFlow.upload(reinterpret_cast<const float2*>(flow), static_cast<size_t>(num) * npx, s);
  dMaskS.upload(mask, static_cast<size_t>(num) * npx, s);
  if (triplet) {
    dFlow2.upload(reinterpret_cast<const float2*>(flow2), static_cast<size_t>(num) * npx, s);
    dMaskS2.upload(mask2, static_cast<size_t>(num) * npx, s);
  }
  SamplingArgs A{W, H, h->invAspect, matchSeparation, minDynamicDistance, dCorner.p, dyn ? dDyn.p : nullptr,
                 dyn ? dw : W, dyn ? dh : H};
  // batches: keys (2 x 8 B) and the output slab (8 B x width) per pixel, ~1 GiB at a time
  const int PB = static_cast<int>(std::max<size_t>(1, std::min<size_t>(num, (size_t(1) << 30) / (npx * (16 + 8 * width)))));
  if (static_cast<size_t>(PB) * npx > 0xFFFFFFFFull) throw std::runtime_error("sampling batch too large");
  dKeys.ensure(static_cast<size_t>(PB) * npx);
  dKeysOut.ensure(static_cast<size_t>(PB) * npx);
  dSlab.ensure(static_cast<size_t>(PB) * npx * width);
  dNValid.ensure(PB);
  dCount.ensure(PB);
  std::vector<unsigned int> seg(PB + 1);
  for (int i = 0; i <= PB; ++i) seg[i] = static_cast<unsigned int>(static_cast<size_t>(i) * npx);
  dSeg.upload(seg.data(), seg.size(), s);
  size_t tmpBytes = 0;
  HIP_CHECK(rocprim::segmented_radix_sort_keys_desc(nullptr, tmpBytes, dKeys.p, dKeysOut.p,
                                                    static_cast<unsigned int>(static_cast<size_t>(PB) * npx),
                                                    static_cast<unsigned int>(PB), dSeg.p, dSeg.p + 1, 0, 64, s));
  dTmp.ensure(tmpBytes);
  DevBuf<float2>& result = triplet ? h->dSampledTrip : h->dSampledLoc;
  std::vector<long long> off(num + 1, 0);
  std::vector<unsigned int> cnt(PB);
  result.ensure(1);
  for (int p0 = 0; p0 < num; p0 += PB) {
    const int nb = std::min(PB, num - p0);
    HIP_CHECK(hipMemsetAsync(dNValid.p, 0, sizeof(unsigned int) * nb, s));
    const dim3 gridC(static_cast<unsigned>((npx + 255) / 256), nb);
    if (triplet)
      hipLaunchKernelGGL(k_fc_triplet_candidates, gridC, dim3(256), 0, s, A, p0, dKeysF.p, dFlow.p, dMaskS.p, dFlow2.p,
                         dMaskS2.p, dKeys.p, dNValid.p);
    else
      hipLaunchKernelGGL(k_fc_candidates, gridC, dim3(256), 0, s, A, p0, dKeysF.p, dFlow.p, dMaskS.p, dKeys.p, dNValid.p);
    HIP_CHECK(hipGetLastError());
    size_t tb = tmpBytes;
    HIP_CHECK(rocprim::segmented_radix_sort_keys_desc(dTmp.p, tb, dKeys.p, dKeysOut.p,
                                                      static_cast<unsigned int>(static_cast<size_t>(nb) * npx),
                                                      static_cast<unsigned int>(nb), dSeg.p, dSeg.p + 1, 0, 64, s));
    const size_t ldsBytes = (npx + 31) / 32 * 4;
    if (triplet) {
      allowLds(k_fc_greedy<true>, ldsBytes);
      hipLaunchKernelGGL(k_fc_greedy<true>, dim3(nb), dim3(64), ldsBytes, s, A, p0, dKeysOut.p, dNValid.p, dFlow.p,
                         dFlow2.p, dSlab.p, dCount.p);
    } else {
      allowLds(k_fc_greedy<false>, ldsBytes);
      hipLaunchKernelGGL(k_fc_greedy<false>, dim3(nb), dim3(64), ldsBytes, s, A, p0, dKeysOut.p, dNValid.p, dFlow.p,
                         static_cast<const float2*>(nullptr), dSlab.p, dCount.p);
    }
    HIP_CHECK(hipGetLastError());
    dCount.download(cnt.data(), nb, s);
    HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < nb; ++i) off[p0 + i + 1] = off[p0 + i] + cnt[i];
    // grow the result buffer and compact this batch into it
    const size_t total = static_cast<size_t>(off[p0 + nb]) * width;
    if (total > result.n) {
      DevBuf<float2> bigger;
      bigger.ensure(std::max<size_t>(total, result.n * 2));
      if (off[p0] > 0)
        HIP_CHECK(hipMemcpyAsync(bigger.p, result.p, sizeof(float2) * off[p0] * width, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
      std::swap(bigger.p, result.p);
      std::swap(bigger.n, result.n);
    }
    dOff.upload(off.data(), off.size(), s);
    hipLaunchKernelGGL(k_fc_compact, dim3(16, nb), dim3(256), 0, s, static_cast<int>(npx), width, p0, dOff.p, dSlab.p,
                       result.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
  }
  (triplet ? h->sampledTripOff : h->sampledOff) = off;
  for (int i = 0; i <= num; ++i) offsets[i] = off[i];
}

// ---- dense consumers of the result (cvd_dense.h) ----------------------------------------------
// kind 0: DepthXform::apply -> f32 [n][H][W]; 1: GridDepthXform::paramMap -> f64 [n][H][W][N];
// 2: SpatialXform::warp -> f32 [n][h][w][2] for the raster (w, h).  Host buffer out; the device buffer is kept for
// the next call.  Returns the kernel time in ms through *kernelMs when asked (HIP events on the solver stream).
static void denseMaps(cvd_handle* h, int kind, int first, int count, int w, int hh, void* out, double* kernelMs) {
  if (h->F <= 0) throw std::runtime_error("no video set");
  if (first < 0 || count < 0 || first + count > h->F) throw std::runtime_error("frame range out of bounds");
  if (!h->poseParamsValid) posesToParams(h);
  cvd_opt_params p;
  cvd_opt_params_default(&p);
  Layout L = makeLayout(h, p, 0.0, PK_POSE_STEP);
  int KD, KS;
  tapCounts(L, KD, KS);
  if (kind == 1 && L.depthType != CVD_DEPTH_GRID)
    throw std::runtime_error("Parameter map not implemented for this transform type.");  // reference :422-425
  if (kind != 2) { w = h->W; hh = h->H; }
  if (w < 2 || hh < 2) throw std::runtime_error("raster too small");
  uploadState(h, L, h->dX);
  hipStream_t s = h->stream;
  const size_t pixels = static_cast<size_t>(count) * hh * w;
  const size_t bytes = pixels * (kind == 0 ? sizeof(float) : kind == 1 ? sizeof(double) * std::max(L.N, 1) : sizeof(float2));
  h->dDense.ensure((bytes + 7) / 8);
  if (count == 0) return;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernelMs) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventRecord(e0, s)); }
  const dim3 grid((w * hh + 255) / 256, 1, count), block(256);
  if (kind == 0) {
    CVD_DISPATCH_KD(KD, {
      hipLaunchKernelGGL((k_apply_depth<KD>), grid, block, 0, s, L, w, hh, first, h->dDepth.p, h->dX.p,
                         reinterpret_cast<float*>(h->dDense.p));
    });
  } else if (kind == 1) {
    CVD_DISPATCH_KD(KD, {
      hipLaunchKernelGGL((k_param_map<KD>), grid, block, 0, s, L, w, hh, first, h->dDepth.p, h->dX.p, h->dDense.p);
    });
  } else {
    if (KS == 0) hipLaunchKernelGGL((k_warp_map<0>), grid, block, 0, s, L, w, hh, first, h->dX.p, reinterpret_cast<float2*>(h->dDense.p));
    else if (KS == 4) hipLaunchKernelGGL((k_warp_map<4>), grid, block, 0, s, L, w, hh, first, h->dX.p, reinterpret_cast<float2*>(h->dDense.p));
    else hipLaunchKernelGGL((k_warp_map<16>), grid, block, 0, s, L, w, hh, first, h->dX.p, reinterpret_cast<float2*>(h->dDense.p));
  }
  HIP_CHECK(hipGetLastError());
  if (kernelMs) HIP_CHECK(hipEventRecord(e1, s));
  if (out) HIP_CHECK(hipMemcpyAsync(out, h->dDense.p, bytes, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (kernelMs) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *kernelMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
}

// cornerMinEigenVal of n BGR float images (kind 0) / chamfer distance transform of n 8-bit masks (kind 1)
static void imageOps(cvd_handle* h, int kind, int n, int w, int hh, const void* in, float* out, double* kernelMs) {
  if (n < 0 || w < 1 || hh < 1) throw std::runtime_error("invalid image batch");
  if (n == 0) return;
  if (!in) throw std::runtime_error("null image input");
  hipStream_t s = h->stream;
  const size_t px = static_cast<size_t>(w) * hh, pixels = px * n;
  if (pixels > (1ull << 31)) throw std::runtime_error("image batch too large for one call");
  h->dImgOut.ensure(pixels);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernelMs) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); }
  if (kind == 0) {
    h->dImgIn.ensure(pixels * 3);
    h->dImgGray.ensure(pixels);
    h->dImgCov.ensure(pixels * 3);
    HIP_CHECK(hipMemcpyAsync(h->dImgIn.p, in, pixels * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    if (kernelMs) HIP_CHECK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_bgr_to_gray, dim3(static_cast<unsigned>((pixels + 255) / 256)), dim3(256), 0, s, h->dImgIn.p, pixels,
                       h->dImgGray.p);
    const dim3 grid(static_cast<unsigned>((px + 255) / 256), 1, n);
    hipLaunchKernelGGL(k_sobel_cov, grid, dim3(256), 0, s, h->dImgGray.p, w, hh, h->dImgCov.p);
    hipLaunchKernelGGL(k_box_min_eigenval, grid, dim3(256), 0, s, h->dImgCov.p, w, hh, h->dImgOut.p);
  } else {
    const size_t tmpPer = static_cast<size_t>(w + 4) * (hh + 4);
    h->dImgMask.ensure(pixels);
    h->dImgTmp.ensure(tmpPer * n);
    HIP_CHECK(hipMemcpyAsync(h->dImgMask.p, in, pixels, hipMemcpyHostToDevice, s));
    if (kernelMs) HIP_CHECK(hipEventRecord(e0, s));
    const size_t lds = kChamferThreads * sizeof(long long) + static_cast<size_t>(w) * sizeof(unsigned int);
    if (lds > 64 * 1024) throw std::runtime_error("mask too wide for the distance transform kernel");
    hipLaunchKernelGGL(k_chamfer_5x5, dim3(n), dim3(kChamferThreads), lds, s, h->dImgMask.p, w, hh, h->dImgTmp.p,
                       h->dImgOut.p);
  }
  HIP_CHECK(hipGetLastError());
  if (kernelMs) HIP_CHECK(hipEventRecord(e1, s));
  if (out) HIP_CHECK(hipMemcpyAsync(out, h->dImgOut.p, pixels * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (kernelMs) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *kernelMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
}

// Quaternion (x, y, z, w) times vector, the way Eigen evaluates it in float (uv = 2 q.vec x v; v + w uv + q.vec x uv)
static void quatRotate(const float* q, const float* v, float* out) {
  float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  for (int i = 0; i < 3; ++i) uv[i] += uv[i];
  const float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; ++i) out[i] = v[i] + q[3] * uv[i] + c[i];
}

static void flowGuidedFilter(cvd_handle* h, int n, int first, int count, int w, int hh, int dw, int dh, float invAspect,
                             const float* depth, const float* cameras, const float* flowF, const uint8_t* maskF,
                             const float* flowB, const uint8_t* maskB, int frameRadius, int spatialRadius, int median,
                             float* out, double* kernelMs) {
  if (n < 1 || first < 0 || count < 0 || first + count > n) throw std::runtime_error("invalid frame batch");
  if (w < 1 || hh < 1 || dw < 1 || dh < 1 || !(invAspect > 0.f)) throw std::runtime_error("invalid raster");
  if (frameRadius < 0 || spatialRadius < 0) throw std::runtime_error("negative filter radius");
  if (!depth || !cameras || (n > 1 && frameRadius > 0 && (!flowF || !maskF || !flowB || !maskB)))
    throw std::runtime_error("null filter input");
  if (count == 0) return;
  const long long side = 2ll * spatialRadius + 1, maxSamples = side * side * (2ll * frameRadius + 1);
  if (median && maxSamples > 256)
    throw std::runtime_error("flow guided median filter: (2 spatialRadius + 1)^2 (2 frameRadius + 1) > 256 samples per pixel");
  hipStream_t s = h->stream;
  const size_t px = static_cast<size_t>(w) * hh, dpx = static_cast<size_t>(dw) * dh;
  std::vector<FilterCam> cams(n);
  for (int k = 0; k < n; ++k) {
    const float* c = cameras + static_cast<size_t>(k) * 9;
    const float ex[3] = {1.f, 0.f, 0.f}, ey[3] = {0.f, 1.f, 0.f}, ez[3] = {0.f, 0.f, -1.f};
    for (int i = 0; i < 3; ++i) cams[k].pos[i] = c[i];
    quatRotate(c + 3, ex, cams[k].right);
    quatRotate(c + 3, ey, cams[k].up);
    quatRotate(c + 3, ez, cams[k].front);
    cams[k].tanH = std::tan(c[7] / 2.f);
    cams[k].tanV = std::tan(c[8] / 2.f);
  }
  h->dFltCams.upload(cams.data(), n, s);
  h->dFltDepth.upload(depth, dpx * n, s);
  const size_t links = n > 1 && frameRadius > 0 ? static_cast<size_t>(n - 1) : 0;
  h->dFltFlowF.upload(flowF, links * px * 2, s);
  h->dFltFlowB.upload(flowB, links * px * 2, s);
  h->dFltMaskF.upload(maskF, links * px, s);
  h->dFltMaskB.upload(maskB, links * px, s);
  h->dFltOut.ensure(px * count);
  FilterArgs A;
  A.n = n; A.first = first; A.count = count; A.w = w; A.h = hh; A.dw = dw; A.dh = dh; A.invAspect = invAspect;
  A.frameRadius = links ? frameRadius : 0; A.spatialRadius = spatialRadius; A.median = median;
  A.depth = h->dFltDepth.p; A.cams = h->dFltCams.p;
  A.flowFwd = reinterpret_cast<const float2*>(h->dFltFlowF.p); A.maskFwd = h->dFltMaskF.p;
  A.flowBwd = reinterpret_cast<const float2*>(h->dFltFlowB.p); A.maskBwd = h->dFltMaskB.p;
  A.out = h->dFltOut.p;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernelMs) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventRecord(e0, s)); }
  const dim3 grid(static_cast<unsigned>((px + 255) / 256), 1, count), block(256);
  if (!median) hipLaunchKernelGGL((k_flow_guided_filter<0>), grid, block, 0, s, A);
  else if (maxSamples <= 16) hipLaunchKernelGGL((k_flow_guided_filter<16>), grid, block, 0, s, A);
  else if (maxSamples <= 64) hipLaunchKernelGGL((k_flow_guided_filter<64>), grid, block, 0, s, A);
  else hipLaunchKernelGGL((k_flow_guided_filter<256>), grid, block, 0, s, A);
  HIP_CHECK(hipGetLastError());
  if (kernelMs) HIP_CHECK(hipEventRecord(e1, s));
  if (out) HIP_CHECK(hipMemcpyAsync(out, h->dFltOut.p, px * count * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (kernelMs) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *kernelMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
}

}  // namespace cvd

// =======================================================================================================
// C ABI
// =======================================================================================================
#define CVD_TRY(h, ...)                        \
  try {                                        \
    if (!(h)) return -1;                       \
    HIP_CHECK(hipSetDevice((h)->device));      \
    __VA_ARGS__;                               \
    return 0;                                  \
  } catch (const std::exception& e) {          \
    (h)->err = e.what();                       \
    return -1;                                 \
  }

extern "C" {

static std::string g_createError;

cvd_handle* cvd_create(int32_t device) {
  try {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      throw std::runtime_error("no HIP device available: the optimizer has no CPU path");
    if (device < 0 || device >= count) throw std::runtime_error("invalid device ordinal");
    HIP_CHECK(hipSetDevice(device));
    auto* h = new cvd_handle_t();
    h->device = device;
    HIP_CHECK(hipDeviceGetAttribute(&h->numCU, hipDeviceAttributeMultiprocessorCount, device));
    HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    // side stream of the asynchronous coarse rebuild (created here: the first use of a new stream costs ~10 ms)
    {
      // the side stream carries the coarse level's rebuild: long chains of SMALL kernels (rocSOLVER's panel factorisations
      // run on one workgroup) that must not queue behind the solver's device-filling launches -- highest priority
      int prioLow = 0, prioHigh = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
      static const bool flatPrio = std::getenv("CVD_SIDE_STREAM_FLAT") != nullptr;  // comparison knob
      HIP_CHECK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, flatPrio ? prioLow : prioHigh));
    }
    HIP_CHECK(hipEventCreateWithFlags(&h->evCoarseIn, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&h->evCoarseDone, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&h->evCoarseRead, hipEventDisableTiming));
    cvd_solver_options_default(&h->opt);
    return h;
  } catch (const std::exception& e) {
    g_createError = e.what();
    return nullptr;
  }
}
void cvd_destroy(cvd_handle* h) { delete h; }
const char* cvd_last_error(cvd_handle* h) { return h ? h->err.c_str() : g_createError.c_str(); }

void cvd_abi_sizes(int32_t* out6) {
  out6[0] = sizeof(cvd_xform_desc);
  out6[1] = sizeof(cvd_opt_params);
  out6[2] = sizeof(cvd_frame_pose);
  out6[3] = sizeof(cvd_iteration_record);
  out6[4] = sizeof(cvd_solve_summary);
  out6[5] = sizeof(cvd_solver_options);
}

void cvd_opt_params_default(cvd_opt_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->max_iterations = 1000;
  p->num_threads = 12;
  p->num_steps = 4;
  p->robustness = 0.5;
  p->static_loss_type = CVD_STATIC_REPRO_DISPARITY;
  p->static_spatial_weight = 1.0;
  p->static_depth_weight = 1.0;
  p->smooth_loss_type = CVD_SMOOTH_REPRO_DISPARITY_LAPLACIAN;
  p->scale_reg = 1.0;
  p->scale_reg_grid_size = 10;
  p->depth_deform_reg_initial = 1.0;
  p->depth_deform_reg_final = 0.1;
  p->spatial_deform_reg = 1.0;
  p->focal_reg = 1.0;
  p->coarse_to_fine = 1;
  p->ctf_long = 17;
  p->ctf_short = 10;
  p->dso_long = 4;
  p->dso_short = 3;
  p->focal_long = 0.3461538376301239;
  p->intr_opt = CVD_INTR_PER_FRAME;
  p->normalize_depth_from_first_frame = 1;
}

void cvd_solver_options_default(cvd_solver_options* o) {
  o->pcg_relative_tolerance = 5e-3;  // near-exact LM steps: what reproducing the reference's exact-step end state takes (cvd_hip.h)
  o->pcg_max_iterations = 300;
  o->pcg_check_every = 4;
  o->verbose = 0;
  o->force_iterations = 0;
  o->coarse_level = 1;
  o->robust_loss = 0;
}
int32_t cvd_set_solver_options(cvd_handle* h, const cvd_solver_options* o) { CVD_TRY(h, h->opt = *o); }
void cvd_comm_unique_id(uint8_t* out128) {
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  (void)ncclGetUniqueId(&id);
  std::memcpy(out128, &id, sizeof(id));
}
int32_t cvd_comm_init(cvd_handle* h, int32_t rank, int32_t world, const uint8_t* id128) {
  CVD_TRY(h, {
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("invalid rank / world size");
    if (h->comm) { NCCL_CHECK(ncclCommDestroy(h->comm)); h->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    NCCL_CHECK(ncclCommInitRank(&h->comm, world, id, rank));
    h->rank = rank;
    h->world = world;
    // test hook: with one rank the collectives are no-ops, but the sharded-mode kernels and call sequence still run
    h->distForced = world == 1 && std::getenv("CVD_FORCE_DIST") != nullptr;
    h->tableValid = false;
  });
}
#ifdef CVD_ASM_PROFILE
int32_t cvd_debug_asm_profile(unsigned long long* out) {
  hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cvd::g_asmProf), sizeof(unsigned long long) * 2048 * 16) == hipSuccess ? 0 : 1;
}
#endif
int32_t cvd_set_generic_kernels(cvd_handle* h, int32_t enabled) { CVD_TRY(h, h->forceGeneric = enabled != 0); }

int32_t cvd_set_video(cvd_handle* h, int32_t numFrames, int32_t width, int32_t height, float aspect, float invAspect) {
  CVD_TRY(h, {
    if (numFrames <= 0 || width <= 0 || height <= 0) throw std::runtime_error("invalid video dimensions");
    h->F = numFrames; h->W = width; h->H = height; h->aspect = aspect; h->invAspect = invAspect;
    h->haveDynMasks = false;
    h->adaptGx = h->adaptGy = 0;
    h->dDepth.ensure(static_cast<size_t>(numFrames) * width * height);
    HIP_CHECK(hipMemsetAsync(h->dDepth.p, 0, static_cast<size_t>(numFrames) * width * height * sizeof(float), h->stream));
    h->median.assign(numFrames, 0.f);
    h->medianDirty = true;
    h->poses.assign(numFrames, cvd_frame_pose{{0, 0, 0}, {0, 0, 0, 1}, 0.f, 0.f});
    h->poseParamsValid = false;
    cvd_xform_desc dd{};
    dd.type = CVD_XFORM_DEPTH;
    dd.depth_type = CVD_DEPTH_IDENTITY;
    cvd_xform_desc sd{};
    sd.type = CVD_XFORM_SPATIAL;
    sd.spatial_type = CVD_SPATIAL_IDENTITY;
    resetXforms(h, dd, false);
    resetXforms(h, sd, true);
    h->tableValid = false;
    h->P = 0;
    h->C = 0;
    h->dense = false;
    // everything keyed by frame index belongs to the previous video (stale triplet centres / pair graph
    // indexed past a smaller F)
    h->haveTriplets = false;
    h->tripCenter.clear();
    h->tripOff.clear();
    h->tripC = 0;
    h->tripActive.clear();
    h->haveGlobalEdges = false;
    h->globalEdges.clear();
    h->coarse.valid = false;
    h->pairA.clear();
    h->pairB.clear();
    h->pairOff.assign(1, 0);
  });
}

int32_t cvd_set_depth(cvd_handle* h, int32_t frame, const float* depth) {
  CVD_TRY(h, {
    if (frame < 0 || frame >= h->F) throw std::runtime_error("frame out of range");
    const size_t n = static_cast<size_t>(h->W) * h->H;
    HIP_CHECK(hipMemcpyAsync(h->dDepth.p + frame * n, depth, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->medianDirty = true;  // (the medians of the source depth are formed on the device before the next solve: refreshMedians)
    h->tableValid = false;
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

int32_t cvd_set_depth_all(cvd_handle* h, const float* depth) {
  CVD_TRY(h, {
    if (h->F <= 0) throw std::runtime_error("no video set");
    const size_t n = static_cast<size_t>(h->F) * h->W * h->H;
    HIP_CHECK(hipMemcpyAsync(h->dDepth.p, depth, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->medianDirty = true;
    h->tableValid = false;
    HIP_CHECK(hipStreamSynchronize(h->stream));
  });
}

int32_t cvd_set_pair_constraints(cvd_handle* h, int32_t numPairs, const int32_t* pairFrames, const int64_t* offsets,
                                 const float* loc4, const uint8_t* isStatic) {
  CVD_TRY(h, {
    // the reference iterates a std::map<std::pair<int,int>> (lib/FlowConstraints.h:149): sort by key
    std::vector<int> order(numPairs);
    for (int i = 0; i < numPairs; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (pairFrames[2 * a] != pairFrames[2 * b]) return pairFrames[2 * a] < pairFrames[2 * b];
      return pairFrames[2 * a + 1] < pairFrames[2 * b + 1];
    });
    if (numPairs < 0 || !offsets || (numPairs > 0 && !pairFrames)) throw std::runtime_error("invalid pair constraints");
    if (offsets[0] != 0) throw std::runtime_error("pair constraint offsets must start at 0");
    for (int i = 0; i < numPairs; ++i)
      if (offsets[i + 1] < offsets[i]) throw std::runtime_error("pair constraint offsets must be non-decreasing");
    for (int k = 1; k < numPairs; ++k)  // (order is sorted by key: duplicates are neighbours)
      if (pairFrames[2 * order[k]] == pairFrames[2 * order[k - 1]] && pairFrames[2 * order[k] + 1] == pairFrames[2 * order[k - 1] + 1])
        throw std::runtime_error("duplicate directed frame pair in the constraint list (merge the two lists: the reference keeps "
                                 "one entry per pair key, lib/FlowConstraints.h:149)");
    const long long C = offsets[numPairs];
    if (C > 0 && !loc4) throw std::runtime_error("invalid pair constraints");
    h->dense = false;
    h->dFlow.release();
    h->dFMask.release();
    h->P = numPairs;
    h->C = C;
    h->pairA.resize(numPairs);
    h->pairB.resize(numPairs);
    h->pairOff.assign(numPairs + 1, 0);
    std::vector<float> loc(static_cast<size_t>(C) * 4);
    std::vector<unsigned char> st(C, 1);
    std::vector<int> cpair(C);
    long long o = 0;
    for (int k = 0; k < numPairs; ++k) {
      const int src = order[k];
      const int a = pairFrames[2 * src], b = pairFrames[2 * src + 1];
      if (a < 0 || a >= h->F || b < 0 || b >= h->F) throw std::runtime_error("pair frame out of range");
      h->pairA[k] = a;
      h->pairB[k] = b;
      h->pairOff[k] = o;
      const long long n = offsets[src + 1] - offsets[src];
      std::memcpy(&loc[o * 4], loc4 + offsets[src] * 4, sizeof(float) * 4 * n);
      if (isStatic) std::memcpy(&st[o], isStatic + offsets[src], n);
      for (long long i = 0; i < n; ++i) cpair[o + i] = k;
      o += n;
    }
    h->pairOff[numPairs] = o;
    hipStream_t s = h->stream;
    h->dPairA.upload(h->pairA.data(), numPairs, s);
    h->dPairB.upload(h->pairB.data(), numPairs, s);
    h->dPairOff.upload(h->pairOff.data(), numPairs + 1, s);
    h->dLoc.upload(reinterpret_cast<const float4*>(loc.data()), C, s);
    h->dStatic.upload(st.data(), C, s);
    h->dCPair.upload(cpair.data(), C, s);
    HIP_CHECK(hipStreamSynchronize(s));
    h->tableValid = false;
  });
}

int32_t cvd_set_pair_flows(cvd_handle* h, int32_t numPairs, const int32_t* pairFrames, const float* flow, const uint8_t* mask) {
  CVD_TRY(h, {
    if (h->F <= 0) throw std::runtime_error("no video set");
    if (numPairs < 0 || (numPairs > 0 && (!pairFrames || !flow || !mask))) throw std::runtime_error("invalid pair flows");
    const long long npx = static_cast<long long>(h->W) * h->H;
    // the reference iterates a std::map<std::pair<int,int>> (lib/FlowConstraints.h:149): sort by key, reject duplicates
    std::vector<int> order(numPairs);
    for (int i = 0; i < numPairs; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (pairFrames[2 * a] != pairFrames[2 * b]) return pairFrames[2 * a] < pairFrames[2 * b];
      return pairFrames[2 * a + 1] < pairFrames[2 * b + 1];
    });
    h->pairA.resize(numPairs);
    h->pairB.resize(numPairs);
    h->pairOff.assign(numPairs + 1, 0);
    for (int k = 0; k < numPairs; ++k) {
      const int a = pairFrames[2 * order[k]], b = pairFrames[2 * order[k] + 1];
      if (a < 0 || a >= h->F || b < 0 || b >= h->F) throw std::runtime_error("pair frame out of range");
      if (k > 0 && h->pairA[k - 1] == a && h->pairB[k - 1] == b) throw std::runtime_error("duplicate frame pair");
      h->pairA[k] = a;
      h->pairB[k] = b;
      h->pairOff[k + 1] = static_cast<long long>(k + 1) * npx;
    }
    hipStream_t s = h->stream;
    h->dFlow.ensure(static_cast<size_t>(std::max(numPairs, 1)) * npx);
    h->dFMask.ensure(static_cast<size_t>(std::max(numPairs, 1)) * npx);
    for (int k = 0; k < numPairs; ++k) {  // pair-major in key order on the device
      const size_t src = static_cast<size_t>(order[k]) * npx, dst = static_cast<size_t>(k) * npx;
      HIP_CHECK(hipMemcpyAsync(h->dFlow.p + dst, reinterpret_cast<const float2*>(flow) + src, npx * sizeof(float2), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(h->dFMask.p + dst, mask + src, npx, hipMemcpyHostToDevice, s));
    }
    h->dPairA.upload(h->pairA.data(), numPairs, s);
    h->dPairB.upload(h->pairB.data(), numPairs, s);
    h->dPairOff.upload(h->pairOff.data(), numPairs + 1, s);
    HIP_CHECK(hipStreamSynchronize(s));
    h->P = numPairs;
    h->C = static_cast<long long>(numPairs) * npx;
    h->dense = true;
    h->tableValid = false;
  });
}

int32_t cvd_set_triplet_constraints(cvd_handle* h, int32_t numTriplets, const int32_t* centers, const int64_t* offsets,
                                    const float* loc6, const uint8_t* isStatic) {
  CVD_TRY(h, {
    const long long C = numTriplets > 0 ? offsets[numTriplets] : 0;
    h->tripCenter.assign(centers, centers + numTriplets);
    h->tripOff.assign(offsets, offsets + numTriplets + 1);
    h->tripC = C;
    for (int c : h->tripCenter)
      if (c < 1 || c + 1 >= h->F) throw std::runtime_error("triplet centre frame out of range");
    std::vector<int> groupOfC(static_cast<size_t>(std::max<long long>(C, 1)), 0);
    for (int g = 0; g < numTriplets; ++g)
      for (long long c = offsets[g]; c < offsets[g + 1]; ++c) groupOfC[c] = g;
    std::vector<unsigned char> st(static_cast<size_t>(std::max<long long>(C, 1)), 1);
    if (isStatic && C > 0) std::memcpy(st.data(), isStatic, C);
    hipStream_t s = h->stream;
    h->dTLoc.upload(loc6, static_cast<size_t>(C) * 6, s);
    h->dTStatic.upload(st.data(), st.size(), s);
    h->dTGroupOfC.upload(groupOfC.data(), groupOfC.size(), s);
    h->dTCenterAll.upload(h->tripCenter.data(), h->tripCenter.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    h->haveTriplets = true;
    h->tableValid = false;
  });
}

int32_t cvd_set_poses(cvd_handle* h, const cvd_frame_pose* poses) {
  CVD_TRY(h, { h->poses.assign(poses, poses + h->F); h->poseParamsValid = false; });
}
int32_t cvd_get_poses(cvd_handle* h, cvd_frame_pose* poses) {
  CVD_TRY(h, std::memcpy(poses, h->poses.data(), sizeof(cvd_frame_pose) * h->F));
}
int32_t cvd_reset_poses(cvd_handle* h, double focalLong) {
  CVD_TRY(h, {
    for (int f = 0; f < h->F; ++f) {
      cvd_frame_pose& p = h->poses[f];
      p.position[0] = p.position[1] = p.position[2] = 0.f;
      p.orientation[0] = p.orientation[1] = p.orientation[2] = 0.f;
      p.orientation[3] = 1.f;
      const float focal = static_cast<float>(focalLong);
      if (h->aspect >= 1.f) {
        p.hfov = std::atan(focal) * 2.f;
        p.vfov = std::atan(focal / h->aspect) * 2.f;
      } else {
        p.hfov = std::atan(focal * h->aspect) * 2.f;
        p.vfov = std::atan(focal) * 2.f;
      }
    }
    h->poseParamsValid = false;
  });
}
int32_t cvd_reset_depth_xforms(cvd_handle* h, const cvd_xform_desc* d) { CVD_TRY(h, resetXforms(h, *d, false)); }
int32_t cvd_reset_spatial_xforms(cvd_handle* h, const cvd_xform_desc* d) { CVD_TRY(h, resetXforms(h, *d, true)); }
int32_t cvd_grid_xform_split(cvd_handle* h, const cvd_xform_desc* d) { CVD_TRY(h, gridXformSplit(h, *d)); }
int32_t cvd_get_xform_desc(cvd_handle* h, int32_t spatial, cvd_xform_desc* d) {
  CVD_TRY(h, *d = spatial ? h->sdesc : h->ddesc);
}
int32_t cvd_num_xform_params(cvd_handle* h, int32_t spatial) {
  if (!h) return 0;
  try { return spatial ? h->nS() : h->nD(); } catch (...) { return 0; }
}
int32_t cvd_get_xform_params(cvd_handle* h, int32_t spatial, double* out) {
  CVD_TRY(h, {
    const auto& v = spatial ? h->sparams : h->dparams;
    if (!v.empty()) std::memcpy(out, v.data(), sizeof(double) * v.size());
  });
}
int32_t cvd_set_xform_params(cvd_handle* h, int32_t spatial, const double* in) {
  CVD_TRY(h, {
    auto& v = spatial ? h->sparams : h->dparams;
    if (!v.empty()) std::memcpy(v.data(), in, sizeof(double) * v.size());
  });
}
int32_t cvd_get_pose_params(cvd_handle* h, double* pose7) {
  CVD_TRY(h, {
    if (!h->poseParamsValid) posesToParams(h);
    for (int f = 0; f < h->F; ++f)
      for (int i = 0; i < 7; ++i) pose7[f * 7 + i] = h->poseParams[f][i];
  });
}
int32_t cvd_set_pose_params(cvd_handle* h, const double* pose7) {
  CVD_TRY(h, {
    h->poseParams.resize(h->F);
    for (int f = 0; f < h->F; ++f)
      for (int i = 0; i < 7; ++i) h->poseParams[f][i] = pose7[f * 7 + i];
    h->poseParamsValid = true;
  });
}
int32_t cvd_block_size(cvd_handle* h) {
  if (!h) return 0;
  try { return h->Bsz(); } catch (...) { return 0; }
}

int32_t cvd_normalize_depth(cvd_handle* h, const cvd_opt_params* p) { CVD_TRY(h, normalizeDepth(h, *p)); }
int32_t cvd_pose_optimization(cvd_handle* h, const cvd_opt_params* p) { CVD_TRY(h, poseOptimization(h, *p)); }
int32_t cvd_pose_optimization_step(cvd_handle* h, const cvd_opt_params* p, double depthDeformReg, int32_t convert) {
  CVD_TRY(h, {
    if (convert || !h->poseParamsValid) posesToParams(h);
    h->records.clear();
    poseOptimizationStep(h, *p, depthDeformReg);
  });
}
int32_t cvd_evaluate(cvd_handle* h, const cvd_opt_params* p, double depthDeformReg, const double* pose7, double* cost,
                     int32_t* nres, double* gradient, double* hdiag, double* hfull) {
  CVD_TRY(h, evaluate(h, *p, depthDeformReg, pose7, cost, nres, gradient, hdiag, hfull));
}
int32_t cvd_sample_pair_constraints(cvd_handle* h, int32_t numPairs, const int32_t* pairFrames, const float* corner,
                                    const float* flow, const uint8_t* mask, const float* dynDist, int32_t dynW,
                                    int32_t dynH, int32_t matchSeparation, float minDynamicDistance, int64_t* offsets) {
  CVD_TRY(h, sampleConstraints(h, false, numPairs, pairFrames, corner, flow, mask, nullptr, nullptr, dynDist, dynW, dynH,
                               matchSeparation, minDynamicDistance, offsets));
}
int32_t cvd_get_sampled_constraints(cvd_handle* h, float* loc4) {
  CVD_TRY(h, {
    const size_t n = h->sampledOff.empty() ? 0 : static_cast<size_t>(h->sampledOff.back());
    if (n) {
      HIP_CHECK(hipMemcpyAsync(loc4, h->dSampledLoc.p, n * 2 * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
      HIP_CHECK(hipStreamSynchronize(h->stream));
    }
  });
}
int32_t cvd_sample_triplet_constraints(cvd_handle* h, int32_t numTriplets, const int32_t* centers, const float* corner,
                                       const float* flow10, const uint8_t* mask10, const float* flow12,
                                       const uint8_t* mask12, const float* dynDist, int32_t dynW, int32_t dynH,
                                       int32_t matchSeparation, float minDynamicDistance, int64_t* offsets) {
  CVD_TRY(h, sampleConstraints(h, true, numTriplets, centers, corner, flow10, mask10, flow12, mask12, dynDist, dynW, dynH,
                               matchSeparation, minDynamicDistance, offsets));
}
int32_t cvd_get_sampled_triplet_constraints(cvd_handle* h, float* loc6) {
  CVD_TRY(h, {
    const size_t n = h->sampledTripOff.empty() ? 0 : static_cast<size_t>(h->sampledTripOff.back());
    if (n) {
      HIP_CHECK(hipMemcpyAsync(loc6, h->dSampledTrip.p, n * 3 * sizeof(float2), hipMemcpyDeviceToHost, h->stream));
      HIP_CHECK(hipStreamSynchronize(h->stream));
    }
  });
}
int32_t cvd_apply_depth_xforms(cvd_handle* h, int32_t firstFrame, int32_t numFrames, float* out, double* kernelMs) {
  CVD_TRY(h, denseMaps(h, 0, firstFrame, numFrames, 0, 0, out, kernelMs));
}
int32_t cvd_depth_param_maps(cvd_handle* h, int32_t firstFrame, int32_t numFrames, double* out, double* kernelMs) {
  CVD_TRY(h, denseMaps(h, 1, firstFrame, numFrames, 0, 0, out, kernelMs));
}
int32_t cvd_spatial_warp_maps(cvd_handle* h, int32_t firstFrame, int32_t numFrames, int32_t height, int32_t width,
                              float* out, double* kernelMs) {
  CVD_TRY(h, denseMaps(h, 2, firstFrame, numFrames, width, height, out, kernelMs));
}
int32_t cvd_set_dynamic_masks(cvd_handle* h, int32_t height, int32_t width, const uint8_t* masks) {
  CVD_TRY(h, {
    h->adaptGx = h->adaptGy = 0;
    if (!masks) { h->haveDynMasks = false; return 0; }
    if (h->F <= 0) throw std::runtime_error("no video set");
    if (width < 1 || height < 1) throw std::runtime_error("invalid mask size");
    h->dDynMask.upload(masks, static_cast<size_t>(h->F) * width * height, h->stream);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    h->dynW = width;
    h->dynH = height;
    h->haveDynMasks = true;
  });
}
int32_t cvd_corner_min_eigenval(cvd_handle* h, int32_t numImages, int32_t height, int32_t width, const float* bgr,
                                float* out, double* kernelMs) {
  CVD_TRY(h, imageOps(h, 0, numImages, width, height, bgr, out, kernelMs));
}
int32_t cvd_dynamic_distance(cvd_handle* h, int32_t numImages, int32_t height, int32_t width, const uint8_t* mask,
                             float* out, double* kernelMs) {
  CVD_TRY(h, imageOps(h, 1, numImages, width, height, mask, out, kernelMs));
}
int32_t cvd_flow_guided_filter(cvd_handle* h, int32_t numFrames, int32_t firstOutput, int32_t numOutputs, int32_t height,
                               int32_t width, int32_t depthHeight, int32_t depthWidth, float invAspect, const float* depth,
                               const float* cameras, const float* flowFwd, const uint8_t* maskFwd, const float* flowBwd,
                               const uint8_t* maskBwd, int32_t frameRadius, int32_t spatialRadius, int32_t median,
                               float* out, double* kernelMs) {
  CVD_TRY(h, flowGuidedFilter(h, numFrames, firstOutput, numOutputs, width, height, depthWidth, depthHeight, invAspect, depth,
                              cameras, flowFwd, maskFwd, flowBwd, maskBwd, frameRadius, spatialRadius, median, out, kernelMs));
}
int32_t cvd_get_summary(cvd_handle* h, cvd_solve_summary* s) { CVD_TRY(h, *s = h->summary); }
int32_t cvd_num_records(cvd_handle* h) { return h ? static_cast<int32_t>(h->records.size()) : 0; }
int32_t cvd_get_records(cvd_handle* h, cvd_iteration_record* out) {
  CVD_TRY(h, std::memcpy(out, h->records.data(), sizeof(cvd_iteration_record) * h->records.size()));
}
int32_t cvd_get_kernel_times(cvd_handle* h, double* avgMs6, int64_t* launches6) {
  CVD_TRY(h, {
    for (int k = 0; k < KC_COUNT; ++k) {
      avgMs6[k] = h->kcN[k] ? h->kcMs[k] / h->kcN[k] : 0.0;
      launches6[k] = h->kcN[k];
    }
  });
}
int32_t cvd_get_comm_times(cvd_handle* h, double* avgMs3, int64_t* counts3) {
  CVD_TRY(h, {
    for (int k = 0; k < 3; ++k) {
      avgMs3[k] = h->kcN[KC_COUNT + k] ? h->kcMs[KC_COUNT + k] / h->kcN[KC_COUNT + k] : 0.0;
      counts3[k] = h->kcN[KC_COUNT + k];
    }
  });
}
int32_t cvd_set_kernel_timing(cvd_handle* h, int32_t enabled) {
  CVD_TRY(h, {
    // 1 = all classes, otherwise a bit mask (bit k = class k); bits 8..15 = sampling stride - 1 of the event pairs
    // attached to the hot kernel's launches (0: every launch; 3: every 4th -- the start/stop events of
    // hipExtLaunchKernelGGL serialise the dispatch, ~3 % of the iteration rate when every launch carries them)
    h->timingStride = ((enabled >> 8) & 0xff) + 1;
    h->timingCounter = 0;
    enabled &= 0xff;
    h->timing = enabled == 1 ? 0x3f : enabled;
    for (int k = 0; k < KC_TOTAL; ++k) { h->kcMs[k] = 0.0; h->kcN[k] = 0; }
  });
}
int64_t cvd_num_active_constraints(cvd_handle* h) { return h ? h->numValid : 0; }

int32_t cvd_set_pair_graph(cvd_handle* h, int32_t numPairs, const int32_t* pairFrames) {
  CVD_TRY(h, {
    std::set<std::pair<int, int>> uniq;
    for (int i = 0; i < numPairs; ++i) {
      const int a = pairFrames[2 * i], b = pairFrames[2 * i + 1];
      if (a < 0 || a >= h->F || b < 0 || b >= h->F) throw std::runtime_error("pair graph frame out of range");
      if (a != b) uniq.insert({std::min(a, b), std::max(a, b)});
    }
    h->globalEdges.assign(uniq.begin(), uniq.end());
    h->haveGlobalEdges = numPairs > 0;
    h->tableValid = false;
  });
}

int32_t cvd_block_inverse_debug(cvd_handle* h, int32_t num_blocks, int32_t block_size, const double* a, int32_t variant,
                                float* inverse, int32_t* failed) {
  CVD_TRY(h, {
    if (num_blocks <= 0 || block_size <= 0 || block_size > kMaxFrameBlock) throw std::runtime_error("block_inverse_debug: bad sizes");
    if (variant < 0 || variant > 2) throw std::runtime_error("block_inverse_debug: variant must be 0, 1 or 2");
    const size_t n = static_cast<size_t>(num_blocks) * block_size * block_size;
    DevBuf<double> dA;
    DevBuf<double> dL;
    DevBuf<float> dM;
    DevBuf<int> dF;
    dA.ensure(n);
    dL.ensure(static_cast<size_t>(num_blocks) * block_size);
    dM.ensure(n);
    dF.ensure(1);
    hipStream_t s = h->stream;
    dA.upload(a, n, s);
    HIP_CHECK(hipMemsetAsync(dL.p, 0, static_cast<size_t>(num_blocks) * block_size * sizeof(double), s));
    HIP_CHECK(hipMemsetAsync(dM.p, 0, n * sizeof(float), s));
    HIP_CHECK(hipMemsetAsync(dF.p, 0, sizeof(int), s));
    Layout L{};
    L.F = num_blocks;
    L.B = block_size;
    launchBlockInverseRaw(h, L, dA.p, dL.p, dM.p, dF.p, variant);
    int fl = 0;
    dM.download(inverse, n, s);
    dF.download(&fl, 1, s);
    HIP_CHECK(hipStreamSynchronize(s));
    if (failed) *failed = fl;
  });
}

int32_t cvd_coarse_debug(cvd_handle* h, int32_t* num_unknowns, double* a_c, double* a_c_inverse, int32_t* failed) {
  CVD_TRY(h, {
    auto& C = h->coarse;
    if (!C.valid || !h->coarseOn) {
      *num_unknowns = 0;
    } else {
      const int F = h->F;
      const size_t n = static_cast<size_t>(F) * kCB;
      *num_unknowns = static_cast<int32_t>(n);
      hipStream_t s = h->stream;
      int fl = 0;
      C.fail.download(&fl, 1, s);
      if (a_c_inverse) {
        // A_c^-1 column by column through the very kernels the solver uses (rc = unit vector)
        std::vector<double> save(n), unit(n, 0.0);
        C.rc.download(save.data(), n, s);
        HIP_CHECK(hipStreamSynchronize(s));
        DevBuf<double> scalTmp;
        scalTmp.ensure(S_COUNT);
        HIP_CHECK(hipMemsetAsync(scalTmp.p, 0, S_COUNT * sizeof(double), s));
        for (size_t k = 0; k < n; ++k) {
          unit[k] = 1.0;
          C.rc.upload(unit.data(), n, s);
          if (C.denseMode) {
            hipLaunchKernelGGL(k_coarse_dense_apply, dim3(F), dim3(256), 0, s, F, C.denseInv.p, C.rc.p, C.c.p, C.modeActive.p,
                               C.dotPart.p, scalTmp.p, h->dCounters.p + 3, C.fail.p, 1, 0.0, static_cast<double*>(nullptr));
          } else {
            hipLaunchKernelGGL(k_coarse_apply_w, dim3(F), dim3(1024), 0, s, C.plan, C.Wb.p, C.rc.p, C.y.p, C.dotPart.p,
                               scalTmp.p, h->dCounters.p + 3, C.fail.p, 1, 0.0, static_cast<double*>(nullptr));
            hipLaunchKernelGGL(k_coarse_apply_wt, dim3(F), dim3(256), 0, s, coarseView(h, true, true), F, C.c.p,
                               scalTmp.p, 1);
          }
          C.c.download(a_c_inverse + k * n, n, s);
          HIP_CHECK(hipStreamSynchronize(s));
          unit[k] = 0.0;
        }
        C.rc.upload(save.data(), n, s);
        HIP_CHECK(hipStreamSynchronize(s));
      }
      std::vector<double> diag(static_cast<size_t>(F) * kCBB), edges(static_cast<size_t>(std::max(C.nEdges, 1)) * kCBB);
      std::vector<unsigned char> act(n);
      std::vector<int> efa(C.nEdges), efb(C.nEdges);
      C.diag.download(diag.data(), diag.size(), s);
      C.edges.download(edges.data(), static_cast<size_t>(C.nEdges) * kCBB, s);
      C.modeActive.download(act.data(), n, s);
      C.edgeFa.download(efa.data(), efa.size(), s);
      C.edgeFb.download(efb.data(), efb.size(), s);
      HIP_CHECK(hipStreamSynchronize(s));
      if (failed) *failed = fl;
      if (a_c) {
        std::fill(a_c, a_c + n * n, 0.0);
        for (int f = 0; f < F; ++f)
          for (int i = 0; i < kCB; ++i)
            for (int j = 0; j < kCB; ++j) a_c[(static_cast<size_t>(f) * kCB + i) * n + f * kCB + j] = diag[static_cast<size_t>(f) * kCBB + i * kCB + j];
        for (int e = 0; e < C.nEdges; ++e)
          for (int i = 0; i < kCB; ++i)
            for (int j = 0; j < kCB; ++j) {
              double v = edges[static_cast<size_t>(e) * kCBB + i * kCB + j];
              if (!act[efa[e] * kCB + i] || !act[efb[e] * kCB + j]) v = 0.0;
              a_c[(static_cast<size_t>(efa[e]) * kCB + i) * n + efb[e] * kCB + j] = v;
              a_c[(static_cast<size_t>(efb[e]) * kCB + j) * n + efa[e] * kCB + i] = v;
            }
      }
    }
  });
}

}  // extern "C"
