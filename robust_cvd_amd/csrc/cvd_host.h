// cvd_host.h -- host-side state shared by the translation units of libcvd_hip.so: error macros, device buffers, the
// handle (one DepthVideo + depth stream), the per-solve context and the functions the units call across.
// gfx950 only.  There is NO CPU path.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <array>
#include <atomic>
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
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/cvd_hip_debug.h"
#include "cvd_kernels.h"
#include "cvd_coarse.h"
#include "cvd_temporal.h"
#include "cvd_cross.h"
#include "cvd_dense_walk.h"
#include "cvd_triplets.h"
#include "cvd_dense.h"
#include "cvd_sampling.h"
#include "cvd_imageops.h"
#include "cvd_filter.h"


namespace cvd {

inline std::string fmt(const char* f, ...) {
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

inline double nowSeconds() {
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
inline int valueNumParams(int t) {
  if (t == CVD_VALUE_SCALE) return 1;
  if (t == CVD_VALUE_SCALE_SHIFT) return 2;
  throw std::runtime_error("Invalid value transform.");
}
inline int xformBlockSize(const cvd_xform_desc& d) {
  if (d.type == CVD_XFORM_DEPTH) return d.depth_type == CVD_DEPTH_IDENTITY ? 0 : valueNumParams(d.value_xform);
  return d.spatial_type == CVD_SPATIAL_IDENTITY ? 0 : 2;
}
inline int xformNumBlocks(const cvd_xform_desc& d) {
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

enum KernelClass { KC_ASSEMBLE = 0, KC_MATVEC_PAIRS, KC_MATVEC_FINISH, KC_CG_UPDATE, KC_INVERSE, KC_COST, KC_COUNT,
                   // exchange steps of the pair-sharded multi-GPU mode (cvd_get_comm_times): timed whenever any class is
                   KC_COMM_EVAL = KC_COUNT, KC_COMM_PRODUCT, KC_COMM_COARSE,
                   // kernels timed on their own beside their class (cvd_get_dense_times): the dense mode's pixel walk and grid x grid kernel
                   KC_DENSE_WALK, KC_DENSE_GG, KC_TOTAL };

struct Ceres {  // ceres::Solver::Options defaults used on this path
  static constexpr double initial_radius = 1e4, max_radius = 1e16, min_radius = 1e-32;
  static constexpr double min_relative_decrease = 1e-3;
  static constexpr double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  static constexpr int max_consecutive_invalid = 5;
};

enum ProblemKind { PK_POSE_STEP = 0, PK_NORMALIZE = 1 };

// exchange layer of the pair-sharded mode (cvd_comm.hip)
enum CommType { CT_F64 = 0, CT_F32, CT_I32, CT_U64 };
struct LocalGroup;  // test backend: the ranks are handles of one process on one device

}  // namespace cvd

namespace cvd {
std::shared_ptr<LocalGroup> joinLocalGroup(unsigned long long key, int world);
void leaveLocalGroup(LocalGroup& g);
}

using namespace cvd;

struct cvd_handle_t {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  cvd_solver_options opt{};
  cvd_debug_options dbg{};   // test / measurement hooks (cvd_hip_debug.h)

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
  DevBuf<float4> dLoc, dNdc, dNdcOrd;   // (Ord: the table re-ordered for the current depth grid, orderTable)
  DevBuf<float2> dDsrc, dDsrcOrd;
  int orderGx = -1, orderGy = -1;       // grid the ordered copy was made for (0: input order in force, -1: stale)
  bool tableOrdered = false;
  DevBuf<unsigned char> dStatic, dInRange, dRegOwner;
  // dense mode (cvd_set_pair_flows): flow / mask images of every directed pair instead of a constraint list
  bool dense = false;
  DevBuf<float2> dFlow;
  DevBuf<unsigned char> dFMask;

  // multi-GPU (pair-sharded): one RCCL communicator, this rank owns the regularisers of frames f % world == rank
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  std::shared_ptr<LocalGroup> localGroup;  // test backend of the exchange layer (cvd_comm_init_local_group)
  bool phantom = false;                    // measurement aid (cvd_comm_init_phantom): the other ranks do not exist
  DevBuf<unsigned char> dCommStage;
  DevBuf<const unsigned char*> dCommPtrs;
  bool distForced = false;  // test hook (cvd_solver_options::force_sharded_path with a 1-rank communicator): run the multi-rank code path
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
  hipStream_t stream2 = nullptr;                       // side stream of the asynchronous rebuild of the SPARSE coarse level
  rocblas_handle rbMain = nullptr;                     // main-stream handle (batched block inverses beyond B = 256)
  DevBuf<double> dInvScratch;
  DevBuf<int> dInvInfo;
  hipEvent_t evCoarseIn = nullptr, evCoarseDone = nullptr;
  hipStream_t stream3 = nullptr;                       // the frames' block inverses BESIDE an in-line build of the levels (cvd_solve.hip)
  hipEvent_t evInvIn = nullptr, evInvDone = nullptr;
  DevBuf<FrameConst> dFc2;                             // its own frame constants (the main stream rewrites dFc)
  DevBuf<long long> dItemRange;
  // explicit cross blocks of the dense mode (cvd_cross.h): undirected pairs, their rows, the blocks
  std::vector<int> xFa, xFb;
  DevBuf<int> dXFa, dXFb, dXSlot, dXFiOff, dXPairEdge, dXDiagSlot;
  bool xDiagRows = false;           // the product kernel writes H_ff p_f into a row of the frame's range (one GPU)
  int xRows = 0;                    // rows of the partial-product buffer in the explicit-block mode
  DevBuf<long long> dXRange;
  DevBuf<double> dXBlocks;
  // one-walk assembly of the dense mode (cvd_dense_walk.h): records of the directed pairs, per-pixel grid x grid scalars
  DevBuf<int> dDwPair, dDwRecOff, dXDir;
  DevBuf<double> dDwRecords, dDwGg;
  int nDwRecords = 0;
  // dense mode outside the fast scope: the list the images stand for, materialised on the device (denseListEnter / Leave)
  bool denseAsList = false;            // the handle currently runs a solve on the materialised list
  bool denseListValid = false;         // dLoc / dCPair / dStatic / denseListOff hold the list of the current images
  std::vector<long long> denseListOff; // pair offsets of the list
  DevBuf<int> dDlCounts;
  DevBuf<long long> dDlOffsets;
  DevBuf<unsigned int> dCounters;  // [0] k_matvec_finish, [1] k_cg_update (last-workgroup tickets)
  DevBuf<unsigned int> dTailBar;   // grid barrier of k_pcg_tail (tailArrive / tailWait)
  DevBuf<double> dOwnerScal;       // owner-sharded PCG iteration: {r^T z, r^T r} shares of every rank (all-gathered)
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
    // temporal pose level (coarse_level 3; cvd_temporal.h): denseMode is set as well -- same exchange layout, in-line build -- but
    // the matrix is the node-reduced one (ptN = 8 nodes unknowns) and the PCG kernels walk it with tlLevelRows
    bool temporalPose = false;
    int ptNn = 0, ptStepFrames = 0, ptN = 0, ptBlocks = 0;
    std::vector<int> edgeFaHost, edgeFbHost;
    DevBuf<int> ptA, ptB, ptPtr, ptList;
    DevBuf<double> ptMat, ptInv, ptR, ptT, ptDot, ptRec;
    DevBuf<TlStep> ptStepDev;   // [0]: per iteration (restricted products = Z^T q), [1]: first residual (= Z^T r)
    std::vector<unsigned char> ptStepHost;   // what ptStepDev holds (a solve re-uploads the two records only when they changed)
    DevBuf<unsigned int> ptCounter;
    // dense variant (cvd_coarse.h "DENSE coarse level"): A_c^-1 as a full f64 matrix, built in line by k_dense_spd_inverse
    bool denseMode = false;
    int denseForB = 0;        // frame-block size of the problem the last build was for (a coarse-to-fine level)
    bool denseReady = false;  // a build for this plan has been launched: denseValid says whether denseInv holds an inverse
    DevBuf<double> denseA, densePanel;
    DevBuf<double> denseInv;
    DevBuf<int> denseValid;
    int nW = 0;
    DevBuf<unsigned char> modeActive;
    DevBuf<int> fail;
    DevBuf<unsigned int> barrier;  // grid barrier words of k_coarse_factor_mw / k_dense_spd_inverse
    bool ptInvPending = false;     // the temporal pose level's inverse waits for the depth-grid level's (one launch for both)
    int* ptInvFail = nullptr;
    // second set of the factor's outputs: a rebuild runs on a side stream while the PCG of the same LM iteration
    // still uses the previous factor (launchCoarseSetup / the LM loop)
    DevBuf<double> Wb2;
    DevBuf<int> fail2;
    CoarsePlan plan{};
  } coarse;
  bool coarseOn = false;  // this solve uses the coarse level
  // third level (cvd_temporal.h): temporal hats x coarse hats on the depth grid
  struct TemporalHost {
    bool on = false;        // this solve uses the level
    bool built = false;     // A_T^-1 of this solve exists
    int S = 0, Sx = 0, Sy = 0, nn = 0, step = 0, NT = 0, width = 0, nGroups = 0, nBlocks = 0;
    std::vector<unsigned char> stepHost;               // what stepDev holds
    int tabGx = 0, tabGy = 0, tabSx = 0, tabSy = 0;   // what the spatial tables were built for
    int grpF = 0, grpStep = 0;                        // ... the groups / block lists
    std::vector<int> grpFa, grpFb;
    DevBuf<float> hx, hy, elW;
    DevBuf<int> bx, by, gOff, gItems, blkA, blkB, gPtr, gather, fail, valid;
    DevBuf<float4> vW;
    DevBuf<unsigned int> vIdx, counter;
    DevBuf<unsigned char> elV;
    DevBuf<double> Cf, E, part, A, Ainv, sq, rT, t, tl, dotPart, rec;
    // panel / grid-barrier words of ITS dense inverse: the level's inverse runs on the solver's stream while the pose-graph level may
    // be rebuilding on the side stream (k_coarse_factor_mw zeroes and uses the pose-graph level's barrier words)
    DevBuf<double> invPanel;
    DevBuf<unsigned int> invBarrier;
    DevBuf<TlStep> stepDev;  // the kernels read the level's descriptor from memory (see matvecFinishBody)
    hipEvent_t evIn = nullptr, evDone = nullptr;  // fork / join of the assembly on the side stream
    bool sidePending = false;                     // ... forked and not yet joined (a solve that throws in between joins on its way out)
    double* sqPtr = nullptr; // where the frames' restricted products live: sq, or (pair-sharded, fused exchange) behind [q | Z^T q | p.q]
  } temporal;
  // measured on this handle (cvd_solve.hip: denseRebuildThreshold): an in-line rebuild of the dense coarse level and a PCG iteration
  hipEvent_t evRebuild[2] = {nullptr, nullptr};
  bool rebuildTimed = false;
  double coarseRebuildMs = 0.0, pcgIterMs = 0.0;
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

  bool lastFusedTail = false, lastCross = false;  // cvd_path_info: what the last PCG solve ran
  int lastKD = 0;
  bool tailDisabled = false;  // k_pcg_tail abandoned its grid barrier once on this handle: two-launch tail from then on (runPcg)
  bool forceGeneric = false;  // test hook: route the products through the generic (all-variants) kernel

  // kernel timing
  int timing = 0;  // bit mask of KernelClass values to time with HIP events
  int timingStride = 1;        // hipExtLaunchKernelGGL event pairs (tReserve) on every timingStride-th launch only
  long long timingCounter = 0;
  long long timingCounterKc[16] = {0};  // per class: tBegin samples every timingStride-th launch of a class as well
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evPool;
  std::vector<int> evClass;
  std::vector<int> evIter;  // PCG iteration the launch belongs to (-1 outside PCG): launches enqueued past
  int curPcgIter = -1;      // convergence are no-ops and are dropped from the statistics (tDropFrom)
  size_t evUsed = 0;
  double kcMs[KC_TOTAL] = {0};
  long long kcN[KC_TOTAL] = {0};

  ~cvd_handle_t() {
    for (auto& e : evPool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (rbMain) (void)rocblas_destroy_handle(rbMain);
    if (comm) (void)ncclCommDestroy(comm);
    if (localGroup) leaveLocalGroup(*localGroup);
    if (hScal) (void)hipHostFree(hScal);
    for (auto& p : hStage) if (p) (void)hipHostFree(p);
    if (hPcg) (void)hipHostFree(hPcg);
    for (auto& e : pcgEvent) if (e) (void)hipEventDestroy(e);
    for (auto& e : evRebuild) if (e) (void)hipEventDestroy(e);
    if (temporal.evIn) (void)hipEventDestroy(temporal.evIn);
    if (temporal.evDone) (void)hipEventDestroy(temporal.evDone);
    if (evCoarseIn) (void)hipEventDestroy(evCoarseIn);
    if (evCoarseDone) (void)hipEventDestroy(evCoarseDone);
    if (evInvIn) (void)hipEventDestroy(evInvIn);
    if (evInvDone) (void)hipEventDestroy(evInvDone);
    if (stream3) (void)hipStreamDestroy(stream3);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream) (void)hipStreamDestroy(stream);
  }

  int nD() const { return xformNumBlocks(ddesc) * xformBlockSize(ddesc); }
  int nS() const { return xformNumBlocks(sdesc) * xformBlockSize(sdesc); }
  int Bsz() const { return 7 + nD() + nS(); }

  // ---- timing helpers --------------------------------------------------------------------------------
  int tBegin(int kc) {
    if (kc >= KC_COUNT ? timing == 0 : !(timing & (1 << kc))) return -1;
    // (uniform sample of the class' launches; the dense mode's walks, a handful of long launches per solve, are all timed)
    if (timingStride > 1 && kc < KC_DENSE_WALK && (timingCounterKc[kc]++ % timingStride) != 0) return -1;
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

// The per-frame kernels of the solve (k_matvec_finish, k_cg_update, the block inverse, the fast pairs product) hold one
// frame block per workgroup with at most 256 unknowns.  Checked BEFORE any work or state mutation (a coarse-to-fine
// schedule would otherwise fail at its last level with the transforms already refined).
constexpr long long kListChunk = 768;    // constraints per direction and work item (list mode)
constexpr long long kDenseChunk = 8192;  // pixel slots per direction and work item (dense mode)

constexpr int kMaxFrameBlock = 512;
// k_dense_spd_inverse holds the lower triangle in registers, <= 25 tiles per wave of a 14 x 14 super-tile and <= 22 x 23 / 2
// super-tiles on 256 CUs: 22 * 14 = 308 tiles of 16 (launchDenseSpdInverse checks the actual device)
constexpr int kDenseCoarseMaxUnknowns = 4928;
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

template <typename K>
void allowLds(K kernel, size_t bytes) {
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

// Per-device gate for kernels with a grid barrier: constructed right before the launch (the stream first waits for the
// previous gated kernel of ANY handle of this process), destroyed right after it (records the completion the next one waits
// for).  The event belongs to the process, not to a handle.
int liveHandles(int device);  // handles of this process on the device (cvd_api.hip)
struct PersistentGate {
  static constexpr int kMaxDevices = 64;
  struct Slot { std::mutex m; hipEvent_t ev = nullptr; bool recorded = false; };
  static Slot& slot(int device);  // (cvd_api.hip: one table for all translation units)
  Slot& sl;
  hipStream_t s;
  PersistentGate(int device, hipStream_t stream) : sl(slot(device)), s(stream) {
    sl.m.lock();
    try {
      if (!sl.ev) HIP_CHECK(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
      if (sl.recorded) HIP_CHECK(hipStreamWaitEvent(s, sl.ev, 0));
    } catch (...) {
      sl.m.unlock();
      throw;
    }
  }
  ~PersistentGate() {
    sl.recorded = hipEventRecord(sl.ev, s) == hipSuccess;
    sl.m.unlock();
  }
};

// ---- functions shared between the translation units (cvd_setup.hip, cvd_eval.hip, cvd_matvec.hip, cvd_precond.hip,
// cvd_solve.hip, cvd_frontend.hip, cvd_api.hip) ---------------------------------------------------------------------
// in-place sum / sum to the owner's chunk `rank` / concatenation in rank order; send may alias recv as RCCL's in-place forms do
void commAllReduce(cvd_handle* h, void* buf, size_t count, CommType t, hipStream_t s);
void commReduceScatter(cvd_handle* h, const void* send, void* recv, size_t recvCount, CommType t, hipStream_t s);
void commAllGather(cvd_handle* h, const void* send, void* recv, size_t sendCount, CommType t, hipStream_t s);
void commGroupStart(cvd_handle* h);
void commGroupEnd(cvd_handle* h);
std::vector<int> rangeOf(const cvd_opt_params& p, int F);
void posesToParams(cvd_handle* h);
void paramsToPoses(cvd_handle* h, const cvd_opt_params& params);
void resetXforms(cvd_handle* h, const cvd_xform_desc& d, bool spatial);
void gridXformSplit(cvd_handle* h, const cvd_xform_desc& nd);
Layout makeLayout(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, ProblemKind kind);
void checkFrameBlock(size_t B, const char* what);
void tapCounts(const Layout& L, int& KD, int& KS);
bool fastLoss(const Layout& L);
Table makeTable(cvd_handle* h);
void checkDenseScope(cvd_handle* h, const Layout& L, int KS, bool trip);
bool denseFastScope(const cvd_handle* h, const Layout& L, int KS, bool trip);
bool denseFastBlockFits(long long B);   // (the frame's packed triangle in LDS: B <= 199)
// RAII: a solve whose configuration lies outside the dense fast scope runs on the device-materialised list (cvd_dense_walk.h)
struct DenseListScope {
  cvd_handle* h;
  bool entered = false;
  DenseListScope(cvd_handle* h, bool needList);
  ~DenseListScope();
};
bool denseModeSupported(const cvd_opt_params& p, const cvd_xform_desc& dd, const cvd_xform_desc& sd, bool haveTriplets, int world,
                        bool normalize);
AsmPanels makePanels(int B, size_t capDoubles, int& panelCap);
void compileTable(cvd_handle* h, const std::vector<int>& range, bool withTriplets = false, bool ignoreStatic = false);
void orderTable(cvd_handle* h, const Layout& L, int KD);  // after compileTable: the table order for this problem's depth grid
double* pinnedStage(cvd_handle* h, int which, size_t n);
void uploadState(cvd_handle* h, const Layout& L, DevBuf<double>& dst);
void downloadState(cvd_handle* h, const Layout& L, const DevBuf<double>& src);
void buildMask(cvd_handle* h, const Layout& L, const cvd_opt_params& p, ProblemKind kind, const std::vector<int>& range);
void ensureBuffers(Ctx& c);
void refreshMedians(cvd_handle* h);
void launchFrameConsts(Ctx& c, const double* x);
void spinStream(hipStream_t s);
void spinEvent(hipEvent_t ev);
void readScalars(Ctx& c);
void enqueueCost(Ctx& c, const double* x);
double evalCost(Ctx& c, const double* x);
void enqueueStats(Ctx& c);
bool crossScope(cvd_handle* h, const Ctx& c);
CrossPairs crossPairs(cvd_handle* h);
void launchCrossAssemble(Ctx& c, const double* x);
double evalFull(Ctx& c, const double* x, bool withStats = false, bool noReadBack = false);
bool coarseFusedConsumers();
bool fusedExchange(cvd_handle* h, bool withCoarse);
size_t exchangeOffsetQc(const Ctx& c);
size_t exchangeOffsetPq(const Ctx& c, bool withDenseCoarse);
size_t exchangeTemporalCount(cvd_handle* h, bool withCoarse);  // doubles the third level adds to the fused exchange (behind p.q)
bool ownerShardedUpdate(cvd_handle* h, bool withCoarse);
inline int denseRowSplit(const cvd_handle* h) { return std::min(kCB, std::max(0, h->opt.coarse_dense_row_split)); }
void launchDenseSpdInverse(cvd_handle* h, int n, const double* A, double* out, int* fail, hipStream_t s, int* outValid,
                           DevBuf<double>* panelBuf = nullptr, DevBuf<unsigned int>* barrierBuf = nullptr);
struct DinvRequest {
  int n = 0;
  const double* A = nullptr;
  double* out = nullptr;
  int* fail = nullptr;
  int* outValid = nullptr;
  DevBuf<double>* panelBuf = nullptr;
  DevBuf<unsigned int>* barrierBuf = nullptr;
};
void launchDenseSpdInversePair(cvd_handle* h, const DinvRequest& a, const DinvRequest& b, hipStream_t s);
CoarseView coarseView(cvd_handle* h, bool on, bool walk);
void prepareMatvec(Ctx& c, const double* x);
// tailFused: only the partial products are launched; the caller follows with launchPcgTail (finish + update in one launch)
void launchMatvec(Ctx& c, const double* x, const double* z, const double* pOld, double* pNew, int useBeta, const double* lam,
                  double* q, bool withCoarse = false, bool tailFused = false);
bool pcgTailScope(Ctx& c, bool coarse, int nThreads, size_t& lds, int& ldsFinish, int& ldsScratch);
void launchPcgTail(Ctx& c, const double* x, const double* pOld, double* pNew, int useBeta, const double* lam, double* q,
                   bool withCoarse, int nThreads, size_t lds, int ldsFinish, int ldsScratch, double tol2);
void launchBlockInverseRaw(cvd_handle* h, const Layout& L, const double* dH, const double* dLam, float* dMinv, int* dFail, int variant,
                           hipStream_t onStream = nullptr);
void launchBlockInverse(Ctx& c, hipStream_t onStream = nullptr);   // (onStream: one GPU, B <= 256 only)
void launchCoarseSetup(Ctx& c, const double* x, int side = 0);
// third level (cvd_temporal.hip)
bool temporalScope(const Ctx& c);
bool temporalPrepare(Ctx& c);                       // tables and work lists of this solve's problem (false: outside the kernels' limits)
void launchTemporalSetup(Ctx& c, const double* x, int half);  // A_T for the current (H, lam, x): 0 = its assembly (beside the
                                                              // pose-graph level's build), 1 = its inverse
void launchTemporalInit(Ctx& c, bool closeScalars, double tol2);  // first residual of a PCG solve: t, r_T, tl, the level's part of r^T z
TlStep temporalStep(cvd_handle* h);                 // (Ainv == nullptr when the level is off)
const TlStep* temporalStepDev(cvd_handle* h);       // its device copy for the kernels, nullptr when the level is off
// temporal pose level (coarse_level 3)
void poseTemporalPlan(cvd_handle* h);                                   // node-pair lists for the compiled table's edge graph
void poseTemporalPrepare(Ctx& c);                                       // buffers + descriptors of this solve
void launchPoseTemporalBuild(Ctx& c, hipStream_t s, int* failOut, bool deferInverse = false);      // from the coarse level's diag / edge blocks: matrix + inverse
void launchPoseTemporalInit(Ctx& c, double tol2);                       // first residual: t, c, closes the PCG scalars
const TlStep* poseTemporalStepDev(cvd_handle* h);                       // nullptr unless this solve uses the level
void coarseDebug(cvd_handle* h, int32_t* num_unknowns, double* a_c, double* a_c_inverse, int32_t* failed);
int runPcg(Ctx& c, const double* x, const std::function<void()>& tail = nullptr);
bool wantsTriplets(const cvd_opt_params& p, ProblemKind kind);
void bindTriplets(Ctx& c, const cvd_opt_params& p, ProblemKind kind);
void solve(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, ProblemKind kind);
void evaluate(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, const double* pose7, double* cost, int32_t* nres,
              double* gradient, double* hdiag, double* hfull);
void sampleConstraints(cvd_handle* h, bool triplet, int num, const int32_t* keyFrames, const float* corner, const float* flow,
                       const uint8_t* mask, const float* flow2, const uint8_t* mask2, const float* dyn, int dw, int dh,
                       int matchSeparation, float minDynamicDistance, int64_t* offsets);
void denseMaps(cvd_handle* h, int kind, int first, int count, int w, int hh, void* out, double* kernelMs);
void imageOps(cvd_handle* h, int kind, int n, int w, int hh, const void* in, float* out, double* kernelMs);
void touchModule_setup();
void touchModule_eval();
void touchModule_matvec();
void touchModule_precond();
void touchModule_solve();
void touchModule_frontend();
void touchModule_temporal();
void flowGuidedFilter(cvd_handle* h, int n, int first, int count, int w, int hh, int dw, int dh, float invAspect, const float* depth,
                      const float* cameras, const float* flowF, const uint8_t* maskF, const float* flowB, const uint8_t* maskB,
                      int frameRadius, int spatialRadius, int median, float* out, double* kernelMs);

}  // namespace cvd
