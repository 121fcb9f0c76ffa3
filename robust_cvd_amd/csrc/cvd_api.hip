// cvd_api.hip -- the C ABI of include/cvd_hip.h and the coarse-to-fine schedule above the solve.
#include "cvd_host.h"

namespace cvd {

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
    h->dQ.ensure(n + static_cast<size_t>(h->F) * kCB + 8); h->dHd.ensure(n); h->dMask.ensure(n);
    h->dH.ensure(n * Bmax); h->dMinv.ensure(n * Bmax + 4);
    // one undirected work item per ~768 constraints and pair: bounded by pairs + constraints / 768
    h->dQPart.ensure((static_cast<size_t>(h->P) + static_cast<size_t>(h->C / (h->dense ? kDenseChunk : kListChunk)) + 1 + 3 * static_cast<size_t>(h->numCU)) * 2 * Bmax);
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

namespace cvd {
static std::atomic<int> g_liveHandles[PersistentGate::kMaxDevices];
int liveHandles(int device) { return (device >= 0 && device < PersistentGate::kMaxDevices) ? g_liveHandles[device].load() : 0; }
PersistentGate::Slot& PersistentGate::slot(int device) {
  static Slot slots[kMaxDevices];
  if (device < 0 || device >= kMaxDevices) throw std::runtime_error("device ordinal out of range");
  return slots[device];
}
}  // namespace cvd

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
    // side stream of the asynchronous rebuild of the sparse coarse level (created here: the first use of a new stream
    // costs ~10 ms): a few small dependent kernels that must not queue behind the solver's device-filling launches
    {
      int prioLow = 0, prioHigh = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&prioLow, &prioHigh));
      HIP_CHECK(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prioHigh));
    }
    HIP_CHECK(hipEventCreateWithFlags(&h->evCoarseIn, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&h->evCoarseDone, hipEventDisableTiming));
    // ... and the stream of the frames' block inverses while the levels are built in line (a latency-bound sweep, one workgroup per
    // frame, that leaves most of every CU idle: cvd_solve.hip)
    HIP_CHECK(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&h->evInvIn, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&h->evInvDone, hipEventDisableTiming));
    cvd_solver_options_default(&h->opt);
    cvd_debug_options_default(&h->dbg);
    // device code of every translation unit now, not inside the first solve (first handle of the process: ~0.1 s)
    touchModule_setup(); touchModule_eval(); touchModule_matvec(); touchModule_precond(); touchModule_temporal(); touchModule_solve(); touchModule_frontend();
    if (device < PersistentGate::kMaxDevices) {
      //  A handle that is alone on its device launches k_pcg_tail without the gate's events (launchPcgTail holds the
      // slot's mutex while it checks the count and launches).  Becoming the second handle: count under the same mutex, then drain
      // the device -- every tail launch enqueued before this point has left it, every later one sees the count and is gated.
      PersistentGate::Slot& sl = PersistentGate::slot(device);
      std::lock_guard<std::mutex> lock(sl.m);
      if (++g_liveHandles[device] > 1) HIP_CHECK(hipDeviceSynchronize());
    }
    return h;
  } catch (const std::exception& e) {
    g_createError = e.what();
    return nullptr;
  }
}
void cvd_destroy(cvd_handle* h) {
  if (h && h->device >= 0 && h->device < PersistentGate::kMaxDevices) --g_liveHandles[h->device];
  delete h;
}
const char* cvd_last_error(cvd_handle* h) { return h ? h->err.c_str() : g_createError.c_str(); }

int32_t cvd_abi_revision(void) { return CVD_ABI_REVISION; }
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
  o->struct_size = CVD_STRUCT_STAMP(cvd_solver_options);
  o->pcg_relative_tolerance = 1e-3;  // near-exact LM steps: what reproducing the reference's exact-step end state takes (cvd_hip.h)
  o->pcg_max_iterations = 300;
  o->verbose = 0;
  o->coarse_level = 1;
  o->robust_loss = 0;
  o->dense_matrix_free = 0;
  o->block_inverse_variant = 0;
  o->coarse_dense_max_unknowns = 4096;
  o->coarse_rebuild_excess = 16;
  o->coarse_update_budget = 40000;
  o->coarse_dense_shift = 1e-5;
  o->constraint_order = 1;
  o->coarse_rebuild_excess_dense = 0;
  o->pcg_fused_tail = 1;
  o->coarse_dense_row_split = 5;
  o->dist_owner_update = 1;
  o->temporal_level = 1;
  o->temporal_step = 32;
  o->temporal_grid_x = 0;
  o->temporal_grid_y = 0;
  o->coarse_temporal_step = 8;
  o->coarse_over_budget = 0;
  o->coarse_temporal_min_frames = 128;
  o->temporal_weight = 0.7;
}
int32_t cvd_set_solver_options(cvd_handle* h, const cvd_solver_options* o) {
  CVD_TRY(h, {
    //  the struct is copied whole: refuse a caller built against another revision of the header, and values no
    // code path is defined for, before anything is stored
    if (o->struct_size != CVD_STRUCT_STAMP(cvd_solver_options))
      throw std::runtime_error(fmt("cvd_solver_options: struct_size %#llx != %#llx = sizeof | revision << 32 (caller built against "
                                   "another cvd_hip.h; start from cvd_solver_options_default)",
                                   static_cast<unsigned long long>(o->struct_size),
                                   static_cast<unsigned long long>(CVD_STRUCT_STAMP(cvd_solver_options))));
    if (o->pcg_fused_tail < 0 || o->pcg_fused_tail > 1) throw std::runtime_error("pcg_fused_tail in {0, 1}");
    if (!(o->pcg_relative_tolerance > 0.0 && o->pcg_relative_tolerance < 1.0)) throw std::runtime_error("pcg_relative_tolerance must lie in (0, 1)");
    if (o->pcg_max_iterations < 1) throw std::runtime_error("pcg_max_iterations must be >= 1");
    if (!(o->coarse_dense_shift >= 0.0 && o->coarse_dense_shift < 1.0)) throw std::runtime_error("coarse_dense_shift must lie in [0, 1)");
    if (o->coarse_rebuild_excess < 0 || o->coarse_rebuild_excess_dense < -1 || o->coarse_update_budget < 0)
      throw std::runtime_error("coarse_rebuild_excess / coarse_update_budget must be >= 0, coarse_rebuild_excess_dense >= -1");
    if (o->coarse_dense_max_unknowns < 0 || o->coarse_dense_max_unknowns > kDenseCoarseMaxUnknowns)
      throw std::runtime_error(fmt("coarse_dense_max_unknowns must lie in [0, %d] (what k_dense_spd_inverse holds in registers)",
                                   kDenseCoarseMaxUnknowns));
    if (o->coarse_level < 0 || o->coarse_level > 3 || o->coarse_temporal_step < 2 || o->robust_loss < 0 || o->robust_loss > 1 || o->block_inverse_variant < 0 ||
        o->block_inverse_variant > 2)
      throw std::runtime_error("coarse_level in {0, 1, 2, 3}, coarse_temporal_step >= 2, robust_loss in {0, 1}, block_inverse_variant in {0, 1, 2}");
    if (o->coarse_over_budget < 0 || o->coarse_over_budget > 1) throw std::runtime_error("coarse_over_budget in {0, 1}");
    if (o->coarse_temporal_min_frames < 0) throw std::runtime_error("coarse_temporal_min_frames must be >= 0");
    if (!(o->temporal_weight > 0.0 && o->temporal_weight <= 2.0)) throw std::runtime_error("temporal_weight must lie in (0, 2]");
    if (o->coarse_dense_row_split < 0 || o->coarse_dense_row_split > 8) throw std::runtime_error("coarse_dense_row_split must lie in [0, 8]");
    if (o->temporal_level < 0 || o->temporal_level > 2 || o->temporal_step < 2 || o->temporal_grid_x < 0 || o->temporal_grid_y < 0 ||
        o->temporal_grid_x == 1 || o->temporal_grid_y == 1)
      throw std::runtime_error("temporal_level in {0, 1, 2}, temporal_step >= 2, temporal_grid_x / _y 0 (automatic) or >= 2");
    if (o->coarse_update_budget != h->opt.coarse_update_budget || o->coarse_dense_max_unknowns != h->opt.coarse_dense_max_unknowns ||
        (o->coarse_level == 3) != (h->opt.coarse_level == 3) || o->coarse_temporal_step != h->opt.coarse_temporal_step ||
        o->coarse_over_budget != h->opt.coarse_over_budget || o->coarse_temporal_min_frames != h->opt.coarse_temporal_min_frames ||
        o->pcg_fused_tail != h->opt.pcg_fused_tail)
      h->tableValid = false;
    if (o->constraint_order != h->opt.constraint_order) h->orderGx = h->orderGy = -1;  // (the coarse level's variant is chosen when the table is compiled)
    h->opt = *o;
  });
}
void cvd_debug_options_default(cvd_debug_options* o) {
  o->struct_size = CVD_STRUCT_STAMP(cvd_debug_options);
  o->force_iterations = 0;
  o->force_sharded_path = 0;
  o->pcg_lockstep = 0;
  o->stall_fused_tail_once = 0;
}
int32_t cvd_set_debug_options(cvd_handle* h, const cvd_debug_options* o) {
  CVD_TRY(h, {
    if (o->struct_size != CVD_STRUCT_STAMP(cvd_debug_options))
      throw std::runtime_error("cvd_debug_options: struct_size does not match this library's cvd_hip_debug.h (start from cvd_debug_options_default)");
    h->dbg = *o;
    h->distForced = h->world == 1 && (h->comm != nullptr || h->localGroup || h->phantom) && o->force_sharded_path != 0;
  });
}
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
    h->localGroup.reset();
    // test hook: with one rank the collectives are no-ops, but the sharded-mode kernels and call sequence still run
    h->distForced = world == 1 && h->dbg.force_sharded_path != 0;
    h->tableValid = false;
  });
}
int32_t cvd_comm_init_local_group(cvd_handle* h, int32_t rank, int32_t world, uint64_t group_key) {
  CVD_TRY(h, {
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("invalid rank / world size");
    if (h->comm) { NCCL_CHECK(ncclCommDestroy(h->comm)); h->comm = nullptr; }
    h->localGroup = joinLocalGroup(group_key, world);
    h->rank = rank;
    h->world = world;
    h->distForced = world == 1 && h->dbg.force_sharded_path != 0;
    h->tableValid = false;
  });
}
int32_t cvd_comm_init_phantom(cvd_handle* h, int32_t rank, int32_t world) {
  CVD_TRY(h, {
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("invalid rank / world size");
    if (h->comm) { NCCL_CHECK(ncclCommDestroy(h->comm)); h->comm = nullptr; }
    if (h->localGroup) { leaveLocalGroup(*h->localGroup); h->localGroup.reset(); }
    h->phantom = true;
    h->rank = rank;
    h->world = world;
    h->distForced = world == 1;
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
    h->denseListValid = false;
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
    if (numPairs < 0 || !offsets || (numPairs > 0 && !pairFrames)) throw std::runtime_error("invalid pair constraints");
    // the reference iterates a std::map<std::pair<int,int>> (lib/FlowConstraints.h:149): sort by key
    std::vector<int> order(numPairs);
    for (int i = 0; i < numPairs; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (pairFrames[2 * a] != pairFrames[2 * b]) return pairFrames[2 * a] < pairFrames[2 * b];
      return pairFrames[2 * a + 1] < pairFrames[2 * b + 1];
    });
    if (offsets[0] != 0) throw std::runtime_error("pair constraint offsets must start at 0");
    for (int i = 0; i < numPairs; ++i)
      if (offsets[i + 1] < offsets[i]) throw std::runtime_error("pair constraint offsets must be non-decreasing");
    for (int k = 1; k < numPairs; ++k)  // (order is sorted by key: duplicates are neighbours)
      if (pairFrames[2 * order[k]] == pairFrames[2 * order[k - 1]] && pairFrames[2 * order[k] + 1] == pairFrames[2 * order[k - 1] + 1])
        throw std::runtime_error("duplicate directed frame pair in the constraint list (merge the two lists: the reference keeps "
                                 "one entry per pair key, lib/FlowConstraints.h:149)");
    const long long C = offsets[numPairs];
    if (C > 0 && !loc4) throw std::runtime_error("invalid pair constraints");
    for (int k = 0; k < numPairs; ++k)  // (everything is validated before the handle changes)
      if (pairFrames[2 * k] < 0 || pairFrames[2 * k] >= h->F || pairFrames[2 * k + 1] < 0 || pairFrames[2 * k + 1] >= h->F)
        throw std::runtime_error("pair frame out of range");
    h->dense = false;
    h->denseListValid = false;
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
    // (built aside and committed to the handle only after every check has passed: a throw leaves the old list intact)
    std::vector<int> newA(numPairs), newB(numPairs);
    std::vector<long long> newOff(numPairs + 1, 0);
    for (int k = 0; k < numPairs; ++k) {
      const int a = pairFrames[2 * order[k]], b = pairFrames[2 * order[k] + 1];
      if (a < 0 || a >= h->F || b < 0 || b >= h->F) throw std::runtime_error("pair frame out of range");
      if (k > 0 && newA[k - 1] == a && newB[k - 1] == b) throw std::runtime_error("duplicate frame pair");
      newA[k] = a;
      newB[k] = b;
      newOff[k + 1] = static_cast<long long>(k + 1) * npx;
    }
    h->tableValid = false;
    h->pairA.swap(newA);
    h->pairB.swap(newB);
    h->pairOff.swap(newOff);
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
    h->denseListValid = false;
    h->tableValid = false;
  });
}

int32_t cvd_dense_mode_supported(const cvd_opt_params* p, const cvd_xform_desc* depth, const cvd_xform_desc* spatial,
                                 int32_t have_triplets, int32_t world_size, int32_t problem) {
  if (!p || !depth || !spatial) return 0;
  return denseModeSupported(*p, *depth, *spatial, have_triplets != 0, world_size, problem == 1) ? 1 : 0;
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
int32_t cvd_get_dense_times(cvd_handle* h, double* avgMs2, int64_t* launches2) {
  CVD_TRY(h, {
    for (int k = 0; k < 2; ++k) {
      avgMs2[k] = h->kcN[KC_DENSE_WALK + k] ? h->kcMs[KC_DENSE_WALK + k] / h->kcN[KC_DENSE_WALK + k] : 0.0;
      launches2[k] = h->kcN[KC_DENSE_WALK + k];
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
    for (auto& k : h->timingCounterKc) k = 0;
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

int32_t cvd_dense_inverse_debug(cvd_handle* h, int32_t n, const double* a, double* inverse, int32_t* failed) {
  CVD_TRY(h, {
    if (n <= 0 || n > 8192) throw std::runtime_error("dense_inverse_debug: bad size");
    const size_t nn = static_cast<size_t>(n) * n;
    DevBuf<double> dA;
    DevBuf<double> dM;
    DevBuf<int> dF;
    dA.ensure(nn);
    dM.ensure(nn);
    dF.ensure(2);
    hipStream_t s = h->stream;
    dA.upload(a, nn, s);
    HIP_CHECK(hipMemsetAsync(dM.p, 0, nn * sizeof(double), s));
    HIP_CHECK(hipMemsetAsync(dF.p, 0, 2 * sizeof(int), s));
    launchDenseSpdInverse(h, n, dA.p, dM.p, dF.p, s, dF.p + 1);
    int fl[2] = {0, 0};
    dM.download(inverse, nn, s);
    dF.download(fl, 2, s);
    HIP_CHECK(hipStreamSynchronize(s));
    if (failed) *failed = fl[0];
  });
}

int32_t cvd_temporal_debug(cvd_handle* h, int32_t* dims6, double* a_t, double* a_t_inverse, double* lam, int32_t* failed) {
  CVD_TRY(h, {
    auto& T = h->temporal;
    hipStream_t s = h->stream;
    const bool on = T.on && T.built;
    const int dims[6] = {on ? T.NT : 0, T.S, T.nn, T.step, T.Sx, T.Sy};
    for (int i = 0; i < 6; ++i) dims6[i] = dims[i];
    if (on) {
      const size_t n2 = static_cast<size_t>(T.NT) * T.NT;
      if (a_t) T.A.download(a_t, n2, s);
      if (a_t_inverse) T.Ainv.download(a_t_inverse, n2, s);
      if (lam) h->dLam.download(lam, static_cast<size_t>(h->F) * h->Bsz(), s);
      int fl = 0;
      T.fail.download(&fl, 1, s);
      HIP_CHECK(hipStreamSynchronize(s));
      if (failed) *failed = fl;
    }
  });
}
int32_t cvd_path_info(cvd_handle* h, int32_t* out8) {
  CVD_TRY(h, {
    out8[0] = h->coarseOn ? 1 : 0;
    out8[1] = !h->coarseOn ? -1 : (h->coarse.temporalPose ? 2 : (h->coarse.denseMode ? 1 : 0));
    out8[2] = h->temporal.on ? 1 : 0;
    out8[3] = h->lastFusedTail ? 1 : 0;
    out8[4] = h->tailDisabled ? 1 : 0;
    out8[5] = h->lastKD;
    out8[6] = static_cast<int32_t>(h->itemFa.size());
    out8[7] = h->lastCross ? 1 : 0;
  });
}
int32_t cvd_coarse_debug(cvd_handle* h, int32_t* num_unknowns, double* a_c, double* a_c_inverse, int32_t* failed) {
  CVD_TRY(h, {
    auto& C = h->coarse;
    if (!C.valid || !h->coarseOn) {
      *num_unknowns = 0;
    } else if (C.temporalPose) {
      // temporal pose level (coarse_level 3): the node-reduced matrix (unknown = mode * nodes + node) and its inverse as stored
      const size_t n = static_cast<size_t>(C.ptN);
      *num_unknowns = static_cast<int32_t>(n);
      hipStream_t s = h->stream;
      int fl = 0;
      C.fail.download(&fl, 1, s);
      if (a_c) C.ptMat.download(a_c, n * n, s);
      if (a_c_inverse) C.ptInv.download(a_c_inverse, n * n, s);
      HIP_CHECK(hipStreamSynchronize(s));
      if (failed) *failed = fl;
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
