/* include/cvd_hip_debug.h -- TEST and MEASUREMENT hooks of libcvd_hip.so.
 *
 * Nothing here is part of the drop-in boundary (include/cvd_hip.h): these entry points exist for tests/ (parity hooks, simulated
 * ranks on one GPU, forced code paths) and tools/ (profiling aids).  A caller of the library never needs them; the reference has
 * no counterpart for any of them.  They live in the same shared object so that the tests exercise the product binary.
 */
#ifndef CVD_HIP_DEBUG_H
#define CVD_HIP_DEBUG_H

#include "cvd_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Hooks that change what a solve does for the sake of a test or a measurement (per handle; all default 0 = the product path).
 * Until round 5 these sat in cvd_solver_options. */
typedef struct cvd_debug_options {
  uint64_t struct_size;          /* CVD_STRUCT_STAMP(cvd_debug_options): cvd_debug_options_default sets, cvd_set_debug_options checks */
  int32_t force_iterations;      /* measurement: ignore the convergence tests, run exactly max_iterations */
  int32_t force_sharded_path;    /* 1: a 1-rank communicator runs the multi-rank code path (owner chunks, exchange calls) */
  int32_t pcg_lockstep;          /* profiling: 1 = the host never enqueues a PCG iteration ahead of the convergence flag, so
                                    that per-launch counter averages contain no early-exit launches */
  int32_t stall_fused_tail_once; /* 1: the host treats the third PCG iteration of the handle's first solve as a stalled grid barrier
                                    of k_pcg_tail (exercises the fall-back to the two-launch tail; until round 5: pcg_fused_tail = 2) */
} cvd_debug_options;
void cvd_debug_options_default(cvd_debug_options* o);
int32_t cvd_set_debug_options(cvd_handle* h, const cvd_debug_options* o);

/* Test hook: 1 = run the generic all-variants kernels even where a specialised fast kernel exists. */
int32_t cvd_set_generic_kernels(cvd_handle* h, int32_t enabled);

/* Test backend of the exchange layer: the `world` ranks are handles of THIS process on ONE device, each driven by its
 * own host thread; handles that pass the same `group_key` form one group.  RCCL refuses two ranks on one device, so
 * this is how the multi-rank code paths run with world > 1 on a single-GPU box (tests/test_gpu_two_ranks.py).
 * Host-synchronous; never used by a multi-GPU run. */
int32_t cvd_comm_init_local_group(cvd_handle* h, int32_t rank, int32_t world, uint64_t group_key);
/* Measurement aid (tools/shard_sim.py): this handle becomes rank `rank` of a `world`-rank run whose OTHER ranks do not exist --
 * every collective returns at once and the other ranks' contributions are simply missing.  The sharded code path runs with the
 * real owner chunks, offsets and launch geometry of that rank, so its kernels can be timed on one GPU; the numbers the solve
 * produces mean nothing. */
int32_t cvd_comm_init_phantom(cvd_handle* h, int32_t rank, int32_t world);

/* Parity hook for the per-frame dense solve of the block-Jacobi preconditioner (no reference counterpart: Ceres'
 * SPARSE_NORMAL_CHOLESKY, lib/PoseOptimizer.cpp:956, is replaced by PCG): inverts `num_blocks` symmetric positive
 * definite block_size x block_size f64 matrices `a` (row-major, block after block) with the very kernel the solver uses
 * and returns the f32 inverses.  variant 0: blocked sweep on the f64 matrix cores (the default path), 1: scalar
 * register-tile sweep, 2: LDS Cholesky.  failed = number of non-positive pivots met. */
int32_t cvd_block_inverse_debug(cvd_handle* h, int32_t num_blocks, int32_t block_size, const double* a, int32_t variant,
                                float* inverse, int32_t* failed);

/* Test hook for the coarse level of the preconditioner (state of the last LM iteration of the last solve):
 * n = 8 * frames (0 when the level was off), a_c = Z^T (J^T J + diag(lam)) Z as a dense n x n matrix assembled
 * from its blocks, a_c_inverse = the inverse the solver applied, failed = pivot failures of the factorisation.
 * Any output pointer may be NULL. */
/* Test hook: the dense SPD inverse of the dense coarse level (cvd_dense_inverse.h) on one n x n f64 matrix (row-major,
 * symmetric); inverse = n x n f64; failed = 1 on a non-positive pivot (inverse untouched), bit 30 = barrier timeout. */
int32_t cvd_dense_inverse_debug(cvd_handle* h, int32_t n, const double* a, double* inverse, int32_t* failed);
int32_t cvd_coarse_debug(cvd_handle* h, int32_t* num_unknowns, double* a_c, double* a_c_inverse, int32_t* failed);
/* Test hook for the third level of the preconditioner (cvd_solver_options::temporal_level; state of its last build in the last
 * solve): dims6 = {NT unknowns (0: the level was off), S hats per node, nn nodes, step, Sx, Sy}; a_t = the assembled Galerkin
 * matrix (NT x NT, unknown s * nn + a, diagonal shifted by coarse_dense_shift), a_t_inverse = the inverse in use, lam = the LM
 * damping vector (frames x block) of the last LM iteration.  Any output pointer but dims6 may be NULL. */
int32_t cvd_temporal_debug(cvd_handle* h, int32_t* dims6, double* a_t, double* a_t_inverse, double* lam, int32_t* failed);

#ifdef __cplusplus
}
#endif
#endif
