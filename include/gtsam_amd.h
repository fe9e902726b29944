/* gtsam_amd.h -- C ABI of the MI355X-native Levenberg-Marquardt inner loop.
 *
 * This is the drop-in boundary for ONE path of borglab/gtsam: the
 * linearize -> damp -> eliminate -> solve -> retract -> error loop of
 * gtsam::LevenbergMarquardtOptimizer.  GTSAM has no FFI/plugin ABI of its own; its extension
 * point is C++ subclassing (NonlinearOptimizer::iterate() nonlinear/NonlinearOptimizer.h:136,
 * LevenbergMarquardtOptimizer::linearize() LevenbergMarquardtOptimizer.h:112-113,
 * NonlinearOptimizer::solve() NonlinearOptimizer.h:129-130).  The functions below are what a
 * subclass overriding iterate() binds to (see INTEGRATION.md for the GTSAM-side shim and
 * gtsam_amd/host/ for the shipped one).  Each entry point cites the reference code it replaces.
 *
 * Conventions: extern "C"; plain pointers + explicit sizes; all host arrays are caller-owned and
 * copied during the call; int status returns (GTG_OK=0, GTG_INDETERMINATE=1 means "the damped
 * system was not positive definite" = where the reference throws
 * IndeterminantLinearSystemException, linear/HessianFactor.cpp:476-483, and LM raises lambda,
 * nonlinear/LevenbergMarquardtOptimizer.cpp:154-160; negative = usage / HIP error, message via
 * gtg_last_error()).  No exceptions cross the boundary, no GTSAM or torch types appear.
 * All floating point is IEEE double (gtsam::Matrix = Eigen::MatrixXd, base/Matrix.h:39).
 */
#ifndef GTSAM_AMD_H
#define GTSAM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gtg_context* gtg_handle;

enum { GTG_OK = 0, GTG_INDETERMINATE = 1, GTG_ERR_USAGE = -1, GTG_ERR_HIP = -2,
       GTG_ERR_UNSUPPORTED = -3 };

/* Variable (gtsam::Value) types and their packed storage.
 *   POSE3      gtsam::Pose3 (geometry/Pose3.h): 12 doubles = R row-major (9) then t (3); tangent 6 = [omega; v]
 *   SFM_CAMERA gtsam::PinholeCamera<Cal3Bundler> (geometry/PinholeCamera.h, Cal3Bundler.h):
 *              17 doubles = pose (12) then f,k1,k2,u0,v0; tangent 9 = [pose(6); f,k1,k2]
 *   POINT3     gtsam::Point3: 3 doubles; tangent 3
 *   POSE2      gtsam::Pose2 (geometry/Pose2.h): 3 doubles = x, y, theta (theta() = atan2(s, c), wrapped); tangent 3,
 *              retract = compose with Pose2(v0, v1, v2) (Pose2.cpp:99-109, GTSAM_SLOW_BUT_CORRECT_EXPMAP off)    */
enum { GTG_VAR_POSE3 = 0, GTG_VAR_SFM_CAMERA = 1, GTG_VAR_POINT3 = 2, GTG_VAR_POSE2 = 3 };

/* Factor types on the path. */
enum { GTG_FAC_GENERAL_SFM = 0,   /* GeneralSFMFactor<PinholeCamera<Cal3Bundler>,Point3>  slam/GeneralSFMFactor.h:127-177 */
       GTG_FAC_PROJECTION = 1,    /* GenericProjectionFactor<Pose3,Point3,Cal3_S2>        slam/ProjectionFactor.h:138-166 */
       GTG_FAC_BETWEEN_POSE3 = 2, /* BetweenFactor<Pose3>, or BetweenFactor<Pose2> when its  slam/BetweenFactor.h:111-124
                                     two variables are POSE2 (same table, see between_z)                                  */
       GTG_FAC_PRIOR = 3 };       /* PriorFactor<T>                                        nonlinear/PriorFactor.h:98-102 */

/* Noise models (linear/NoiseModel.cpp).  whiten(v) is
 *   UNIT v ; ISOTROPIC v*invsigma (:641-663) ; DIAGONAL v.*invsigmas (:311-325) ; GAUSSIAN R*v (:163-181)
 * noise_data holds the constructor arguments: UNIT nothing; ISOTROPIC {sigma}; DIAGONAL sigmas[dim];
 * GAUSSIAN R[dim*dim] row-major (upper-triangular sqrt information).  The library derives
 * invsigma = 1.0/sigma exactly like the reference constructors do (NoiseModel.cpp:275-281). */
enum { GTG_NOISE_UNIT = 0, GTG_NOISE_ISOTROPIC = 1, GTG_NOISE_DIAGONAL = 2, GTG_NOISE_GAUSSIAN = 3 };

/* Optional m-estimator wrapped around a noise-table entry = noiseModel::Robust(robust, noise)
 * (linear/NoiseModel.h:678-740, linear/LossFunctions.cpp), Block re-weighting scheme (the default):
 * whitened system scaled by sqrt(weight(||whitened b||)) (LossFunctions.cpp:51-124), factor error =
 * loss(||whitened r||) (NoiseModel.h:716-718).  Parameter = the estimator's modelParameter() (c or k).
 * As in the reference, GTG_FAC_GENERAL_SFM factors are NOT re-weighted when they linearize (their own linearize()
 * whitens H1, H2, b one at a time through Robust::Whiten(Matrix), slam/GeneralSFMFactor.h:162-168, where the
 * weight evaluates to 1); the m-estimator only enters their error(). */
enum { GTG_ROBUST_NONE = 0, GTG_ROBUST_FAIR = 1, GTG_ROBUST_HUBER = 2, GTG_ROBUST_CAUCHY = 3, GTG_ROBUST_TUKEY = 4,
       GTG_ROBUST_WELSCH = 5, GTG_ROBUST_GEMANMCCLURE = 6, GTG_ROBUST_DCS = 7, GTG_ROBUST_L2WITHDEADZONE = 8 };
/* (AsymmetricTukey / AsymmetricCauchy, LossFunctions.cpp:433-500: on this path the distance is a norm, >= 0, where they coincide with
 * Tukey / Cauchy -- the C++ shim maps them.) */

/* The factor graph in structure-of-arrays form: what the extractor produces by walking a
 * gtsam::NonlinearFactorGraph once (FactorGraph.h:92 factors_, Factor.h keys_). Variable ids are
 * dense 0..n_vars-1 (the shim maps gtsam::Key -> id in Values order, Values.h:74-79). */
typedef struct gtg_problem {
  int32_t n_vars;
  const int32_t* var_type;       /* [n_vars] GTG_VAR_* */

  int32_t n_noise;               /* shared noise-model table */
  const int32_t* noise_kind;     /* [n_noise] GTG_NOISE_* */
  const int32_t* noise_dim;      /* [n_noise] */
  const int64_t* noise_off;      /* [n_noise] offset into noise_data */
  const double* noise_data;
  const int32_t* noise_robust;   /* [n_noise] GTG_ROBUST_* or NULL (no robust models) */
  const double* noise_robust_param; /* [n_noise] c / k of the m-estimator */

  int64_t n_sfm;                 /* GTG_FAC_GENERAL_SFM */
  const int32_t* sfm_cam;        /* [n_sfm] variable id of the SFM_CAMERA */
  const int32_t* sfm_point;      /* [n_sfm] variable id of the POINT3 */
  const double* sfm_z;           /* [n_sfm*2] measured pixel */
  const int32_t* sfm_noise;      /* [n_sfm] index into noise table (dim 2) */

  int64_t n_proj;                /* GTG_FAC_PROJECTION */
  const int32_t* proj_pose;      /* [n_proj] variable id of the POSE3 */
  const int32_t* proj_point;     /* [n_proj] variable id of the POINT3 */
  const double* proj_z;          /* [n_proj*2] */
  const int32_t* proj_noise;     /* [n_proj] */
  const int32_t* proj_calib;     /* [n_proj] index into calib table */
  const int32_t* proj_sensor;    /* [n_proj] index into sensor table or -1 (body_P_sensor_ absent) */
  int32_t n_calib;
  const double* calib;           /* [n_calib*5] Cal3_S2 fx,fy,s,u0,v0 (geometry/Cal3_S2.cpp:44-50); distortion: calib_distortion below */
  int32_t n_sensor;
  const double* sensor;          /* [n_sensor*12] body_P_sensor poses */

  int64_t n_between;             /* GTG_FAC_BETWEEN_POSE3 */
  const int32_t* between_v1;     /* [n_between] */
  const int32_t* between_v2;     /* [n_between] */
  const double* between_z;       /* [n_between*12] measured relative pose (Pose3 packing; for a POSE2 pair x, y, theta
                                    in the first 3 doubles of the factor's 12) */
  const int32_t* between_noise;  /* [n_between] (dim 6, or 3 for a POSE2 pair) */

  int64_t n_prior;               /* GTG_FAC_PRIOR */
  const int32_t* prior_var;      /* [n_prior] */
  const int64_t* prior_off;      /* [n_prior] offset into prior_data (storage size of the var type) */
  const double* prior_data;
  const int32_t* prior_noise;    /* [n_prior] (dim = tangent dim of the variable) */

  const double* calib_distortion; /* [n_calib*4] k1, k2, p1, p2 of a Cal3DS2 calibration (radial + tangential distortion,
                                     geometry/Cal3DS2_Base.cpp:93-132: GenericProjectionFactor<Pose3, Point3, Cal3DS2>), or NULL:
                                     every entry of the calib table is a plain Cal3_S2.  (Appended in round 2: a caller built
                                     against the older struct must be recompiled.) */

  /* SmartProjectionFactor<PinholeCamera<Cal3Bundler>> (slam/SmartProjectionFactor.h, the factor of timing/timeSFMBALsmart.cpp):
   * one factor = one track; its landmark is NOT a variable but re-triangulated from the cameras at every linearisation / error
   * evaluation (DLT, geometry/triangulation.cpp:27-57, checks of triangulateSafe triangulation.h:697-752, cached while no camera
   * pose moves by more than retriangulationThreshold, SmartProjectionFactor.h:127-183) and eliminated without damping (the Schur
   * complement of the point, :190-233; JacobianFactorQ / JacobianFactorSVD are the same normal equations).  NULL / 0: no smart factors. */
  int64_t n_smart;
  const int64_t* smart_ptr;       /* [n_smart+1] offsets of a factor's measurements in smart_cam / smart_z (>= 1 measurement each) */
  const int32_t* smart_cam;       /* variable id (SFM_CAMERA) of every measurement */
  const double* smart_z;          /* 2 per measurement */
  const int32_t* smart_noise;     /* [n_smart] index into the noise table (dim 2, Unit or Isotropic as the reference requires) */
  const double* smart_params;     /* [n_smart*8] rankTolerance, landmarkDistanceThreshold, dynamicOutlierRejectionThreshold,
                                     retriangulationThreshold, degeneracyMode (0 IGNORE_DEGENERACY, 1 ZERO_ON_DEGENERACY,
                                     2 HANDLE_INFINITY), linearizationMode (0 HESSIAN, 2 JACOBIAN_Q, 3 JACOBIAN_SVD; 1 =
                                     IMPLICIT_SCHUR is refused: the reference's direct solvers cannot eliminate it either),
                                     enableEPI (0 / 1: the DLT point refined by the reference's LM on TriangulationFactors,
                                     triangulation.h:211-221, 531-534), 1 reserved.  A track that does not triangulate is what SmartProjectionFactor.h makes of it:
                                     nothing (ZERO_ON_DEGENERACY; the Jacobian modes when linearising), a point at infinity
                                     (:356-371 when linearising in HESSIAN mode, :419-427 in the error under HANDLE_INFINITY), error
                                     0.0 otherwise.  useLOST is not supported.  Where the reference THROWS out of
                                     linearize() / error() -- the point at infinity behind one of the track's cameras
                                     or the enableEPI refinement linearising behind a camera (CheiralityException),
                                     Cal3Bundler::calibrate not converging -- the call returns an error */
} gtg_problem;

/* ---- lifetime -------------------------------------------------------------------------------- */
int gtg_create(gtg_handle* out, int device_id);
int gtg_destroy(gtg_handle h);
/* Optional: do the one-time work of the process on `device_id` now (HIP runtime start, code-object load, function objects of every
 * kernel, the factorisation's masked streams) instead of inside the first gtg_create / gtg_upload_problem / gtg_try_lambda.  Idempotent
 * and thread-safe: meant to run on a helper thread while the caller loads or extracts its problem. */
int gtg_prewarm(int device_id);
const char* gtg_last_error(void);
const char* gtg_version(void);

/* One-time symbolic analysis + upload.  Replaces what the reference redoes on EVERY lambda try:
 * VariableIndex (inference/VariableIndex-inl.h:27-49), EliminationTree
 * (EliminationTree-inst.h:77-155), JunctionTree (JunctionTree-inst.h:63-151), Scatter
 * (linear/Scatter.cpp:39-73).  Landmarks (POINT3 touched only by projection factors / priors) are
 * eliminated first -- the Schur ordering of timing/timeSFMBAL.h:74-83 -- the remaining variables
 * form the reduced system.  shard/n_shards: this handle owns landmark-factors with
 * (landmark rank) % n_shards == shard and other factors with (factor index) % n_shards == shard
 * (multi-GPU); pass 0,1 for a single GPU. */
int gtg_upload_problem(gtg_handle h, const gtg_problem* p, int shard, int n_shards);

/* Optional: position of each non-landmark variable in the reduced system (a fill-reducing
 * ordering, inference/Ordering.cpp:42-124).  order[i] = variable id placed i-th; n = number of
 * non-landmark variables.  Must be called before gtg_upload_problem.  Default: a reverse
 * Cuthill-McKee ordering of the reduced graph computed at upload (band-limits the tile fill). */
int gtg_set_reduced_ordering(gtg_handle h, const int32_t* order, int32_t n);

/* Values in packed storage, variable id order (see GTG_VAR_*).  Values.h:74-79. */
int64_t gtg_values_size(gtg_handle h);   /* doubles in the packed Values (the caller's variables: the landmarks of smart factors are internal) */
int64_t gtg_tangent_size(gtg_handle h);  /* dimension of delta (VectorValues) */
int gtg_set_values(gtg_handle h, const double* packed, int64_t n);
int gtg_get_values(gtg_handle h, double* packed, int64_t n);        /* current (accepted) values */
int gtg_get_trial_values(gtg_handle h, double* packed, int64_t n);  /* values.retract(delta) of the last try */

/* NonlinearFactorGraph::error(values) = sum_i 0.5*||whiten(r_i)||^2
 * (nonlinear/NonlinearFactorGraph.cpp:170-179, NonlinearFactor.cpp:136-147) at the current values. */
int gtg_error(gtg_handle h, double* error);

/* NonlinearFactorGraph::linearize(values) (NonlinearFactorGraph.cpp:239-278) fused with the
 * lambda-invariant part of elimination: J^T J / J^T b block accumulation
 * (JacobianFactor::updateHessian linear/JacobianFactor.cpp:563-598) and
 * GaussianFactorGraph::hessianDiagonal (GaussianFactorGraph.cpp:279-287). */
int gtg_linearize(gtg_handle h);

/* One body of LevenbergMarquardtOptimizer::tryLambda (LevenbergMarquardtOptimizer.cpp:121-270):
 * buildDampedSystem (internal/LevenbergMarquardtState.h:125-156) -> solve
 * (GaussianFactorGraph::optimize, linear/GaussianFactorGraph.cpp:316-319: landmark elimination +
 * Cholesky of the reduced system + back-substitution) -> linear.error(0), linear.error(delta)
 * (GaussianFactorGraph.cpp:71-78) -> values.retract(delta) (nonlinear/Values.cpp:52-63) ->
 * graph.error(newValues).  Returns GTG_OK or GTG_INDETERMINATE; on GTG_OK fills
 *   out[0] = linear.error(0)   out[1] = linear.error(delta)   out[2] = graph.error(trial values)
 *   out[3] = ||delta||_2
 * The trial values stay on the device until gtg_accept(). If linear cost change < 0 the retract /
 * error step is skipped exactly like LM.cpp:178 and out[2] = +inf.
 * Errors (negative return, text in gtg_last_error): a dependency wait of the factorisation ran into its bound (GPU shared or
 * preempted: "the step was not computed" -- never reported as GTG_INDETERMINATE); a smart factor met one of the two cases in
 * which the reference throws (gtg_problem.smart_params; also from gtg_linearize and gtg_error). */
int gtg_try_lambda(gtg_handle h, double lambda, int diagonal_damping, double min_diagonal,
                   double max_diagonal, double out[4]);

/* Same contract as gtg_try_lambda, but the damped reduced system is solved by the reference's preconditioned
 * conjugate gradients (linear/ConjugateGradientSolver.h:106-169; NonlinearOptimizerParams::Iterative with
 * PCGSolverParameters, NonlinearOptimizer.cpp:154-172) with a block-Jacobi preconditioner, applied to the IMPLICIT Schur
 * complement of the landmarks (never formed).  cg = {maxIterations, minIterations, epsilon_rel, epsilon_abs}
 * (ConjugateGradientParameters defaults: 500, 1, 1e-3, 1e-3); *iterations receives the CG iteration count.
 * Single-shard graphs only. */
int gtg_try_lambda_pcg(gtg_handle h, double lambda, int diagonal_damping, double min_diagonal, double max_diagonal,
                       const double cg[4], double out[4], int32_t* iterations);

/* state_ = decreaseLambda(... newValues ...) (LevenbergMarquardtState.h:81-94): trial -> current. */
int gtg_accept(gtg_handle h);

/* Replicated handles -- one per GPU, each holding the WHOLE graph (n_shards = 1), driven in lock step by one process per GPU: the
 * device address of the packed values (which = 0: the current values, 1: the trial values of the last gtg_try_lambda), their
 * length in doubles and the stream the handle's kernels run on, for a device-to-device exchange run by the caller on that stream
 * (e.g. ncclBroadcast of the accepted trial values from the replica whose lambda was taken into every other replica's current
 * values).  After writing a handle's current values in place, gtg_values_changed() tells it so (what gtg_set_values does after
 * its copy): the linearisation and the trial values are no longer valid.  Used by the speculative lambda search of
 * gtsam_amd/speculative.py: replica r tries the r-th lambda of the sequence LevenbergMarquardtOptimizer::tryLambda would walk
 * through on consecutive rejections (LevenbergMarquardtOptimizer.cpp:121-270, LevenbergMarquardtState.h:70-76), the decisions are
 * replayed in order from the gathered results: the trajectory is the sequential one, an iteration costs one try. */
int gtg_values_device_ptr(gtg_handle h, int which, void** ptr, int64_t* n_doubles, void** stream);
int gtg_values_changed(gtg_handle h);

/* ---- parity / debug getters (host copies) --------------------------------------------------- */
int gtg_get_delta(gtg_handle h, double* delta, int64_t n);            /* VectorValues of last solve, variable id order */
int gtg_get_gradient(gtg_handle h, double* g, int64_t n);             /* J^T b, variable id order */
int gtg_get_hessian_diagonal(gtg_handle h, double* d, int64_t n);     /* GaussianFactorGraph::hessianDiagonal */
/* whitened Jacobian blocks + rhs of factor type `factor_type` after gtg_linearize():
 * row-major per factor: SFM [A1 2x9 | A2 2x3 | b 2], PROJECTION [2x6 | 2x3 | 2],
 * BETWEEN [6x6 | 6x6 | 6], PRIOR [d x d | d] padded to d=9 (81+9). n = doubles available in out. */
int gtg_get_jacobians(gtg_handle h, int factor_type, double* out, int64_t n);
int64_t gtg_reduced_dim(gtg_handle h);
/* dense damped reduced system of the last try: S (n x n row-major, lower triangle valid; after the
 * solve it holds the Cholesky factor L) */
int gtg_get_reduced_matrix(gtg_handle h, double* S, int64_t n_elems);

/* ---- multi-GPU: the one exchange step ------------------------------ */
/* Called by the library with a DEVICE pointer whenever per-shard partial sums must be combined
 * (sum) across shards: reduced Hessian+gradient+scalars.  The callback runs the collective
 * (RCCL all-reduce over xGMI when driven from torch.distributed) and must complete (or be stream
 * ordered on `stream`) before returning. */
typedef int (*gtg_allreduce_fn)(void* device_ptr, int64_t n_doubles, void* stream, void* user);
int gtg_set_allreduce(gtg_handle h, gtg_allreduce_fn fn, void* user);

/* ---- measurement ---------------------------------------------------------------------------- */
/* HIP-event timings (ms, accumulated since the last reset) per phase; names via gtg_phase_name. */
enum { GTG_PH_LINEARIZE = 0, GTG_PH_ASSEMBLE, GTG_PH_POINT_ELIM, GTG_PH_SCHUR, GTG_PH_CHOLESKY,
       GTG_PH_SOLVE, GTG_PH_LINEAR_ERROR, GTG_PH_RETRACT, GTG_PH_ERROR, GTG_PH_COUNT };
int gtg_enable_timing(gtg_handle h, int on);
int gtg_get_phase_ms(gtg_handle h, double* ms, int64_t* calls, int n);
int gtg_reset_timing(gtg_handle h);
const char* gtg_phase_name(int phase);
/* flops of the dense Cholesky factorisation of the last try (n^3/3 + lower order) and the
 * algorithmic HBM bytes of one linearize pass (DESIGN.md section "rooflines") */
double gtg_cholesky_flops(gtg_handle h);
/* the same factorisation counted at the granularity of the variables: sum over the d x d block columns (symbolic fill included)
 * of f^3/3 + f^2 s + f s^2 -- the flops a perfectly sparse elimination would need; gtg_cholesky_flops() counts what the 128x128
 * tile kernels execute (zeros inside stored tiles included) */
double gtg_cholesky_flops_block_level(gtg_handle h);
/* What the dataflow factorisation executes: gtg_cholesky_flops() minus the contraction products it skips because one of their 16 x 16
 * operand sub-tiles is structurally zero (strip-level symbolic factorisation; round 6).  Equal to gtg_cholesky_flops() for the stream
 * schedule and for dense plans. */
double gtg_cholesky_flops_executed(gtg_handle h);
/* identity (63-bit hash) of the layout of the reduced system: elimination order of the reduced variables, their offsets,
 * padding, tile structure and the list of exchanged tiles.  It is derived from the WHOLE graph, so every shard of one job
 * reports the same value; gtg_upload_problem verifies that through the all-reduce callback and fails if they differ. */
int64_t gtg_structure_hash(gtg_handle h);
double gtg_linearize_bytes(gtg_handle h);

/* standalone kernels exposed for unit parity tests (tests/ only): dense FP64 Cholesky of an
 * n x n row-major SPD matrix (lower triangle) held in host memory; returns GTG_INDETERMINATE on a
 * non-positive pivot (Eigen LLT info, base/cholesky.cpp:124-127). */
int gtg_dense_cholesky_host(gtg_handle h, double* A, int32_t n, double* rhs_inout /* may be NULL */);

/* tests only: the tile schedule of the reduced-system Cholesky exactly as the kernels read it (index lists, no numbers), so
 * that the symbolic phase (fill, update lists, elimination-tree parts, backward-solve lists) can be executed and checked
 * on the CPU (tests/test_chol_plan.py).  sizes = {nt, |rows|, |pairs| (int32 entries), |bcols|, n_stored, n_exch, n_pairs
 * of columns, n_parts}; per_tile[nt][4] = trsm_off, trsm_cnt, bwd_off, bwd_cnt; per_pair[np][8] = s1, nar, rest, anc (off, cnt). */
int gtg_debug_plan_sizes(gtg_handle h, int64_t sizes[8]);
int gtg_debug_plan_lists(gtg_handle h, int32_t* rows, int32_t* pairs, int32_t* bcols, int32_t* stored, int32_t* exch,
                         int64_t* per_tile, int64_t* per_pair, int32_t* pair_part, int32_t* part_parent);
/* The dataflow schedule of the same factorisation (csrc/chol_dataflow.hip, the default): sizes = {nt, n_tasks, |klist|, active};
 * tasks[n_tasks][6] = I, J, offset and count into klist, piece r of R (I == J: accumulation of the diagonal tile, I == nt: rhs
 * row; a long contraction is cut into pieces that accumulate in place, the last piece finishes the tile), in the order in which
 * the persistent workgroups take them.  Executed in numpy by tests/test_chol_plan.py. */
int gtg_debug_df_plan(gtg_handle h, int64_t sizes[4], int32_t* tasks, int32_t* klist);
/* The same schedule as the kernels read it, copied back from the device: sizes = {int32 words of the task table (12 per task: the six
 * above, slot of the tile, slot of its diagonal tile, accumulator lanes, first scratch slot, the tile's 64-bit sub-tile mask), words of
 * the step table (6 per contraction step: slots of the two operand tiles, their sub-tile masks), words of the chain table}.  The tables
 * are resolved by a kernel (k_df_resolve) on a real runtime and by a host loop otherwise / with GTG_HOST_SYMBOLIC=1:
 * tests/test_gpu_device_analysis.py compares the two word for word. */
int gtg_debug_df_device_tables(gtg_handle h, int64_t sizes[3], int32_t* tasks, int32_t* steps, int32_t* chain);
/* The diagonal chains of the same schedule: sizes = {chain workgroups W, diagonal tiles, block columns}; workgroup w of the chain
 * kernel factors chain_tiles[chain_off[w] .. chain_off[w + 1]) in that order (one chain = two workgroups that alternate; several
 * chains when a nested-dissection ordering gave the elimination tree independent subtrees -- the reference's parallel elimination of
 * independent cliques, inference/ClusterTree-inst.h:218-317); seq = the order in which the block columns' tasks are queued. */
int gtg_debug_df_chains(gtg_handle h, int64_t sizes[3], int32_t* chain_off, int32_t* chain_tiles, int32_t* seq);
/* The elimination order of the reduced variables the analysis chose (or was given, gtg_set_reduced_ordering): the variable id at
 * every position, n = number of non-landmark variables.  (tests/test_device_analysis_spec.py pins a level-synchronous
 * formulation of the reverse Cuthill-McKee ordering against it.) */
int gtg_debug_reduced_order(gtg_handle h, int32_t* var_of_position, int32_t n);
/* out[0] = tickets taken in the last factorisation; out[8..15] = record of the first dependency wait that gave up (kind 1/2: tile
 * flags of a contraction step, 3: panel of a diagonal tile, 4: accumulated diagonal tile; I, J, k; flag values seen / wanted --
 * out[15], the second wanted value, is replaced by a host counter: the lambda tries of this handle that were repeated with the
 * stream schedule after a time-out of the dataflow pass) */
int gtg_debug_df_ctrl(gtg_handle h, int32_t out[16]);
/* Poll statistics of the dataflow factorisation's dependency waits, counted over the handle's life (round 6): out[0] waits that lasted
 * >= 256 polls, out[1] of those the ones that ended on the read-modify-write poll, out[2] of THOSE the ones whose next ordinary (sc1)
 * load of the same words still returned the old value -- a line the XCD's L2 kept serving --, out[3] / out[4] the same for the shadow
 * words (chol_dataflow.hip::wait_flags; tools/df_stress.py prints them). */
int gtg_debug_df_poll_stats(gtg_handle h, int64_t out[5]);
/* Device memory kept aside for the next handle (released blocks of >= 16 MB, at most GTG_ALLOC_CACHE_MB per device: default 2048,
 * 0 = nothing is kept; the default was 8192 in round 3 and 0 in round 4): give every kept block back to the driver.  Returns the bytes
 * released.  (A failed allocation does this by itself before it gives up.) */
int64_t gtg_release_cached_memory(void);
/* bytes of released device blocks the library currently keeps for the next handle of the process (all devices; GTG_ALLOC_CACHE_MB caps it per device,
 * default 2048, 0 = keep nothing) */
int64_t gtg_cached_memory_bytes(void);
/* GTG_DF_TRACE=1 (read at upload): 100 MHz time stamps of the last factorisation, n = 8 n_tasks + 2 nt: per task {taken,
 * contraction done, done, xcc << 32 | HW_ID, panel 0..3 of the diagonal tile seen}, then per diagonal tile {accumulated tile in, factored} (tools/df_trace.py) */
int gtg_debug_df_trace(gtg_handle h, int64_t* out, int64_t n);

/* tests: the hand-written scan / stable radix sort / run detection of the set-up passes (csrc/primitives.hip) on host arrays.
 * gtg_debug_sort_pairs: key_bytes 4 or 8; val NULL sorts keys only (8-byte keys).  Returns 0, or GTG_ERR_HIP. */
int gtg_debug_scan(int device, const int64_t* in, int64_t n, int64_t* out);
int gtg_debug_sort_pairs(int device, int key_bytes, const void* key, const uint32_t* val, int64_t n, int bits, void* key_out, uint32_t* val_out);
int gtg_debug_runs(int device, const uint64_t* sorted, int64_t n, uint64_t* uniq, int64_t* start, int32_t* n_runs);

/* ---- wire format on the bundle-adjustment side of the path: BAL text files straight to / from the
 * SoA arrays of gtg_problem.  Host-only (no GPU needed).  Replaces SfmData::FromBalFile (gtsam/sfm/SfmData.cpp:189-246: every
 * number goes through a `float`, pose = openGL2gtsam(Rodrigues(w), t), measurement (u, -v), observations grouped by point in
 * file order) and writeBAL (gtsam/sfm/SfmData.cpp:249-327: precision 20, gtsam2openGL, Rot3::Logmap).
 *   cams17  : n_cams x 17 = Pose3 (R row-major 9, t 3) + Cal3Bundler (f, k1, k2, u0, v0)   -- the packed SfmCamera value
 *   points3 : n_points x 3
 *   obs_*   : n_obs observations (camera index, point index, (u, v)), grouped by point
 * Errors: GTG_ERR_USAGE with gtg_io_last_error() (file missing = the reference's runtime_error text). */
const char* gtg_io_last_error(void);
int gtg_io_bal_sizes(const char* path, int64_t* n_cams, int64_t* n_points, int64_t* n_obs);
int gtg_io_read_bal(const char* path, int64_t n_cams, int64_t n_points, int64_t n_obs, double* cams17, double* points3,
                    int32_t* obs_cam, int32_t* obs_point, double* obs_z);
int gtg_io_write_bal(const char* path, int64_t n_cams, int64_t n_points, int64_t n_obs, const double* cams17,
                     const double* points3, const int32_t* obs_cam, const int32_t* obs_point, const double* obs_z);

/* ---- wire formats on the pose-graph side of the path: g2o (VERTEX_SE2 / EDGE_SE2 / VERTEX_SE3:QUAT /
 * EDGE_SE3:QUAT) and TORO (VERTEX2 / EDGE2 / VERTEX3 / EDGE3, also VERTEX / EDGE / ODOMETRY) text files straight to / from the arrays the
 * between-factor tables of gtg_problem are filled from.  Host-only.  Replaces readG2o / load2D / load3D (gtsam/slam/dataset.cpp:505-570,
 * 621-633, 738-944) and writeG2o (:636-735).
 *   is_3d = 1: poses are 12 doubles (R row-major 9, t 3), noise parameters 36 doubles per edge; vertices only where the file lists them
 *              (load3D does not create missing ones);
 *   is_3d = 0: poses are (x, y, theta), noise parameters 9 doubles per edge; `noise_format` says what the six numbers of an edge line
 *              are (readG2o: GTG_IO_NOISE_G2O; load2D's default: GTG_IO_NOISE_AUTO); a vertex an edge refers to and the file does not
 *              list is created along the odometry chain like load2D does;
 *   edge_noise_kind / edge_noise: GTG_NOISE_* + sigma | sigmas | R row-major, the model Gaussian::Information / Covariance(M, smart = true)
 *              returns (linear/NoiseModel.cpp:83-131);  vertices sorted by key (the order gtsam::Values iterates in).
 * gtg_io_write_g2o: VERTEX_* lines of the estimate, then EDGE_* lines with the upper triangle of the information matrix (3-D: in g2o's
 * t,R block order); numbers at the stream's default precision (%g) as writeG2o prints them, or -- full_precision = 1 -- with 17 digits.
 * Errors: GTG_ERR_USAGE with gtg_io_last_error() (file missing = the reference's invalid_argument text). */
enum { GTG_IO_NOISE_AUTO = 0, GTG_IO_NOISE_G2O = 1, GTG_IO_NOISE_TORO = 2, GTG_IO_NOISE_GRAPH = 3, GTG_IO_NOISE_COV = 4 };
int gtg_io_g2o_sizes(const char* path, int is_3d, int noise_format, int64_t* n_edges, int64_t* n_vertices);
int gtg_io_read_g2o(const char* path, int is_3d, int noise_format, int64_t n_edges, int64_t n_vertices, int64_t* edge_v1, int64_t* edge_v2,
                    double* edge_z, int32_t* edge_noise_kind, double* edge_noise, int64_t* vertex_key, double* vertex_pose);
int gtg_io_write_g2o(const char* path, int is_3d, int64_t n_edges, const int64_t* edge_v1, const int64_t* edge_v2, const double* edge_z,
                     const int32_t* edge_noise_kind, const double* edge_noise, int64_t n_vertices, const int64_t* vertex_key,
                     const double* vertex_pose, int full_precision);

#ifdef __cplusplus
}
#endif
#endif /* GTSAM_AMD_H */
