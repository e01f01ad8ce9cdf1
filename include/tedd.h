/* tedd.h -- C ABI of libtedd_hip.so: the MI355X (gfx950) implementation of the time-stepping hot
 * path of phao/NA-nonlinear-temperature-enhanced-diffusion-model-DD.
 *
 * The reference has no FFI; its boundary for this path is the Python class contract
 *   integrator_cls(semi_discrete_field=field, **params).step(at_t0: StateVars, *, t0, dt) -> StateVars
 * (reference src/prob1base.py:2879-2882, 3117-3149; instantiated at src/mms_trial_utils.py:247-249).
 * The host-side Python classes that keep that contract call the entry points below through ctypes.
 * Every comment "replaces" cites the reference code (file:line under src/) whose work the entry
 * point performs on the GPU.
 *
 * Conventions
 *  - All arrays are IEEE binary64.  Host arrays are dense C-order (rows x (M+1)); index i (x, slow
 *    axis) by j (y, fast axis), as reference src/prob1base.py:243-248,353.
 *  - A context owns one slab of rows [row_lo, row_hi) of the (N+1) x (M+1) mesh plus `ghost` rows
 *    on each side that is not a physical boundary; host arrays passed to/from a context cover
 *    its local rows (ghost rows included), i.e. tedd_local_rows() x (M+1).
 *  - All calls return 0 on success or a negative tedd_status; tedd_last_error() describes it.
 *  - Calls on one context are not thread-safe.  Work is enqueued on the context's HIP stream;
 *    calls that return host data synchronise that stream.
 */
#ifndef TEDD_H
#define TEDD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tedd_ctx tedd_ctx;

enum tedd_status {
    TEDD_OK = 0,
    TEDD_ERR_INVALID = -1,     /* bad argument (maps to ValueError) */
    TEDD_ERR_ASSERT = -2,      /* a precondition the reference asserts (dt > 0, shapes): AssertionError */
    TEDD_ERR_RUNTIME = -3,     /* HIP failure, out of memory, unsupported configuration: RuntimeError */
    TEDD_ERR_NOT_CONVERGED = -4 /* a Jacobi solve missed its tolerance (tedd_check_converged, tedd_advance) */
};

enum tedd_mms_case {
    TEDD_MMS_STREAM = -1, /* forcing arrays are uploaded with tedd_set_forcing */
    TEDD_MMS_POL = 0,     /* src/prob1_mms_cases.py:250-277 */
    TEDD_MMS_EXPSIN = 1,  /* src/prob1_mms_cases.py:280-337 */
    TEDD_MMS_PEAKS = 2,   /* src/prob1_mms_cases.py:151-247; params [leading_spatial_const, evol_speed] */
    TEDD_MMS_NFS = 3,     /* src/prob1_mms_cases.py:406-511; params [gamma_cp,gamma_T,gamma_cl,gamma_cd,gamma_cs,theta] */
    TEDD_MMS_PROGRAM = 4  /* any exact solution, as expression programs set with tedd_set_mms_program */
};

enum tedd_reaction {
    TEDD_REACT_CSTRIPLE = 0,    /* src/prob1base.py:2842-2876 */
    TEDD_REACT_REGHCSTRIPLE = 1,/* src/prob1base.py:3553-3593 */
    TEDD_REACT_HCSTRIPLE = 2    /* src/prob1base.py:3298-3337: sharp Heaviside H(cs) = (cs > 0); the cs corrector
                                   is the sign-case solve of :3376-3430, whose ValueError ("Denominator ... below
                                   positiveness treshold") is reported as TEDD_ERR_INVALID by tedd_get_stats /
                                   tedd_advance, the first calls that synchronise after the step */
};

typedef struct tedd_config {
    int32_t N, M;             /* mesh has (N+1) x (M+1) nodes; N, M >= 1 (a 1 x 1 mesh has no interior node:
                                 a step then only sets the boundary values) */
    int32_t row_lo, row_hi;   /* owned global rows [row_lo, row_hi); 0 and N+1 on a single device */
    int32_t ghost;            /* ghost rows kept on each non-physical side (0 on a single device) */
    int32_t model_kind;       /* 1 = DefaultModel01, 2 = DefaultModel02 (src/prob1base.py:71-217) */
    int32_t reaction_kind;    /* tedd_reaction */
    int32_t mms_case;         /* tedd_mms_case */
    int32_t device;           /* HIP device ordinal; -1 = current device */
    int32_t reserved;
    double eta;               /* regularization_factor of H_eta */
    double consts[17];        /* ModelConsts in the reference's field order (src/prob1base.py:28-45) */
    double mms_params[8];
} tedd_config;

/* Create a context for the grid x[0..N], y[0..M] (any strictly increasing coordinates; replaces
 * Grid(x, y), src/prob1base.py:220-362, FiveDiagonalsMatrixFactory :601-619 and the lambdified MMS
 * set-up :1226-1347). */
int tedd_create(tedd_ctx** out, const tedd_config* cfg, const double* x, const double* y);
void tedd_destroy(tedd_ctx* ctx);
const char* tedd_last_error(const tedd_ctx* ctx); /* ctx may be NULL: error of the last failed tedd_create */

/* Re-read the model constants, eta and MMS parameters (the reference's tests mutate the model in
 * place between steps, e.g. tests/test_time_integrator_hcs_triple.py:377-387). */
int tedd_set_model(tedd_ctx* ctx, const double consts[17], int32_t model_kind, double eta,
                   const double mms_params[8]);
/* The exact solution of a TEDD_MMS_PROGRAM context.  Replaces MMSCaseSymbolic's lambdified SymPy
 * expressions (src/prob1base.py:1225-1347): the caller lowers the expressions of (t, x, y) to straight-line
 * programs of three-address operations on at most 96 double registers (registers 0, 1, 2 hold t, x, y on
 * entry), which every node then runs on the device.  which 0: the node program, 21 outputs
 *   cp cp_t cp_x cp_y | T T_t T_x T_y lap_T | cl cl_t cl_x cl_y lap_cl | cd cd_t cd_x cd_y lap_cd | cs cs_t;
 * which 1: the cell program, 4 outputs cp cp_t cl T, run at the 3 x 3 Gauss points of a node's cell for the
 * cell-averaged cp-forcing (src/prob1base.py:493-598, 2313-2328).  Both must be set before the first call that
 * needs the exact solution or the forcing.  Word: op | dst << 8 | a << 16 | b << 24 with op
 *   0 CONST dst = consts[a | b << 8]   1 ADD  2 SUB  3 MUL  4 DIV  5 NEG   6 POWI dst = reg[a] ** (int8) b
 *   7 POW  8 EXP  9 SIN  10 COS  11 SQRT  12 LOG  13 ABS  14 SIGN  15 TANH  16 OUT out[dst] = reg[a]  17 MOV.
 * Every word is validated (register, constant and output indices, use before definition, all outputs
 * written); a malformed program is TEDD_ERR_INVALID. */
int tedd_set_mms_program(tedd_ctx* ctx, int32_t which, const uint32_t* code, int32_t n_ops, const double* consts,
                         int32_t n_consts, int32_t n_regs);
/* Execution options (no counterpart in the reference; they select between implementations that give
 * the same results).  "strip_solve" (default 1): solve the Newton systems with the register-resident
 * kernel chain (several Jacobi sweeps per launch) instead of one launch per sweep; "strip_hmax": sweeps
 * per launch (0 = automatic, at most 16); "strip_shape" (default -1: by mesh size and system; 0: 64-row tiles
 * of 4 strips, 4: 64-row tiles of 8 strips for the T system (cl, cd: as -1), 1: 32-row tiles of 8 short strips for meshes too small to fill the GPU, 2: 48-row tiles of 12
 * short strips, 3: 16-row tiles of 8 strips of two rows for the cl and cd systems (T: as 1) on meshes with fewer
 * small tiles than compute units); "cache_forcing" (default 1): evaluate the closed-form MMS
 * forcing once per time level into arrays instead of inside the step kernels; "fuse_forcing" (default -1:
 * slabs of at most 4 M nodes): those arrays are made by the predictor launch instead of a launch of their own;
 * "fuse_rhs" (default -1: slabs of at most 4 M nodes): a cl or cd solve that is a single launch evaluates its
 * right-hand side itself instead of reading it from an array a launch of its own wrote (bit-identical results);
 * "fuse_corrector" (default 1): such a cd launch of the last Newton step also carries the cp corrector and the
 * record pass of the RegHCsTriple cs corrector on whole-mesh contexts (one launch fewer, bit-identical results);
 * "warm_start" (default 1): a Newton solve whose sweep count the library chooses starts its Jacobi iteration
 * from the increment of the step before instead of zero (same tolerance, fewer sweeps); "march" (default -1:
 * the right-hand sides on slabs above 4 M nodes, the predictor from 2^27 nodes; 0 never, 1 always): the predictor and the right-hand sides of the cl and cd
 * systems run in the marching launcher (a wavefront walks down 64 columns with the stencil rows in registers:
 * every array row is read once) instead of one thread per node -- bit-identical results; "march_rows",
 * "march_occ": rows per segment and waves per SIMD of that launcher (0: the kernel's own choice); "march_lds"
 * (default -1: the right-hand sides; 0 never): a marching launch without look-ahead in registers has the next
 * row written into LDS by the load unit while it computes the present one -- bit-identical results;
 * "march_split" (default -1 = 1): the marching launches wait for the next row's loads between a node's arithmetic
 * and its stores (0: where the compiler puts the wait, at the top of the next row behind the stores: the launches of
 * before, kept for comparison) -- bit-identical results; "cs_lean"
 * (default 1): RegHCsTriple corrector launches of at most 5 Newton iterations use a build that keeps 5 instead of
 * 8 per-iteration records per thread and runs 5 instead of 4 waves per SIMD -- same arithmetic;
 * "forcing_columns" (default 1): the forcing arrays of a closed-form case are made by threads that walk down
 * columns with the column's table entries in registers, on slabs large enough to fill the GPU that way
 * (0: one thread per node everywhere) -- bit-identical results; "strip_panel" (default 8; 0: row-major):
 * tile rows per panel of the order in which the solve launches' tiles are dealt to the GPU (down the columns of
 * a panel first, so that the rows two tiles share are re-read from L2) -- same results; "persistent" (default
 * 0; RegHCsTriple, whole-mesh contexts of at most 4 M nodes, at most 8 cs Newton iterations; other calls keep
 * separate launches): the launches of the predictor-corrector steps of one call (tedd_step_pc, tedd_advance
 * with its norms) are recorded and run as ONE persistent launch -- 1: one workgroup, phases separated by the
 * workgroup barrier; k > 1: k workgroups and a barrier over the grid -- with the per-sweep solver from zero
 * iterates, bit-identical to "strip_solve" 0 + "warm_start" 0.  Slower than separate launches on every mesh
 * measured on MI355X, hence off; a grid barrier that does not fill within ~0.2 s (compute units held by
 * another program) ends the launch and the next synchronising call returns TEDD_ERR_RUNTIME;
 * "prog_codegen" (TEDD_MMS_PROGRAM contexts; default -1): the two expression programs are turned into HIP
 * source, compiled for the context's device at first use (hiprtc; the headers under ../csrc beside the library,
 * or under $TEDD_CSRC_DIR, are included) and run as ordinary kernels with the program registers in VGPRs --
 * -1: where that succeeds, the interpreter otherwise; 1: or TEDD_ERR_INVALID with the compiler's message;
 * 0: interpreter only -- bit-identical arrays either way, about twice the steps per second on a 256^2 mesh;
 * "jacobi_floor" (default 2^-62; 0 switches it off): floor of the stopping rule of the Newton solves relative to the
 * size of the unknown -- a last update below jacobi_floor x max|u| counts as converged whatever the increment is
 * (2^-10 of an ulp of the unknown).  It binds only where jacobi_tol x increment is smaller still, i.e. where a
 * time step moves the unknown by less than jacobi_floor / jacobi_tol (2e-6) of its size, and only for
 * warm-started solves, which record max|u|. */
int tedd_set_option(tedd_ctx* ctx, const char* name, double value);
/* Use the caller's HIP stream (hipStream_t passed as void*) for all subsequent work. */
int tedd_set_stream(tedd_ctx* ctx, void* hip_stream);
int tedd_sync(tedd_ctx* ctx);
int32_t tedd_local_rows(const tedd_ctx* ctx);
int32_t tedd_row_offset(const tedd_ctx* ctx); /* global index of local row 0 */
int32_t tedd_pitch(const tedd_ctx* ctx);      /* device row pitch in doubles (>= M+1) */
/* How the context's field buffers were placed in device memory, as text (no counterpart in the reference, whose
 * arrays are NumPy's).  A context whose ~32 field buffers take 8 GiB or more carves them from ONE allocation, two
 * buffer sizes apart while that fits into half of the free device memory (the gaps stay unused), packed back to
 * back otherwise: launches that stream many arrays at once run up to 8 % slower with the buffers in consecutive
 * separate allocations.  TEDD_FIELD_GAP_MB in the environment sets the gap (MiB; 0 packed, -1 separate
 * allocations).  An arena of a quarter of the device memory or more is streamed through when it is taken and
 * taken again if it lies in the slower of the two kinds of memory such an allocation is given in turn (the
 * measured rates are part of the string; TEDD_FIELD_PROBE=0 keeps the first allocation).  The string names the layout chosen, or why the buffers were allocated one by one; it belongs to the
 * context and is valid until tedd_destroy. */
const char* tedd_placement(const tedd_ctx* ctx);

/* State (cp, T, cl, cd, cs): replaces StateVars' five arrays (src/prob1base.py:1959-2028). */
int tedd_set_state(tedd_ctx* ctx, const double* cp, const double* T, const double* cl, const double* cd,
                   const double* cs);
int tedd_get_state(tedd_ctx* ctx, double* cp, double* T, double* cl, double* cd, double* cs);
/* Device addresses of the current state arrays (pitched, local rows); for halo exchange. */
int tedd_state_device_ptrs(tedd_ctx* ctx, void* ptrs[5]);
/* Tell the context that the caller refreshed the ghost rows of the current state arrays (after a
 * halo exchange written through tedd_state_device_ptrs). */
int tedd_ghosts_refreshed(tedd_ctx* ctx);
/* Load the state with the exact MMS solution at time t (state_from_mms_when, src/prob1base.py:3433-3449). */
int tedd_set_state_exact(tedd_ctx* ctx, double t);

/* Forcing arrays for TEDD_MMS_STREAM at the two time levels of the next step (which: 0 = t0,
 * 1 = t0+dt): cell-averaged fcp, fT, fcl, fcd, fcs (src/prob1base.py:2313-2425, 3468-3551). */
int tedd_set_forcing(tedd_ctx* ctx, int32_t which, const double* fcp, const double* fT, const double* fcl,
                     const double* fcd, const double* fcs);

/* Evaluate the forcing arrays of time level t ahead of the step that needs them, keeping the arrays of
 * level keep_t (the time the next step starts from).  No counterpart in the reference (it evaluates its
 * lambdified forcing inside every step, src/prob1base.py:2313-2425): the forcing does not depend on the
 * state, so a multi-GPU driver enqueues this before it exchanges ghost rows and the launch overlaps the
 * transfer.  A no-op when the forcing is streamed or evaluated inside the step kernels. */
int tedd_prefetch_forcing(tedd_ctx* ctx, double t, double keep_t);

/* One predictor-corrector step of P_ModifiedEuler_C_Trapezoidal_TimeIntegrator_{RegHCsTriple,CsTriple}
 * (src/prob1base.py:3117-3149 with 2953-3115, 3174-3219, 3631-3702).  The 5-diagonal systems the
 * reference solves with spsolve (src/prob1base.py:2103,2130) are solved by matrix-free Jacobi
 * sweeps: jacobi_sweeps[v] > 0 fixes the sweep count for v = T, cl, cd (from a zero iterate); <= 0 lets the
 * library choose it so that the last update is below jacobi_tol relative to the solution increment, or below the
 * floor relative to the unknown itself (option "jacobi_floor"), whichever is larger (capped at jacobi_max): from the contraction bound of the system and, in a run of consecutive steps, from the
 * convergence record of the step two before (the iteration then starts from the last step's increment).
 * Solves with a chosen count are judged afterwards (tedd_check_converged).  jacobi_sweeps may be NULL. */
int tedd_step_pc(tedd_ctx* ctx, double t0, double dt, int32_t num_pc_steps, int32_t num_newton_steps,
                 int32_t cs_newton_iterations, double cs_consec_xs_rtol, const int32_t jacobi_sweeps[3],
                 double jacobi_tol, int32_t jacobi_max);
/* ForwardEulerIntegrator.step (src/prob1base.py:2885-2903). */
int tedd_step_feuler(tedd_ctx* ctx, double t0, double dt);

/* nsteps steps from t0 with the error norms of run_simulation_collect_data fused in
 * (src/mms_trial_utils.py:56-147): err_series receives (nsteps+1) x 8 doubles, per time level
 * ||e_v||_H^2 for v = cp,T,cl,cd,cs then ||grad_H e_v||_p^2 for v = T,cl,cd (owned rows only).
 * integrator: 0 = forward Euler, 1 = predictor-corrector.  err_series may be NULL. */
int tedd_advance(tedd_ctx* ctx, double t0, double dt, int32_t nsteps, int32_t integrator, int32_t num_pc_steps,
                 int32_t num_newton_steps, int32_t cs_newton_iterations, double cs_consec_xs_rtol,
                 const int32_t jacobi_sweeps[3], double jacobi_tol, int32_t jacobi_max, double* err_series);

/* The pieces of a predictor-corrector step, as the reference exposes them on the integrator.  The state
 * set by tedd_set_state is their `at_t` / `at_t0`; all arrays are host arrays of the local mesh shape.
 *
 * tedd_predict_cp_cs: initial_cp_pred (src/prob1base.py:2953-2965) and initial_cs_pred (:3174-3188
 * CsTriple, :3354-3374 HCsTriple, :3631-3645 RegHCsTriple); either output may be NULL.
 * tedd_correct_cp_cs: corrector_cp_step (:2967-2996) and corrector_cs_step (:3190-3219, :3376-3430,
 * :3665-3702) given the new T, cl, cd; either output may be NULL.
 * tedd_newton_step: newton_step_T / _cl / _cd (:2998-3045, :3047-3080, :3082-3115) for var = 1 / 2 / 3 with
 * the state as the lagged state: Y = dt F_v(S0,t0) + 2 v0 from the caller, T1 (var 2, 3) and cl1 (var 3) the
 * already updated unknowns; out = the updated unknown, residual (may be NULL) = last_residual[v]. */
int tedd_predict_cp_cs(tedd_ctx* ctx, double t, double dt, double* cp1, double* cs1);
int tedd_correct_cp_cs(tedd_ctx* ctx, double t0, double dt, const double* T1, const double* cl1, const double* cd1,
                       int32_t cs_newton_iterations, double cs_consec_xs_rtol, double* cp1, double* cs1);
int tedd_newton_step(tedd_ctx* ctx, int32_t var, double t0, double dt, const double* Y, const double* T1,
                     const double* cl1, int32_t jacobi_sweeps, double jacobi_tol, int32_t jacobi_max, double* out,
                     double* residual);

/* last_residual["T"|"cl"|"cd"] of the last predictor-corrector step (src/prob1base.py:3041-3043,
 * 3076-3078, 3111-3113); computed on demand. */
int tedd_get_residuals(tedd_ctx* ctx, double* rT, double* rcl, double* rcd);

/* Parity hooks.  F(state, t): Fcp, FT, Fcl, Fcd, Fcs (src/prob1base.py:2599-2672). */
int tedd_eval_field(tedd_ctx* ctx, double t, double* Fcp, double* FT, double* Fcl, double* Fcd, double* Fcs);
/* Exact MMS solution at the nodes (closed-form cases). */
int tedd_eval_exact(tedd_ctx* ctx, double t, double* cp, double* T, double* cl, double* cd, double* cs);
/* Derivatives of the exact MMS solution at the nodes (MMSCaseBase.dt_*, dx_*, dy_*, lap_*,
 * src/prob1base.py:736-798): what 0 value, 1 d/dt, 2 d/dx, 3 d/dy, 4 Laplacian.  Entries the forcing
 * terms never use (Laplacian of cp, space derivatives of cs) come back as NaN. */
int tedd_eval_mms(tedd_ctx* ctx, double t, int32_t what, double* cp, double* T, double* cl, double* cd, double* cs);
/* The five forcing arrays at time t (closed-form cases). */
int tedd_eval_forcing(tedd_ctx* ctx, double t, double* fcp, double* fT, double* fcl, double* fcd, double* fcs);
/* Jacobian stencil coefficients at the current state: op 0 dFT/dT, 1 dFcl/dT, 2 dFcl/dcl, 3 dFcd/dT,
 * 4 dFcd/dcl, 5 dFcd/dcd (src/prob1base.py:2674-2839); out arrays: (0,0),(+1,0),(-1,0),(0,+1),(0,-1). */
int tedd_eval_jacobian(tedd_ctx* ctx, int32_t op, double* c, double* e, double* w, double* n, double* s);
/* Error norms of the current state against the exact solution at t: 8 doubles as in tedd_advance
 * (src/mms_trial_utils.py:81-110).  For TEDD_MMS_STREAM pass the exact fields in exact[5]. */
int tedd_error_norms(tedd_ctx* ctx, double t, const double* const exact[5], double out[8]);

/* tedd_error_norms for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same kernels and the same sums,
 * bit for bit, as tedd_error_norms for a closed-form MMS case, but the call allocates nothing on the host
 * runtime or the device: the per-workgroup partial sums go to a fixed 128 KiB array in the library's device
 * image instead of the context's buffer, no timing events are taken, and the 8 results are written to
 * out_device, device memory of the caller (8 doubles), by work enqueued on the context's stream -- the
 * call does not synchronise.  Limits: closed-form cases only (TEDD_MMS_POL .. TEDD_MMS_NFS), slabs of at
 * most 2^26 nodes; anything else is TEDD_ERR_INVALID.  Every call on a GPU shares that GPU's array: calls
 * that overlap in time on one GPU (two contexts on different streams) race. */
int tedd_error_norms_noalloc(tedd_ctx* ctx, double t, double* out_device);

/* Record of the last step: sweeps[3] Jacobi sweeps used for T, cl, cd; last_update[3] and
 * increment[3]: max-norms of the last sweep's update and of the solution increment;
 * cs_iterations: Newton iterations the cs corrector kept (global early exit, src/prob1base.py:3661);
 * warm_started[v]: that solve started from the increment of the step before instead of zero, and
 * first_update[v] is the max-norm of its first sweep's update; worst_ratio[v]: the largest
 * last_update / max(jacobi_tol x increment, jacobi_floor x max|u|) over the steps since the last
 * tedd_check_converged, counting only solves whose sweep count the library chose. */
typedef struct tedd_step_stats {
    int32_t sweeps[3];
    int32_t cs_iterations;
    double last_update[3];
    double increment[3];
    int32_t warm_started[3];
    int32_t reserved;
    double first_update[3];
    double worst_ratio[3];
} tedd_step_stats;
int tedd_get_stats(tedd_ctx* ctx, tedd_step_stats* out);

/* The reference solves its Newton systems directly (spsolve, src/prob1base.py:2103,2130), which cannot stop
 * short; Jacobi sweeps can.  Every solve whose sweep count the library chose is therefore judged on the
 * device after the fact (last update <= max(jacobi_tol x increment, jacobi_floor x max|u|)) and the worst ratio of
 * all steps is kept.
 * tedd_check_converged returns TEDD_ERR_NOT_CONVERGED if some step since the last check missed its tolerance by
 * more than a factor 10, and clears the record.  tedd_advance performs this check before it returns; callers
 * that step with tedd_step_pc call it at their own synchronisation points (it synchronises the stream). */
int tedd_check_converged(tedd_ctx* ctx);

/* Reduction across the slabs of a decomposed mesh.  The reference decides the early exit of the cs corrector's
 * Newton iteration from maxima and minima over ALL nodes (src/prob1base.py:3661); a context sees its own rows
 * only.  With a hook set, the context calls it (on the calling thread, inside tedd_step_pc /
 * tedd_correct_cp_cs) between the pass that records, per iteration k, dev_max[k] = max|dx_k| and
 * dev_min[k] = min|x_k| over its rows and the pass that acts on them; the hook must replace dev_max[k] by the
 * maximum and dev_min[k] by the minimum over all slabs (n values each, device memory of this context), by work
 * ordered on the context's stream, and return 0 (anything else fails the step with TEDD_ERR_RUNTIME).
 * fn = NULL removes the hook. */
typedef int (*tedd_reduce_fn)(void* user, double* dev_max, double* dev_min, int32_t n);
int tedd_set_reduce_hook(tedd_ctx* ctx, tedd_reduce_fn fn, void* user);

/* The convergence record of the Newton solves of the step in flight: *n non-negative doubles in the context's
 * device memory, each a maximum over the context's rows (last update, increment, first update of a warm-started
 * solve, size of the unknown, per system).  No counterpart in the reference (one process, direct solves).  The
 * library chooses the sweep counts and the warm start of later steps from these records; on a decomposed mesh the
 * driver replaces every entry by its maximum over all slabs after each step (one all-reduce with MAX of *n doubles,
 * ordered on the context's stream after tedd_step_pc), so that every slab takes the decisions a single context
 * holding the whole mesh would take. */
int tedd_record_device_ptr(tedd_ctx* ctx, void** ptr, int32_t* n);

/* Ghost rows in flight.  On a decomposed mesh the ghost rows of the new state travel between the slabs after every
 * step (no counterpart in the reference, which is one process).  A driver that does not want to wait for them calls
 * tedd_ghosts_pending after it has posted the transfer: the context then treats the ghost rows of the current state
 * as not yet valid.  The next tedd_step_pc runs its first launch (the predictor, the largest of the step) on the rows
 * whose stencils do not reach into the ghost zone, calls the ghost hook -- which must complete the transfer into the
 * arrays of tedd_state_device_ptrs, by work ordered on the context's stream, and return 0 -- and then runs that
 * launch on the two remaining bands and the rest of the step: the transfer overlaps the bulk of the predictor.
 * Without a hook, tedd_ghosts_refreshed ends the pending state; any other use of a state whose ghost rows are
 * pending is TEDD_ERR_INVALID. */
typedef int (*tedd_ghost_fn)(void* user);
int tedd_set_ghost_hook(tedd_ctx* ctx, tedd_ghost_fn fn, void* user);
int tedd_ghosts_pending(tedd_ctx* ctx);

/* Per-kernel device timing with HIP events on the context's stream.  enable != 0 starts
 * recording (and clears the totals).  tedd_profile_get returns, for up to `cap` kernel families,
 * the name, the number of launches and the summed milliseconds; returns the number of families. */
int tedd_profile_enable(tedd_ctx* ctx, int32_t enable);
int tedd_profile_get(tedd_ctx* ctx, int32_t cap, const char** names, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* TEDD_H */
