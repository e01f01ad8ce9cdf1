// Context of one slab: geometry tables, device buffers, and the sequencing of one
// predictor-corrector / forward-Euler step over the node functors of tedd_nodes.hpp.
// The class is a template over the execution backend: the shipped library instantiates it
// with the HIP backend (tedd_hip_backend.hpp); tests/cpu_twin instantiates it with a host
// loop backend to debug the sequencing without a GPU.
//
// Reference sequencing: src/prob1base.py:3117-3149 (step), 2998-3115 (Newton steps),
// 2885-2903 (forward Euler); src/mms_trial_utils.py:56-147 (advance with error norms).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/tedd.h"
#include "tedd_mms_host.hpp"
#include "tedd_nodes.hpp"
#include "tedd_prog.hpp"

namespace tedd {

struct AssertError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct InvalidError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct NotConverged : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// kernel families for the timing record
enum Fam : int {
    FAM_PREDICT = 0, FAM_RHS_T, FAM_RHS_CL, FAM_RHS_CD, FAM_JACOBI_T, FAM_JACOBI_CL, FAM_JACOBI_CD,
    FAM_CORRECT, FAM_CS_NEWTON, FAM_FIELD, FAM_RESIDUAL, FAM_NORMS, FAM_MISC, FAM_SOLVE_T, FAM_SOLVE_CL,
    FAM_SOLVE_CD, FAM_FORCING, FAM_COUNT
};
inline const char* fam_name(int f) {
    static const char* n[FAM_COUNT] = {"predict", "rhs_T", "rhs_cl", "rhs_cd", "jacobi_T", "jacobi_cl",
                                       "jacobi_cd", "correct", "cs_newton", "field", "residual", "norms",
                                       "misc", "solve_T", "solve_cl", "solve_cd", "forcing"};
    return n[f];
}

struct Range {
    int lo, hi;
};

template <class Backend>
class Ctx {
public:
    Backend be;
    tedd_config cfg;
    int N, M, nrows, ncols, ld, i0, gl, gr;
    std::vector<double> xg, yg;  // global coordinates
    ModelC model;
    MmsParams mp;
    int mms_case, react;
    std::string last_error;

    // device geometry + MMS tables
    double* d_geo_rows = nullptr;  // rh, rhm, hm, h : 4 x nrows
    double* d_geo_cols = nullptr;  // rk, rkm, km, k : 4 x ld
    double *d_bx = nullptr, *d_by = nullptr, *d_bxq = nullptr, *d_byq = nullptr;
    double q_metric = 0.0;  // max over the mesh of Cx- + Cx+ + Cy- + Cy+

    // field buffers: a pool of (nrows x ld) arrays.  Named slots are dedicated; state and
    // intermediate unknowns are acquired from / released to the free list.
    enum Slot : int { Y0 = 0, RHS = 3, DA, DB, FCPA, FCPB, SF00 /*5*/, SF10 = 13 /*5*/, EX0 = 18 /*5*/, NSLOT = 23 };
    double* slot[NSLOT] = {nullptr};
    static constexpr int kArenaSlots = 32;  // field buffers a predictor-corrector context holds at its peak
    const char* placement() const {
        if constexpr (Backend::kHasArena) return be.placement();
        return "host memory";
    }
    std::vector<double*> all_bufs, free_bufs;
    std::vector<double*> retained;  // kept alive after a step for tedd_get_residuals
    double* cur[5];
    std::map<const double*, Range> valid;

    // small device scratch: reductions and records
    double* d_red = nullptr;      // 64 doubles: [0..15] convergence record of the step in flight, [16..18] sticky
                                  // worst ratios (see StepRecordArgs), [32..39] error norms
    // History of the convergence records.  Every predictor-corrector step has a sequence number that is
    // never reused; the prologue launch of step g+1 publishes the record of step g to ring slot g % kRing
    // (host-visible memory), and step g+kLag reads it there to choose its sweep counts -- by then the
    // prologue of step g+1 has run unless the host is more than kLag-1 steps ahead of the GPU, in which
    // case it waits for that one small launch, never for a step.
    static constexpr int kRing = 8;
    static constexpr int kLag = 4;
    double* h_ring = nullptr;     // kRing x 16 doubles, host-visible
    double* d_ring = nullptr;     // the same memory as the device addresses it
    struct StepMeta {
        long long seq = -1;
        int sweeps[3] = {0, 0, 0};
        bool warm[3] = {false, false, false};
        double tol[3] = {0, 0, 0};  // > 0: the library chose the sweeps of that system (its record is judged)
        double rho[3] = {0, 0, 0};  // a-priori contraction bound the choice was made with
    } meta[kRing];
    long long next_seq = 0;       // sequence number of the next predictor-corrector step
    long long chain_start = 0;    // first sequence number of the present run of consecutive steps
    bool record_unfolded = false; // step next_seq-1 left a record that has not been published to the ring
    bool record_unjudged = false; // ... and has not been folded into the sticky ratios
    bool judged_since_check = false;
    // T, cl, cd of the time level before the current state (warm start of the Newton solves)
    double* prev[3] = {nullptr, nullptr, nullptr};
    double prev_dt = 0.0;
    int opt_warm = 1;             // start the Jacobi iteration from the increment of the step before
    // Floor of the stopping rule of the Newton solves, relative to the size of the unknown: a last update below
    // opt_floor x max|u| counts as converged whatever the increment is (2^-62: 2^-10 of an ulp of the unknown; the
    // verdict allows ten times that, 1 % of an ulp).  It comes into play only where tol x increment is smaller
    // still, i.e. where a time step moves the unknown by less than 2e-6 of its size (jacobi_tol = 1e-13): the
    // 16384^2 mesh with dt = h^1.5 (T moves by 9e-9 per step: tol x increment = 1e-21), not the study meshes up
    // to 256^2.  Known only to warm-started solves, whose first sweep records max|u|.
    double opt_floor = 0x1p-62;
    // a warm start reads two more arrays than a start from zero: where the record says it would not save
    // three sweeps or a launch, that system starts from zero until this sequence number (then it is tried again)
    long long cold_until[3] = {0, 0, 0};
    // ... for 64 steps after the first such verdict, twice as long after every further one (the steps that try
    // the warm start again are the dearest of a run on a small mesh: two more arrays and a second launch)
    static constexpr int kColdSpell = 64, kColdSpellMax = 4096;
    int cold_spell[3] = {kColdSpell, kColdSpell, kColdSpell};
    double* d_series = nullptr;   // error-norm series of tedd_advance
    size_t series_cap = 0;
    double* d_partials_all = nullptr;  // partial sums of the norms of all steps of a tedd_advance call
    size_t partials_cap = 0;
    int* d_ints = nullptr;        // [0] cs iterations kept, [1] HCsTriple corrector: denominator not positive

    // record of the last PC step (for residuals and stats)
    struct LastStep {
        bool have = false;
        double t1 = 0, dt = 0;
        const double *cp_used = nullptr, *cs_used = nullptr;
        int sweeps[3] = {0, 0, 0};
        int cs_iters_requested = 0;
        bool cs_newton_ran = false;
        // Y_T was not stored (nothing in the step read it): what remakes it for the residuals
        const double *lazy_T0 = nullptr, *lazy_cp0 = nullptr, *lazy_fT0 = nullptr;
    } last;
    bool stats_pending = false;

    // cp-forcing cell-average cache: fcp_buf[k] holds the array at time fcp_time[k]
    double fcp_time[2] = {NAN, NAN};
    bool stream_forcing_set[2] = {false, false};
    // forcing cache (closed-form cases): the five forcing arrays of a time level are evaluated
    // once by ForcingFn and the step kernels read them like streamed forcing
    int opt_cache_forcing = 1;
    // with the cache: the predictor launch evaluates the t1 set itself (one launch fewer: +2..4 % on
    // launch-bound meshes) or a launch of its own does (same time on big meshes, and the predictor
    // then runs closer to the HBM roofline).  -1: fuse up to 4 M nodes per slab.
    int opt_fuse_forcing = -1;
    bool fuse_forcing_now() const {
        if (mms_case == CASE_PROG || persist_step) return false;
        return opt_fuse_forcing < 0 ? (size_t)nrows * ncols <= ((size_t)1 << 22) : opt_fuse_forcing != 0;
    }
    bool predict_makes_f1 = false;
    // CASE_PROG: the exact solution as expression programs (tedd_prog.hpp).  [0] node program, [1] cell program
    struct ProgStore {
        uint32_t* d_code = nullptr;
        double* d_consts = nullptr;
        int n_ops = 0, n_regs = 0;
    } prog[2];
    // Programs compiled at run time (backends that have a compiler, tedd_codegen.hpp): -1 use generated kernels and
    // keep the interpreter if they cannot be built (no compiler library, headers not beside the library), 0
    // interpreter only, 1 generated kernels or an error.  Same arrays bit for bit either way.
    HostProg hprog[2];
    int opt_prog_codegen = -1;
    int gen_state = 0;  // 0 not tried for the present programs, 1 loaded, -1 failed (gen_error)
    std::string gen_error;
    bool gen_now() {
        if constexpr (!Backend::kHasCodegen) {
            return false;
        } else {
            if (opt_prog_codegen == 0 || !prog[0].d_code || !prog[1].d_code) return false;
            if (gen_state == 0) {
                try {
                    be.gen_load(hprog[0], hprog[1], react);
                    gen_state = 1;
                } catch (const std::exception& e) {
                    gen_state = -1;
                    gen_error = e.what();
                }
            }
            if (gen_state < 0 && opt_prog_codegen == 1) throw InvalidError(gen_error);
            return gen_state == 1;
        }
    }
    double *d_px = nullptr, *d_py = nullptr, *d_pxq = nullptr, *d_pyq = nullptr;
    int prog_k0 = 0;  // forcing set serving time level 0 of the current call
    int fset[2] = {0, 1};            // which physical array set serves time level 0 / 1
    double fset_time[2] = {NAN, NAN};
    bool forcing_from_cache = false;

    Ctx(const tedd_config& c, const double* x, const double* y) : cfg(c) {
        N = c.N;
        M = c.M;
        if (N < 1 || M < 1) throw InvalidError("mesh needs N >= 1 and M >= 1");
        if (c.row_lo < 0 || c.row_hi > N + 1 || c.row_lo >= c.row_hi) throw InvalidError("bad row range");
        if (c.ghost < 0) throw InvalidError("ghost < 0");
        if (c.model_kind != 1 && c.model_kind != 2) throw InvalidError("model_kind must be 1 or 2");
        if (c.reaction_kind != REACT_CS3 && c.reaction_kind != REACT_REG && c.reaction_kind != REACT_HCS) throw InvalidError("bad reaction_kind");
        if (c.mms_case < CASE_STREAM || c.mms_case > CASE_PROG) throw InvalidError("bad mms_case");
        xg.assign(x, x + N + 1);
        yg.assign(y, y + M + 1);
        for (int i = 1; i <= N; ++i)
            if (!(xg[i] > xg[i - 1])) throw InvalidError("x must be strictly increasing");
        for (int j = 1; j <= M; ++j)
            if (!(yg[j] > yg[j - 1])) throw InvalidError("y must be strictly increasing");
        gl = (c.row_lo > 0) ? std::min(c.ghost, c.row_lo) : 0;
        gr = (c.row_hi < N + 1) ? std::min(c.ghost, N + 1 - c.row_hi) : 0;
        if ((c.row_lo > 0 || c.row_hi < N + 1) && c.ghost < 1) throw InvalidError("a partial slab needs ghost >= 1");
        i0 = c.row_lo - gl;
        nrows = (c.row_hi - c.row_lo) + gl + gr;
        ncols = M + 1;
        ld = ((ncols + 15) / 16) * 16;
        mms_case = c.mms_case;
        react = c.reaction_kind;
        for (int v = 0; v < 5; ++v) cur[v] = nullptr;
        // (a constructor that throws runs no destructor: what was allocated so far is released here)
        try {
            be.init(c.device);
            if constexpr (Backend::kHasArena) {
                // large slabs (their ~32 field buffers take 8 GiB or more): see HipBackend::arena_setup
                const size_t pool = 32 * field_elems() * sizeof(double);
                if (pool >= ((size_t)8 << 30)) be.arena_setup(field_elems() * sizeof(double), kArenaSlots);
            }
            set_model(c.consts, c.model_kind, c.eta, c.mms_params);
            build_geometry();
            build_mms_tables();
            d_red = (double*)be.alloc(64 * sizeof(double));
            d_ints = (int*)be.alloc(16 * sizeof(int));
            be.memset0(d_red, 64 * sizeof(double));
            be.memset0(d_ints, 16 * sizeof(int));
            h_ring = (double*)be.alloc_host_visible((size_t)kRing * 16 * sizeof(double));
            d_ring = (double*)be.device_view(h_ring);
            for (int k = 0; k < kRing; ++k) h_ring[16 * k] = -1.0;
        } catch (...) {
            release_all();
            throw;
        }
    }
    ~Ctx() { release_all(); }
    void release_all() noexcept {
        for (double* p : all_bufs) be.field_free(p);
        for (double*& p : slot) be.field_free(p);
        if constexpr (Backend::kHasArena) be.arena_release();
        be.free(d_geo_rows);
        be.free(d_geo_cols);
        be.free(d_bx);
        be.free(d_by);
        be.free(d_bxq);
        be.free(d_byq);
        for (ProgStore& q : prog) {
            be.free(q.d_code);
            be.free(q.d_consts);
        }
        be.free(d_px);
        be.free(d_py);
        be.free(d_pxq);
        be.free(d_pyq);
        be.free(d_red);
        be.free(d_series);
        be.free(d_partials_all);
        be.free(d_ints);
        be.sync_quiet();  // a prologue launch may still be writing the ring
        be.free_host_visible(h_ring);
        h_ring = nullptr;
        be.shutdown();
    }

    void set_model(const double* consts, int kind, double eta, const double* params) {
        // ModelConsts order: R0 Ea K1 K2 K3 K4 DT Dl_max phi_l gamma_T Kd Sd Dd_max phi_d phi_T r_sp T_ref
        if (kind != 1 && kind != 2) throw InvalidError("model_kind must be 1 or 2");
        model.K1 = consts[2];
        model.K2 = consts[3];
        model.K3 = consts[4];
        model.K4 = consts[5];
        model.DT = consts[6];
        model.Dl_max = consts[7];
        model.phi_l = consts[8];
        model.gamma_T = consts[9];
        model.Kd = consts[10];
        model.Sd = consts[11];
        model.Dd_max = consts[12];
        model.phi_d = consts[13];
        model.phi_T = consts[14];
        model.r_sp = consts[15];
        model.T_shift = (kind == 2) ? consts[16] : 0.0;
        model.eta = eta;
        bool params_changed = false;
        for (int k = 0; k < 8; ++k) {
            const double v = params ? params[k] : 0.0;
            if (std::memcmp(&v, &mp.p[k], sizeof(double)) != 0) params_changed = true;
            mp.p[k] = v;
        }
        if (params_changed && d_bx) build_mms_tables();
        fcp_time[0] = fcp_time[1] = NAN;  // cached forcing depends on the constants
        fset_time[0] = fset_time[1] = NAN;
        last.have = false;
    }

    // ------------------------------------------------------------------ geometry
    void build_geometry() {
        std::vector<double> rows(4 * (size_t)nrows, 0.0), cols(4 * (size_t)ld, 0.0);
        auto hx = [&](int i) { return xg[i] - xg[i - 1]; };
        auto ky = [&](int j) { return yg[j] - yg[j - 1]; };
        for (int l = 0; l < nrows; ++l) {
            const int i = i0 + l;
            if (i >= 1) {
                rows[l] = 1.0 / hx(i);
                rows[3 * (size_t)nrows + l] = hx(i);
            }
            if (i >= 1 && i <= N - 1) {
                const double hm = (hx(i) + hx(i + 1)) * 0.5;
                rows[(size_t)nrows + l] = 1.0 / hm;
                rows[2 * (size_t)nrows + l] = hm;
            }
        }
        for (int j = 0; j <= M; ++j) {
            if (j >= 1) {
                cols[j] = 1.0 / ky(j);
                cols[3 * (size_t)ld + j] = ky(j);
            }
            if (j >= 1 && j <= M - 1) {
                const double km = (ky(j) + ky(j + 1)) * 0.5;
                cols[(size_t)ld + j] = 1.0 / km;
                cols[2 * (size_t)ld + j] = km;
            }
        }
        d_geo_rows = (double*)be.alloc(rows.size() * sizeof(double));
        d_geo_cols = (double*)be.alloc(cols.size() * sizeof(double));
        be.h2d(d_geo_rows, rows.data(), rows.size() * sizeof(double));
        be.h2d(d_geo_cols, cols.data(), cols.size() * sizeof(double));
        double cx = 0, cy = 0;
        for (int i = 1; i <= N - 1; ++i) {
            const double hm = (hx(i) + hx(i + 1)) * 0.5;
            cx = std::max(cx, (1.0 / hx(i) + 1.0 / hx(i + 1)) / hm);
        }
        for (int j = 1; j <= M - 1; ++j) {
            const double km = (ky(j) + ky(j + 1)) * 0.5;
            cy = std::max(cy, (1.0 / ky(j) + 1.0 / ky(j + 1)) / km);
        }
        q_metric = cx + cy;
    }
    GridDev grid() const {
        GridDev g;
        g.nrows = nrows;
        g.ncols = ncols;
        g.ld = ld;
        g.i0 = i0;
        g.N = N;
        g.M = M;
        g.rh = d_geo_rows;
        g.rhm = d_geo_rows + nrows;
        g.hm = d_geo_rows + 2 * (size_t)nrows;
        g.h = d_geo_rows + 3 * (size_t)nrows;
        g.rk = d_geo_cols;
        g.rkm = d_geo_cols + ld;
        g.km = d_geo_cols + 2 * (size_t)ld;
        g.k = d_geo_cols + 3 * (size_t)ld;
        return g;
    }
    void build_mms_tables() {
        be.free(d_bx);
        be.free(d_by);
        be.free(d_bxq);
        be.free(d_byq);
        d_bx = d_by = d_bxq = d_byq = nullptr;
        if (mms_case == CASE_STREAM || mms_case == CASE_PROG) return;
        std::vector<double> nx, qx, ny, qy;
        build_axis_tables(mms_case, mp, xg, i0, nrows, nrows, nx, qx);
        build_axis_tables(mms_case, mp, yg, 0, ncols, ld, ny, qy);
        auto up = [&](const std::vector<double>& v) {
            double* d = (double*)be.alloc(std::max<size_t>(v.size(), 1) * sizeof(double));
            if (!v.empty()) be.h2d(d, v.data(), v.size() * sizeof(double));
            return d;
        };
        d_bx = up(nx);
        d_by = up(ny);
        d_bxq = up(qx);
        d_byq = up(qy);
    }
    MmsTabs tabs() const { return MmsTabs{d_bx, d_by, d_bxq, d_byq, nrows, ld}; }

    // ---- expression programs (CASE_PROG)
    // Every word is checked here, once, so that the interpreter can index registers, constants and
    // outputs without bounds checks.
    void set_mms_program(int which, const uint32_t* code, int n_ops, const double* consts, int n_consts, int n_regs) {
        if (mms_case != CASE_PROG) throw InvalidError("context was not created for an MMS program");
        if (which != 0 && which != 1) throw InvalidError("which must be 0 (node program) or 1 (cell program)");
        if (!code || n_ops < 1 || n_ops > (1 << 20)) throw InvalidError("bad program length");
        if (n_consts < 0 || n_consts > 65536 || (n_consts > 0 && !consts)) throw InvalidError("bad constant table");
        if (n_regs < 3 || n_regs > kProgMaxRegs) throw InvalidError("register count must be in [3, 96]");
        const int n_out = which == 0 ? kProgNodeOutputs : kProgCellOutputs;
        std::vector<char> written(n_out, 0), defined(n_regs, 0);
        defined[0] = defined[1] = defined[2] = 1;
        for (int pc = 0; pc < n_ops; ++pc) {
            const uint32_t w = code[pc];
            const int op = (int)(w & 0xffu), d = (int)((w >> 8) & 0xffu), a = (int)((w >> 16) & 0xffu), b = (int)(w >> 24);
            if (op < 0 || op >= POP_COUNT) throw InvalidError("program: unknown operation");
            if (op == POP_CONST) {
                if ((a | (b << 8)) >= n_consts || d >= n_regs) throw InvalidError("program: constant or register out of range");
                defined[d] = 1;
                continue;
            }
            if (a >= n_regs || !defined[a]) throw InvalidError("program: operand register undefined or out of range");
            if (op == POP_OUT) {
                if (d >= n_out) throw InvalidError("program: output slot out of range");
                written[d] = 1;
                continue;
            }
            const bool binary = (op == POP_ADD || op == POP_SUB || op == POP_MUL || op == POP_DIV || op == POP_POW);
            if (binary && (b >= n_regs || !defined[b])) throw InvalidError("program: operand register undefined or out of range");
            if (d >= n_regs) throw InvalidError("program: destination register out of range");
            defined[d] = 1;
        }
        for (char wv : written)
            if (!wv) throw InvalidError("program: an output is never written");
        ProgStore& q = prog[which];
        be.free(q.d_code);
        be.free(q.d_consts);
        q.d_code = (uint32_t*)be.alloc((size_t)n_ops * sizeof(uint32_t));
        q.d_consts = (double*)be.alloc((size_t)std::max(n_consts, 1) * sizeof(double));
        be.h2d(q.d_code, code, (size_t)n_ops * sizeof(uint32_t));
        if (n_consts > 0) be.h2d(q.d_consts, consts, (size_t)n_consts * sizeof(double));
        q.n_ops = n_ops;
        q.n_regs = n_regs;
        hprog[which].code.assign(code, code + n_ops);
        hprog[which].consts.assign(consts, consts + (n_consts > 0 ? n_consts : 0));
        hprog[which].n_regs = n_regs;
        gen_state = 0;
        if constexpr (Backend::kHasCodegen) be.gen_unload();
        fset_time[0] = fset_time[1] = NAN;
        if (!d_px) build_prog_axes();
    }
    void build_prog_axes() {
        // node coordinates and the 3 Gauss-Legendre points of each interior node's cell
        // [x_{i-1/2}, x_{i+1/2}] (src/prob1base.py:493-598), per axis
        const double gl[3] = {-std::sqrt(3.0 / 5.0), 0.0, std::sqrt(3.0 / 5.0)};
        auto axis = [&](const std::vector<double>& co, int first, int count, int stride, std::vector<double>& nodal,
                        std::vector<double>& quad) {
            const int n = (int)co.size() - 1;
            nodal.assign(stride, 0.0);
            quad.assign((size_t)3 * stride, 0.0);
            for (int l = 0; l < count; ++l) {
                const int i = first + l;
                if (i < 0 || i > n) continue;
                nodal[l] = co[i];
                if (i >= 1 && i <= n - 1) {
                    const double base = 0.5 * (co[i - 1] + co[i]);
                    const double width = 0.5 * ((co[i] - co[i - 1]) + (co[i + 1] - co[i]));
                    for (int a = 0; a < 3; ++a) quad[(size_t)a * stride + l] = base + (gl[a] + 1.0) * 0.5 * width;
                }
            }
        };
        std::vector<double> nx, qx, ny, qy;
        axis(xg, i0, nrows, nrows, nx, qx);
        axis(yg, 0, ncols, ld, ny, qy);
        auto up = [&](const std::vector<double>& v) {
            double* d = (double*)be.alloc(v.size() * sizeof(double));
            be.h2d(d, v.data(), v.size() * sizeof(double));
            return d;
        };
        d_px = up(nx);
        d_py = up(ny);
        d_pxq = up(qx);
        d_pyq = up(qy);
    }
    MmsProg program(int which) const {
        if (!prog[which].d_code) throw InvalidError("the MMS programs have not been set (tedd_set_mms_program)");
        return MmsProg{prog[which].d_code, prog[which].d_consts, prog[which].n_ops};
    }
    ProgAxes prog_axes() const { return ProgAxes{d_px, d_py, d_pxq, d_pyq, nrows, ld}; }
    void prog_exact(double t, double* const out[5], int what) {
        Out5 o;
        for (int v = 0; v < 5; ++v) o.v[v] = out[v];
        bool generated = false;
        if constexpr (Backend::kHasCodegen) {
            if (gen_now()) {
                be.gen_exact(FAM_MISC, grid(), prog_axes(), t, o, what, 0, nrows, ncols);
                generated = true;
            }
        }
        if (!generated) {
            ProgExactFn f{grid(), program(0), prog_axes(), t, o, what};
            be.nodes_prog(FAM_MISC, f, 0, nrows, ncols, prog[0].n_regs);
        }
        for (int v = 0; v < 5; ++v) mark(out[v], full());
    }
    template <int REACT>
    void prog_forcing(double t, double* const out[5]) {
        Out5 o;
        for (int v = 0; v < 5; ++v) o.v[v] = out[v];
        if constexpr (Backend::kHasCodegen) {
            if (gen_now()) {  // (compiled for this context's reaction kind, which REACT is)
                be.gen_forcing(FAM_FORCING, grid(), model, prog_axes(), t, o, 0, nrows, ncols);
                return;
            }
        }
        ProgForcingFn<REACT> f{grid(), model, program(0), program(1), prog_axes(), t, o};
        be.nodes_prog(FAM_FORCING, f, 0, nrows, ncols, std::max(prog[0].n_regs, prog[1].n_regs));
    }

    ForcingSrc forcing_src(double t, int which) {
        ForcingSrc s;
        s.tm = mms_time_fill(mms_case, model, mp, t);
        for (int v = 0; v < 5; ++v) s.f[v] = nullptr;
        if (forcing_from_cache) {
            for (int v = 0; v < 5; ++v) s.f[v] = slot[(fset[which] ? SF10 : SF00) + v];
        } else if (mms_case == CASE_PROG) {
            // the array-fed kernels read the set of this time level, evaluated by the programs on first use
            int k = -1;
            for (int q = 0; q < 2; ++q)
                if (fset_time[q] == t) k = q;
            if (k < 0) {
                k = (which == 0) ? 0 : 1 - prog_k0;  // (keep the set time level 0 of this call reads)
                fill_forcing_set(t, k);
            }
            if (which == 0) prog_k0 = k;
            for (int v = 0; v < 5; ++v) s.f[v] = slot[(k ? SF10 : SF00) + v];
        } else if (mms_case == CASE_STREAM) {
            if (!stream_forcing_set[which]) throw InvalidError("streamed forcing not set for this time level");
            for (int v = 0; v < 5; ++v) s.f[v] = slot[(which ? SF10 : SF00) + v];
        }
        return s;
    }

    // ------------------------------------------------------------------ buffers
    size_t field_elems() const { return (size_t)nrows * ld; }
    double* new_field() {
        double* p = (double*)be.field_alloc(field_elems() * sizeof(double));
        be.memset0(p, field_elems() * sizeof(double));
        return p;
    }
    double* B(int id) {
        if (!slot[id]) slot[id] = new_field();
        return slot[id];
    }
    double* acquire() {
        double* p;
        if (!free_bufs.empty()) {
            p = free_bufs.back();
            free_bufs.pop_back();
        } else {
            p = new_field();
            all_bufs.push_back(p);
        }
        valid.erase(p);
        return p;
    }
    void release(double* p) {
        if (!p) return;
        for (int v = 0; v < 5; ++v)
            if (cur[v] == p) return;
        for (int v = 0; v < 3; ++v)
            if (prev[v] == p) return;
        if (std::find(free_bufs.begin(), free_bufs.end(), p) == free_bufs.end()) free_bufs.push_back(p);
    }
    // a buffer commit_state released (or that is still the warm start's) is held for tedd_get_residuals
    void keep_until_next_step(double* p) {
        if (!p) return;
        for (int v = 0; v < 5; ++v)
            if (cur[v] == p) return;
        for (int v = 0; v < 3; ++v)
            if (prev[v] == p) return;  // (kept anyway, until the next commit)
        auto it = std::find(free_bufs.begin(), free_bufs.end(), p);
        if (it != free_bufs.end()) free_bufs.erase(it);
        if (std::find(retained.begin(), retained.end(), p) == retained.end()) retained.push_back(p);
    }
    void release_retained() {
        std::vector<double*> r;
        r.swap(retained);
        for (double* p : r) release(p);
    }
    Range full() const { return Range{0, nrows}; }
    Range owned() const { return Range{gl, nrows - gr}; }
    Range shrink(Range r) const {
        // one stencil application: rows next to a non-physical slab edge lose validity
        Range o = r;
        if (i0 + r.lo > 0) o.lo = r.lo + 1;
        if (i0 + r.hi - 1 < N) o.hi = r.hi - 1;
        return o;
    }
    Range vof(const double* p) const {
        auto it = valid.find(p);
        return it == valid.end() ? Range{0, 0} : it->second;
    }
    struct Use {
        const double* p;
        bool stencil;
    };
    Range need(std::initializer_list<Use> uses) const {
        Range r = full();
        for (const Use& u : uses) {
            if (!u.p) continue;
            Range v = vof(u.p);
            if (u.stencil) v = shrink(v);
            r.lo = std::max(r.lo, v.lo);
            r.hi = std::min(r.hi, v.hi);
        }
        if (r.hi < r.lo) r.hi = r.lo;
        return r;
    }
    void mark(double* p, Range r) { valid[p] = r; }

    void upload_field(double* dst, const double* src) {
        be.h2d_2d(dst, ld, src, ncols, ncols, nrows);
        mark(dst, full());
    }
    void download_field(double* dst, const double* src) { be.d2h_2d(dst, ncols, src, ld, ncols, nrows); }

    void require_state() const {
        if (!cur[0]) throw InvalidError("state not set");
    }
    void drop_prev() {
        for (int v = 0; v < 3; ++v) {
            double* p = prev[v];
            prev[v] = nullptr;
            if (p && last.have && p == last.lazy_T0) {
                // (tedd_get_residuals may still remake Y_T from it: held until the next step instead)
                if (std::find(retained.begin(), retained.end(), p) == retained.end()) retained.push_back(p);
            } else {
                release(p);
            }
        }
        prev_dt = 0.0;
    }
    void reset_history() {
        fcp_time[0] = fcp_time[1] = NAN;
        last.have = false;
        release_retained();
        drop_prev();
        chain_start = next_seq;  // records of steps before this point say nothing about the new state
        for (int v = 0; v < 3; ++v) {
            cold_until[v] = 0;
            cold_spell[v] = kColdSpell;
        }
    }
    void ensure_state_buffers() {
        for (int v = 0; v < 5; ++v)
            if (!cur[v]) cur[v] = acquire();
    }
    void set_state(const double* const h[5]) {
        ghost_pending = false;
        ensure_state_buffers();
        for (int v = 0; v < 5; ++v) upload_field(cur[v], h[v]);
        reset_history();
    }
    void get_state(double* const h[5]) {
        require_state();
        refuse_pending_ghosts("tedd_get_state");
        be.d2h_fields(h, cur, 5, ld, ncols, nrows);
    }
    // Install nxt as the state; buffers of the old state that are not reused go back to the pool.
    void commit_state(double* const nxt[5], bool keep_old_as_prev = false, double dt = 0.0) {
        Range o = owned();
        for (int v = 0; v < 5; ++v) {
            Range r = vof(nxt[v]);
            if (r.lo > o.lo || r.hi < o.hi)
                throw InvalidError("ghost depth too small for this step (stencil applications exceeded the ghost rows)");
        }
        double* old[5];
        for (int v = 0; v < 5; ++v) {
            old[v] = cur[v];
            cur[v] = nxt[v];
        }
        double* gone[3] = {prev[0], prev[1], prev[2]};
        for (int v = 0; v < 3; ++v) prev[v] = nullptr;
        if (keep_old_as_prev && opt_warm && !persist_step) {
            for (int v = 0; v < 3; ++v) prev[v] = (old[VAR_T + v] != cur[VAR_T + v]) ? old[VAR_T + v] : nullptr;
            prev_dt = dt;
        } else {
            prev_dt = 0.0;
        }
        for (int v = 0; v < 3; ++v) release(gone[v]);
        for (int v = 0; v < 5; ++v) release(old[v]);
    }
    // The caller refreshed the ghost rows of the current state (halo exchange).
    void mark_state_exchanged() {
        require_state();
        for (int v = 0; v < 5; ++v) mark(cur[v], full());
        ghost_pending = false;
    }
    // Ghost rows in flight (tedd_ghosts_pending): the driver posted their transfer and did not wait.  The first
    // launch of the next predictor-corrector step runs on the rows that do not need them, then the hook completes
    // the transfer, then the launch runs on the remaining bands.
    tedd_ghost_fn ghost_hook = nullptr;
    void* ghost_user = nullptr;
    bool ghost_pending = false;
    void complete_pending_ghosts() {
        if (!ghost_pending) return;
        if (!ghost_hook) throw InvalidError("the ghost rows of the state are in flight: refresh them (tedd_ghosts_refreshed) or set a ghost hook");
        if (ghost_hook(ghost_user) != 0) throw std::runtime_error("the ghost hook failed");
        mark_state_exchanged();
    }
    void refuse_pending_ghosts(const char* what) const {
        if (ghost_pending)
            throw InvalidError(std::string(what) + ": the ghost rows of the state are in flight (tedd_ghosts_pending); complete "
                               "the exchange and call tedd_ghosts_refreshed first");
    }

    // ------------------------------------------------------------------ dispatch helpers
    template <class Fn>
    void dispatch(Fn&& fn, int case_override = -100) {
        // (a program case runs the array-fed kernels; the operations that evaluate the exact solution
        // or the forcing check mms_case themselves)
        int c = (case_override == -100) ? mms_case : case_override;
        if (c == CASE_PROG) c = CASE_STREAM;
        const int r = react;
#define TEDD_D(CC, RR)             \
    if (c == CC && r == RR) {      \
        fn.template run<CC, RR>(); \
        return;                    \
    }
        TEDD_D(CASE_STREAM, REACT_CS3)
        TEDD_D(CASE_STREAM, REACT_REG)
        TEDD_D(CASE_STREAM, REACT_HCS)
        TEDD_D(CASE_POL, REACT_CS3)
        TEDD_D(CASE_POL, REACT_REG)
        TEDD_D(CASE_POL, REACT_HCS)
        TEDD_D(CASE_EXPSIN, REACT_CS3)
        TEDD_D(CASE_EXPSIN, REACT_REG)
        TEDD_D(CASE_EXPSIN, REACT_HCS)
        TEDD_D(CASE_PEAKS, REACT_CS3)
        TEDD_D(CASE_PEAKS, REACT_REG)
        TEDD_D(CASE_PEAKS, REACT_HCS)
        TEDD_D(CASE_NFS, REACT_CS3)
        TEDD_D(CASE_NFS, REACT_REG)
        TEDD_D(CASE_NFS, REACT_HCS)
#undef TEDD_D
        throw InvalidError("unsupported case/reaction combination");
    }

    Fields5 fields(double* const p[5]) const {
        Fields5 f;
        for (int v = 0; v < 5; ++v) f.v[v] = p[v];
        return f;
    }
    struct Tmp5 {  // five pool buffers handed back on scope exit
        Ctx* c;
        double* p[5];
        explicit Tmp5(Ctx* c_) : c(c_) {
            for (int v = 0; v < 5; ++v) p[v] = c->acquire();
        }
        ~Tmp5() {
            for (int v = 0; v < 5; ++v) c->release(p[v]);
        }
    };

    // ------------------------------------------------------------------ field evaluation / Euler
    struct FieldOp {
        Ctx* c;
        double t, dt;
        bool euler;
        double* out[5];
        template <int CASE, int REACT>
        void run() {
            Ctx& C = *c;
            Range r = C.need({{C.cur[0], true}, {C.cur[1], true}, {C.cur[2], true}, {C.cur[3], true}, {C.cur[4], false}});
            Out5 o;
            for (int v = 0; v < 5; ++v) o.v[v] = out[v];
            if (euler) {
                FieldFn<CASE, REACT, true> f{C.grid(), C.model, C.tabs(), C.forcing_src(t, 0), C.fields(C.cur), o, dt};
                C.be.nodes(FAM_FIELD, f, r.lo, r.hi, C.ncols);
            } else {
                FieldFn<CASE, REACT, false> f{C.grid(), C.model, C.tabs(), C.forcing_src(t, 0), C.fields(C.cur), o, dt};
                C.be.nodes(FAM_FIELD, f, r.lo, r.hi, C.ncols);
            }
            for (int v = 0; v < 5; ++v) C.mark(out[v], r);
        }
    };
    void eval_field(double t, double* const h[5]) {
        require_state();
        refuse_pending_ghosts("tedd_eval_field");
        Tmp5 tmp(this);
        FieldOp op{this, t, 0.0, false, {tmp.p[0], tmp.p[1], tmp.p[2], tmp.p[3], tmp.p[4]}};
        dispatch(op);
        for (int v = 0; v < 5; ++v)
            if (h[v]) download_field(h[v], op.out[v]);
        be.sync();
    }
    void step_feuler(double t0, double dt) {
        require_state();
        refuse_pending_ghosts("tedd_step_feuler");
        release_retained();
        double* nxt[5];
        for (int v = 0; v < 5; ++v) nxt[v] = acquire();
        FieldOp op{this, t0, dt, true, {nxt[0], nxt[1], nxt[2], nxt[3], nxt[4]}};
        dispatch(op);
        commit_state(nxt);
        last.have = false;
        stream_forcing_set[0] = stream_forcing_set[1] = false;
    }

    // ------------------------------------------------------------------ hooks
    struct ExactOp {
        Ctx* c;
        double t;
        double* out[5];
        int what = 0;
        template <int CASE, int REACT>
        void run() {
            Ctx& C = *c;
            if (C.mms_case == CASE_PROG) {
                C.prog_exact(t, out, what);
                return;
            }
            if (CASE == CASE_STREAM) throw InvalidError("no closed-form exact solution for streamed forcing");
            Out5 o;
            for (int v = 0; v < 5; ++v) o.v[v] = out[v];
            ExactFn<(CASE == CASE_STREAM ? CASE_POL : CASE)> f{C.grid(), C.model, C.tabs(),
                                                               mms_time_fill(C.mms_case, C.model, C.mp, t), o, what};
            C.be.nodes(FAM_MISC, f, 0, C.nrows, C.ncols);
            for (int v = 0; v < 5; ++v) C.mark(out[v], C.full());
        }
    };
    void eval_exact(double t, double* const h[5], int what = 0) {
        if (what < 0 || what > 4) throw InvalidError("what must be 0..4");
        Tmp5 tmp(this);
        ExactOp op{this, t, {tmp.p[0], tmp.p[1], tmp.p[2], tmp.p[3], tmp.p[4]}, what};
        dispatch(op);
        for (int v = 0; v < 5; ++v)
            if (h[v]) download_field(h[v], op.out[v]);
        be.sync();
    }
    void set_state_exact(double t) {
        ghost_pending = false;
        ensure_state_buffers();
        ExactOp op{this, t, {cur[0], cur[1], cur[2], cur[3], cur[4]}};
        dispatch(op);
        reset_history();
    }
    struct ForcingOp {
        Ctx* c;
        double t;
        double* out[5];
        template <int CASE, int REACT>
        void run() {
            Ctx& C = *c;
            if (C.mms_case == CASE_PROG) {
                C.template prog_forcing<REACT>(t, out);
                return;
            }
            if (CASE == CASE_STREAM) throw InvalidError("forcing is streamed: nothing to evaluate");
            Out5 o;
            for (int v = 0; v < 5; ++v) o.v[v] = out[v];
            ForcingSrc src;
            src.tm = mms_time_fill(C.mms_case, C.model, C.mp, t);
            for (int v = 0; v < 5; ++v) src.f[v] = nullptr;
            ForcingFn<(CASE == CASE_STREAM ? CASE_POL : CASE), REACT> f{C.grid(), C.model, C.tabs(), src, o};
            C.be.forcing(FAM_FORCING, f, 0, C.nrows, C.ncols);
        }
    };
    void eval_forcing(double t, double* const h[5]) {
        Tmp5 tmp(this);
        ForcingOp op{this, t, {tmp.p[0], tmp.p[1], tmp.p[2], tmp.p[3], tmp.p[4]}};
        dispatch(op);
        for (int v = 0; v < 5; ++v)
            if (h[v]) download_field(h[v], op.out[v]);
        be.sync();
    }
    void eval_jacobian(int opid, double* const h[5]) {
        require_state();
        refuse_pending_ghosts("tedd_eval_jacobian");
        if (opid < 0 || opid > 5) throw InvalidError("jacobian op must be 0..5");
        Tmp5 tmp(this);
        Out5 o;
        for (int v = 0; v < 5; ++v) o.v[v] = tmp.p[v];
        Range r = need({{cur[0], true}, {cur[1], true}, {cur[2], true}, {cur[3], true}, {cur[4], false}});
        if (react == REACT_HCS) {
            JacobianFn<REACT_HCS> f{grid(), model, fields(cur), o, opid};
            be.nodes(FAM_MISC, f, r.lo, r.hi, ncols);
        } else if (react == REACT_REG) {
            JacobianFn<REACT_REG> f{grid(), model, fields(cur), o, opid};
            be.nodes(FAM_MISC, f, r.lo, r.hi, ncols);
        } else {
            JacobianFn<REACT_CS3> f{grid(), model, fields(cur), o, opid};
            be.nodes(FAM_MISC, f, r.lo, r.hi, ncols);
        }
        for (int v = 0; v < 5; ++v)
            if (h[v]) download_field(h[v], o.v[v]);
        be.sync();
    }
    void set_forcing(int which, const double* const h[5]) {
        if (mms_case != CASE_STREAM) throw InvalidError("context was created with a closed-form MMS case");
        if (which != 0 && which != 1) throw InvalidError("which must be 0 or 1");
        for (int v = 0; v < 5; ++v) upload_field(B((which ? SF10 : SF00) + v), h[v]);
        stream_forcing_set[which] = true;
    }

    // ------------------------------------------------------------------ error norms
    struct NormOp {
        Ctx* c;
        double t;
        double* dev_out;
        bool noalloc = false;
        double* partials_out = nullptr;  // leave the partial sums there, no final pass (advance finishes them all at its end)
        template <int CASE, int REACT>
        void run() {
            Ctx& C = *c;
            if (C.mms_case == CASE_PROG) {  // exact fields from the program, then the array-fed norms
                double* ex[5];
                for (int v = 0; v < 5; ++v) ex[v] = C.B(EX0 + v);
                C.prog_exact(t, ex, 0);
            }
            ErrNormFn<CASE> f;
            f.g = C.grid();
            f.m = C.model;
            f.tb = C.tabs();
            f.tm = mms_time_fill(C.mms_case, C.model, C.mp, t);
            f.S = C.fields(C.cur);
            for (int v = 0; v < 5; ++v) f.ex[v] = (CASE == CASE_STREAM) ? C.slot[EX0 + v] : nullptr;
            Range o = C.owned();
            Range r = C.need({{C.cur[1], true}, {C.cur[2], true}, {C.cur[3], true}});
            if ((C.gl > 0 || C.gr > 0) && (r.lo > o.lo || r.hi < o.hi))
                throw InvalidError("ghost rows are stale: exchange them before taking norms");
            if (noalloc) {
                C.be.nodes_sum_noalloc(f, o.lo, o.hi, C.ncols, dev_out);
            } else if (partials_out) {
                if constexpr (Backend::kHasPersist) C.be.nodes_sum(FAM_NORMS, f, o.lo, o.hi, C.ncols, dev_out, partials_out);
            } else {
                C.be.nodes_sum(FAM_NORMS, f, o.lo, o.hi, C.ncols, dev_out);
            }
        }
    };
    // Largest slab tedd_error_norms_noalloc accepts (nodes, ghost rows included): 2^26, a quarter of the
    // 16384^2 mesh.  The limit is a promise about the fixed scratch array, not a property of the kernel.
    static constexpr long long kNoallocMaxNodes = 1LL << 26;
    void error_norms_noalloc(double t, double* dev_out) {
        require_state();
        refuse_pending_ghosts("tedd_error_norms_noalloc");
        if (!dev_out) throw InvalidError("out is null");
        if (mms_case == CASE_STREAM || mms_case == CASE_PROG)
            throw InvalidError("tedd_error_norms_noalloc needs a closed-form MMS case (the others keep the exact "
                               "fields in buffers that would have to be allocated)");
        if ((long long)nrows * ncols > kNoallocMaxNodes)
            throw InvalidError("tedd_error_norms_noalloc: slab larger than 2^26 nodes");
        NormOp op{this, t, dev_out, true};
        dispatch(op);
    }
    void error_norms_async(double t, double* dev_out) {
        require_state();
        refuse_pending_ghosts("tedd_error_norms");
        NormOp op{this, t, dev_out};
        dispatch(op);
    }
    void error_norms(double t, const double* const exact[5], double* out8) {
        require_state();
        if (mms_case == CASE_STREAM) {
            if (!exact) throw InvalidError("exact fields required when forcing is streamed");
            for (int v = 0; v < 5; ++v) upload_field(B(EX0 + v), exact[v]);
        }
        error_norms_async(t, d_red + 32);
        be.d2h(out8, d_red + 32, 8 * sizeof(double));
        be.sync();
    }

    // ------------------------------------------------------------------ the predictor-corrector step
    int auto_sweeps(double q, double tol, int cap) const {
        // Jacobi on rows (2 + q_ij) d_ij - (off-diagonals summing to <= q_ij): contraction <= q/(2+q)
        if (!(q > 0)) return 2;
        const double rho = q / (2.0 + q);
        const double s = std::ceil(std::log(tol) / std::log(rho)) + 1.0;
        if (!(s < (double)cap)) return cap;
        return std::max(2, (int)s);
    }

    struct StepParams {
        double t0, dt;
        int num_pc, num_newton, cs_iters;
        double cs_rtol;
        int sweeps[3];
        bool warm[3] = {false, false, false};  // start the first Newton solve of v from the last increment
        double gscale = 0.0;                   // dt / dt of the step before
    };
    struct PcOp {
        Ctx* c;
        StepParams p;
        template <int CASE, int REACT>
        void run() {
            c->template step_pc_impl<CASE, REACT>(p);
        }
    };

    // The record the prologue of step `seq + 1` published for step `seq`: waits (bounded) until it is there.
    // Returns false if the record cannot be had (never published, or overwritten).
    bool lagged_record(long long seq, double* upd, double* inc, double* first, double* umax) {
        if (seq < chain_start || seq < 0 || seq + kRing <= next_seq || seq + 1 >= next_seq) return false;
        const StepMeta& ms = meta[seq % kRing];
        if (ms.seq != seq || !(ms.warm[0] || ms.warm[1] || ms.warm[2])) return false;  // (never published)
        volatile double* slot = h_ring + 16 * (seq % kRing);
        const double want = (double)seq;
        for (long long spin = 0; slot[0] != want; ++spin) {
            if (spin == (1LL << 22)) be.sync();  // (the prologue that publishes it has long been enqueued)
            if (spin > (1LL << 22) + 16) return false;
        }
        for (int v = 0; v < 3; ++v) {
            upd[v] = slot[1 + v];
            inc[v] = slot[4 + v];
            first[v] = slot[7 + v];
            umax[v] = slot[10 + v];
        }
        return slot[0] == want;
    }

    // Sweep counts of the three Newton solves.  A count the caller fixes is used as it is, from a zero
    // iterate.  Otherwise: rows of (2 I - dt J) are diagonally dominant with contraction rho <= q / (2 + q)
    // per Jacobi sweep; from a zero iterate ceil(log tol / log rho) + 1 sweeps bring the last update below
    // tol x increment.  With the state of the step before at hand the iteration starts from that step's
    // increment instead (the increments of consecutive steps differ by O(dt)), and the count follows from
    // the first update max|delta^1 - delta^0| / max|delta| that the step kLag before this one recorded for
    // its own warm start (with a factor 4 of margin), never more than the cold count plus one.  The record
    // also carries the contraction actually seen, which replaces the bound where it is worse.  A warm start
    // reads two more arrays (the lagged unknown and the one before): it is used where it saves at least
    // three sweeps or a launch of the chain; otherwise the system goes back to zero iterates for a while.
    void choose_sweeps(StepParams& p, const int32_t* sweeps_in, double jtol, int jmax, StepMeta& mt) {
        const double dcoef[3] = {model.DT, model.Dl_max, model.Dd_max};
        const bool can_warm = opt_warm && !persist_step && p.num_pc > 0 && p.num_newton > 0 && prev_dt > 0.0 &&
                              next_seq > chain_start;
        double upd[3], inc[3], first[3], umax[3];
        bool have_rec = false;
        bool asked = false;
        p.gscale = can_warm ? p.dt / prev_dt : 0.0;
        for (int v = 0; v < 3; ++v) {
            const int s_in = sweeps_in ? sweeps_in[v] : 0;
            mt.warm[v] = false;
            mt.tol[v] = 0.0;
            mt.rho[v] = 0.0;
            if (s_in > 0) {
                p.sweeps[v] = std::max(2, (int)s_in);
                mt.sweeps[v] = p.sweeps[v];
                continue;
            }
            const double q = p.dt * std::fabs(dcoef[v]) * q_metric;
            double rho = (q > 0) ? q / (2.0 + q) : 0.0;
            const int cold = auto_sweeps(q, jtol, jmax);
            int s = cold;
            bool warm = false;
            mt.tol[v] = jtol;
            if (can_warm && prev[v] && next_seq >= cold_until[v]) {
                warm = true;
                // until a record says how good the start is: the cold count, all of it performed on the warm
                // iterate (one sweep more than a cold chain runs after b = delta^1: one more ghost row per system
                // on a slab, which is the "+ 3" of the ghost rule for library-chosen sweeps).  The same on a slab
                // and on the whole mesh, so that a decomposed run repeats the single-domain iteration.
                s = std::min(cold, jmax);
                if (!asked) {
                    have_rec = lagged_record(next_seq - kLag, upd, inc, first, umax);
                    asked = true;
                }
                const StepMeta& m2 = meta[(next_seq - kLag + 2 * kRing) % kRing];
                if (have_rec && m2.seq == next_seq - kLag && m2.warm[v] && inc[v] > 0.0 && first[v] == first[v] &&
                    upd[v] == upd[v]) {
                    // contraction seen then, where the record can tell: a last update within a factor 100
                    // of the rounding of the iterate (2^-53 x increment) is noise, not convergence
                    if (m2.sweeps[v] >= 3 && first[v] > 0.0 && upd[v] > 1e-14 * inc[v]) {
                        const double seen = std::pow(upd[v] / first[v], 1.0 / (m2.sweeps[v] - 1));
                        if (seen > rho && seen < 1.0) rho = seen;
                    }
                    // sweeps after the first so that the last update is below tol / 4 (the start of this
                    // step is judged by that of an earlier one: the factor covers the difference)
                    const double r0 = 4.0 * first[v] / inc[v];
                    // (an update below opt_floor x max|u| needs no further sweep: StepRecordArgs::floor)
                    double tol_eff = jtol;
                    if (umax[v] == umax[v] && opt_floor * umax[v] > jtol * inc[v]) tol_eff = opt_floor * umax[v] / inc[v];
                    if (r0 <= tol_eff || rho <= 0.0) {
                        s = 2;
                    } else if (rho < 1.0) {
                        const double want = 1.0 + std::max(std::ceil(std::log(tol_eff / r0) / std::log(rho)), 0.0);
                        if (want < (double)s) s = (int)want;
                    }
                    if (s < 2) s = 2;
                    // worth the two extra arrays?
                    const int hmax = std::min(opt_strip_hmax > 0 ? opt_strip_hmax : 10, 16);
                    auto launches = [&](int sweeps_in_chain) { return std::max(1, (sweeps_in_chain + hmax - 1) / hmax); };
                    if (!((cold - 1) - s >= 3 || launches(s) < launches(cold - 1))) {
                        cold_until[v] = next_seq + cold_spell[v];
                        cold_spell[v] = std::min(2 * cold_spell[v], (int)kColdSpellMax);
                        warm = false;
                        s = cold;
                    }
                }
            }
            p.sweeps[v] = s;
            p.warm[v] = warm;
            mt.sweeps[v] = s;
            mt.warm[v] = warm;
            mt.rho[v] = rho;
        }
    }

    void step_pc(double t0, double dt, int num_pc, int num_newton, int cs_iters, double cs_rtol,
                 const int32_t* sweeps_in, double jtol, int jmax) {
        require_state();
        if (!(dt > 0)) throw AssertError("dt must be > 0");
        if (num_pc < 0 || num_newton < 0) throw InvalidError("negative iteration counts");
        if (cs_iters < 0 || cs_iters > 4096) throw InvalidError("cs_newton_iterations must be in [0, 4096]");
        if (!(jtol > 0)) jtol = 1e-13;
        if (jmax < 2) jmax = 2;
        StepParams p{t0, dt, num_pc, num_newton, cs_iters, cs_rtol, {0, 0, 0}};
        StepMeta mt;
        mt.seq = next_seq;
        const int pgroups = persist_groups(num_pc, num_newton, cs_iters);
        struct PersistGuard {
            bool& flag;
            ~PersistGuard() { flag = false; }
        } persist_guard{persist_step};
        persist_step = pgroups > 0;
        if (persist_step) {
            drop_prev();  // (recorded steps start their solves from zero: no history is kept or read)
            be.rec_begin(react, pgroups);
        }
        choose_sweeps(p, sweeps_in, jtol, jmax, mt);
        // prologue: fold and publish the record of the step before, clear the records of this one
        {
            StepRecordArgs ra;
            ra.red = d_red;
            const bool have_before = next_seq > chain_start && record_unfolded;
            const StepMeta& mb = meta[(next_seq - 1 + kRing) % kRing];
            ra.fold = (have_before && record_unjudged) ? 1 : 0;
            // (only the record of a warm-started solve is ever read back: choose_sweeps)
            const bool any_warm = mb.warm[0] || mb.warm[1] || mb.warm[2];
            ra.ring_slot = (have_before && any_warm) ? d_ring + 16 * ((next_seq - 1) % kRing) : nullptr;
            ra.seq = (double)(next_seq - 1);
            for (int v = 0; v < 3; ++v) ra.tol[v] = have_before ? mb.tol[v] : 0.0;
            ra.floor = opt_floor;
            ra.clear = 1;
            ra.cs_iters = (react == REACT_REG) ? cs_iters : 0;
            // (carried out by the predictor launch where that is a plain node launch, else by a launch of its own)
            pending_prologue = ra;
            have_pending_prologue = true;
        }
        meta[next_seq % kRing] = mt;
        next_seq += 1;
        record_unfolded = (num_pc > 0 && num_newton > 0);
        record_unjudged = record_unfolded;
        for (int v = 0; v < 3; ++v) judged_since_check = judged_since_check || (record_unfolded && mt.tol[v] > 0.0);
        try {
            step_pc_dispatch(p);
            if (have_pending_prologue) run_pending_prologue();  // (a step without a predictor launch)
        } catch (...) {
            have_pending_prologue = false;
            // nothing of a failed step may serve a later one
            bT_ready = false;
            predict_makes_f1 = false;
            fset_time[0] = fset_time[1] = NAN;
            fcp_time[0] = fcp_time[1] = NAN;
            record_unfolded = record_unjudged = false;
            chain_start = next_seq;
            drop_prev();
            throw;
        }
    }
    StepRecordArgs pending_prologue;
    bool have_pending_prologue = false;
    void run_pending_prologue() {
        if (!have_pending_prologue) return;
        have_pending_prologue = false;
        be.step_prologue(pending_prologue);
    }
    // the predictor launch of a step, carrying the step's prologue where the backend can fold it in
    template <class F>
    void launch_predictor(const F& f, int r0, int r1) {
        if (have_pending_prologue) {
            have_pending_prologue = false;
            be.nodes_with_prologue(FAM_PREDICT, f, r0, r1, ncols, pending_prologue);
        } else {
            be.nodes(FAM_PREDICT, f, r0, r1, ncols);
        }
    }
    void step_pc_dispatch(const StepParams& p) {
        const double t0 = p.t0, dt = p.dt;
        PcOp op{this, p};
        if ((opt_cache_forcing || mms_case == CASE_PROG) && mms_case != CASE_STREAM) {
            // evaluate the closed-form forcing once per time level, then run the array-fed kernels
            int k0 = -1;
            for (int k = 0; k < 2; ++k)
                if (fset_time[k] == t0) k0 = k;
            if (k0 < 0) {
                k0 = 0;
                fill_forcing_set(t0, k0);
            }
            const int k1 = 1 - k0;
            if (fuse_forcing_now()) {
                predict_makes_f1 = true;  // the predictor launch evaluates the t1 set (step_pc_impl)
                for (int v = 0; v < 5; ++v) B((k1 ? SF10 : SF00) + v);
                fset_time[k1] = t0 + dt;
            } else if (fset_time[k1] != t0 + dt) {  // (already there after prefetch_forcing)
                fill_forcing_set(t0 + dt, k1);
            }
            fset[0] = k0;
            fset[1] = k1;
            forcing_from_cache = true;
            try {
                dispatch(op, CASE_STREAM);
            } catch (...) {
                forcing_from_cache = false;
                throw;
            }
            forcing_from_cache = false;
            return;
        }
        dispatch(op);
    }
    // Fold the record of the last step into the sticky ratios now (idempotent: a maximum).
    void fold_last_record() {
        if (!record_unjudged || next_seq <= chain_start) return;
        StepRecordArgs ra;
        ra.red = d_red;
        ra.ring_slot = nullptr;
        ra.seq = 0.0;
        const StepMeta& mb = meta[(next_seq - 1 + kRing) % kRing];
        for (int v = 0; v < 3; ++v) ra.tol[v] = mb.tol[v];
        ra.floor = opt_floor;
        ra.fold = 1;
        ra.clear = 0;
        ra.cs_iters = 0;
        be.step_prologue(ra);
        record_unjudged = false;
    }
    // The reference solves its Newton systems exactly (src/prob1base.py:2103,2130); here every solve whose
    // sweep count the library chose is judged after the fact: last update <= tol x increment.  Throws
    // NotConverged if some step since the last check missed that by more than a factor 10, and clears the
    // sticky record either way.
    void check_converged() {
        if (!judged_since_check) return;
        fold_last_record();
        double worst[3];
        be.d2h(worst, d_red + 16, 3 * sizeof(double));
        be.sync();
        be.memset0(d_red + 16, 3 * sizeof(double));
        judged_since_check = false;
        static const char* nm[3] = {"T", "cl", "cd"};
        for (int v = 0; v < 3; ++v) {
            if (worst[v] > 10.0 || worst[v] != worst[v]) {
                char buf[256];
                std::snprintf(buf, sizeof buf,
                              "the Jacobi solve of the %s system missed its tolerance: last update / (tol x increment) "
                              "reached %.3g in some step; fix the sweep counts (jacobi_sweeps) or raise jacobi_max",
                              nm[v], worst[v]);
                throw NotConverged(buf);
            }
        }
    }
    // Evaluate the forcing arrays of a coming time level now, into the set that does not hold
    // `keep_t` (the level the next step starts from).  The forcing does not depend on the state, so a
    // caller that exchanges ghost rows between steps can overlap this launch with the transfer.
    // No effect when the forcing is streamed, evaluated inside the step kernels, or made by the
    // predictor launch.
    void prefetch_forcing(double t, double keep_t) {
        if (mms_case == CASE_STREAM) return;
        if (!(opt_cache_forcing || mms_case == CASE_PROG) || fuse_forcing_now()) return;
        if (fset_time[0] == t || fset_time[1] == t) return;
        const int k = (fset_time[0] == keep_t) ? 1 : 0;
        fill_forcing_set(t, k);
    }
    void fill_forcing_set(double t, int k) {
        ForcingOp op{this, t, {B((k ? SF10 : SF00)), B((k ? SF10 : SF00) + 1), B((k ? SF10 : SF00) + 2),
                               B((k ? SF10 : SF00) + 3), B((k ? SF10 : SF00) + 4)}};
        dispatch(op);
        for (int v = 0; v < 5; ++v) mark(op.out[v], full());
        fset_time[k] = t;
    }

    // valid range of a launch that reads `uses` with a halo of `depth` rows (+1 where stencil)
    Range need_depth(std::initializer_list<Use> uses, int depth) const {
        Range r = full();
        for (const Use& u : uses) {
            if (!u.p) continue;
            Range v = vof(u.p);
            for (int d = 0; d < depth + (u.stencil ? 1 : 0); ++d) v = shrink(v);
            r.lo = std::max(r.lo, v.lo);
            r.hi = std::min(r.hi, v.hi);
        }
        if (r.hi < r.lo) r.hi = r.lo;
        return r;
    }

    // options
    int opt_strip = 1;      // 1: register-resident solve chain (backends that have it); 0: one launch per sweep
    int opt_strip_hmax = 0;         // sweeps fused per launch (0 = automatic)
    int opt_strip_shape = -1;       // tile shape of the chain: -1 by mesh size and system, 0: 64 rows (4 waves x 16), 1: 32 rows, 2: 48 rows, 3: 16 rows (cl, cd), 4: 64 rows (8 waves x 8, T)
    // single-launch cl / cd chains evaluate their right-hand side themselves: -1 on slabs of at most 4 M nodes
    // (a launch and five to seven array passes fewer per solve; on large slabs the arithmetic repeated in the
    // tile halos costs more than the passes: 16384^2, rhs + solve launches 6.75 / 8.66 ms against 6.79 / 9.47 fused)
    int opt_fuse_rhs = -1;
    int opt_exact_faces = 0;  // (test hook, see StripArgs::xface)
    // the last launch of a cd chain that evaluates its right-hand side also carries the record pass of the cs
    // corrector (RegHCsTriple, whole-mesh slabs): one launch fewer per step, same results
    int opt_fuse_corrector = 1;
    bool fuse_rhs_now() const {
        return opt_fuse_rhs < 0 ? (size_t)nrows * ncols <= ((size_t)1 << 22) : opt_fuse_rhs != 0;
    }
    bool bT_ready = false;    // slot DA holds rhs/diag of the next T system (written by the predictor launch)
    // Persistent step kernel (backends that have it, tedd_persist.hpp): the launches of the step are recorded
    // and run as one launch per entry-point call (or per kRecCap recorded launches), by the per-sweep solver
    // from zero iterates.  0 (default): off; 1: one workgroup (the barrier between phases is the workgroup
    // barrier); k > 1: k workgroups and a barrier over the grid.  Off by default because it is slower than the
    // separate launches on every mesh measured (MI355X, microseconds per step, launches / persistent with the
    // best number of workgroups: 8^2 50 / 65, 32^2 48 / 107, 64^2 50 / 136, 256^2 60 / 232): a phase boundary
    // inside the launch costs more than a kernel boundary, and there are 17-25 of them per step.
    int opt_persistent = 0;
    bool persist_step = false;  // the step being sequenced is recorded
    bool strip_now() const { return opt_strip && !persist_step; }
    int persist_groups(int num_pc, int num_newton, int cs_iters) const {
        if constexpr (!Backend::kHasPersist) {
            return 0;
        } else {
            const int groups = opt_persistent;
            if (groups <= 0) return 0;
            if (!Backend::persist_react_ok(react) || num_pc < 1 || num_newton < 1) return 0;
            if (gl > 0 || gr > 0 || be.reduce_hook || be.profiling) return 0;
            if (mms_case != CASE_STREAM && !(opt_cache_forcing || mms_case == CASE_PROG)) return 0;
            if (cs_iters > 8 || (long long)nrows * ncols > (1LL << 22)) return 0;
            return groups;
        }
    }

    // Register-resident solve chain (tedd_strip.hpp): b = delta^1 = rhs/diag from the node kernel,
    // then ceil((nsweeps-1)/hmax) launches that iterate on (b, delta).
    // Corrector the last launch of a cd chain may carry (tedd_strip.hpp, CORR): filled in by the step, `done` set
    // by the solve if its launch took it.
    struct CorrFuse {
        const double *ncp = nullptr, *ycs = nullptr, *cs0 = nullptr;
        double *cp_out = nullptr, *cs_out = nullptr;
        int iters = 0;
        bool done = false;
    };
    template <int VAR, int CASE, int REACT>
    void newton_solve_strip(const StepParams& p, const ForcingSrc& f1, const NewtonIn& a, double* unew, int nsweeps,
                            int slot_id, const double* u_prev, CorrFuse* cf = nullptr) {
        if constexpr (Backend::kHasStripSolve) {
            const int fam = (VAR == VAR_T) ? FAM_SOLVE_T : (VAR == VAR_CL) ? FAM_SOLVE_CL : FAM_SOLVE_CD;
            const int fam_rhs = (VAR == VAR_T) ? FAM_RHS_T : (VAR == VAR_CL) ? FAM_RHS_CL : FAM_RHS_CD;
            // from a zero iterate the node kernel's b = rhs/diag is the first sweep already; a warm start
            // performs all of them here
            int remaining = u_prev ? nsweeps : nsweeps - 1;
            // up to 10 sweeps per launch: measured best at 16384^2 (two launches of 10 + 9 for the T
            // system beat three of 7 + 6 + 6 although the column halo is then no longer line-aligned)
            int hmax = opt_strip_hmax > 0 ? opt_strip_hmax : 10;
            hmax = std::min(hmax, 16);
            const int L = std::max(1, (remaining + hmax - 1) / hmax);
            StripArgs A;
            A.g = grid();
            A.m = model;
            A.a = a;
            A.dt = p.dt;
            A.stats = reinterpret_cast<unsigned long long*>(d_red + 2 * slot_id);
            A.tiles_x = 0;
            double* bsc = B(DA);
            double* iter[2] = {B(DB), B(RHS)};
            Range r;
            if (VAR == VAR_T)
                r = need({{a.cp1, false}, {a.To, true}, {a.YT, false}});
            else if (VAR == VAR_CL)
                r = need({{a.cp1, true}, {a.To, true}, {a.clo, true}, {a.Tn, true}, {a.Ycl, false}});
            else
                r = need({{a.cp1, true}, {a.To, true}, {a.cdo, true}, {a.clo, false}, {a.cs1, false}, {a.Tn, true},
                          {a.cln, false}, {a.Ycd, false}});
            // a single-launch chain of the cl / cd system on 4-row strips evaluates its right-hand side itself
            bool fused = false;
            if constexpr (VAR != VAR_T && CASE == CASE_STREAM) {
                fused = fuse_rhs_now() && L == 1 && be.strip_can_fuse(VAR, remaining, r.lo, r.hi, ncols, opt_strip_shape);
            }
            if (VAR == VAR_T && bT_ready) {
                bT_ready = false;  // b came with the predictor launch
            } else if (!fused) {
                NewtonRhsFn<VAR, CASE, REACT> fr{A.g, model, tabs(), f1, a, nullptr, bsc, p.dt};
                be.nodes(fam_rhs, fr, r.lo, r.hi, ncols);
                mark(bsc, r);
            }
            A.xface = (opt_exact_faces && L == 1 && !fused) ? 1 : 0;
            if (fused) {
                // (the launch reads the same arrays with the same stencils as the right-hand-side kernel would,
                // then sweeps: the valid range follows from both)
                A.Y = (VAR == VAR_CL) ? a.Ycl : a.Ycd;
                A.fv = f1.f[VAR];
                mark(bsc, r);  // stands for b: valid where the right-hand side can be evaluated
            }
            const double* din = u_prev ? nullptr : bsc;
            const double* uo = (VAR == VAR_T) ? a.To : (VAR == VAR_CL) ? a.clo : a.cdo;
            for (int l = 0; l < L; ++l) {
                const int ns = (remaining + (L - l) - 1) / (L - l);
                remaining -= ns;
                const bool lastl = (l == L - 1);
                const bool warm_first = (u_prev != nullptr && l == 0);
                A.H = ns;
                A.nsweeps = ns;
                A.u_prev = warm_first ? u_prev : nullptr;
                A.gscale = warm_first ? p.gscale : 0.0;
                A.stats_first = warm_first ? reinterpret_cast<unsigned long long*>(d_red + 8 + 2 * slot_id) : nullptr;
                // (the starting iterate of a warm chain is made of the lagged unknown and u_prev)
                const double* s0 = warm_first ? uo : din;
                const double* s1 = warm_first ? u_prev : nullptr;
                if (VAR == VAR_T)
                    r = need_depth({{a.cp1, false}, {bsc, false}, {s0, false}, {s1, false}}, ns);
                else if (VAR == VAR_CL)
                    r = need_depth({{a.cp1, true}, {a.To, true}, {bsc, false}, {s0, false}, {s1, false}}, ns);
                else
                    r = need_depth({{a.cp1, true}, {a.To, true}, {a.clo, false}, {a.cs1, false}, {bsc, false},
                                    {s0, false}, {s1, false}}, ns);
                A.r0 = r.lo;
                A.r1 = r.hi;
                A.b_in = fused ? nullptr : bsc;
                A.d_in = fused ? nullptr : din;
                double* dout = lastl ? nullptr : iter[l & 1];
                A.d_out = dout;
                A.u_out = lastl ? unew : nullptr;
                if (lastl) {
                    if constexpr (VAR == VAR_CD && REACT == REACT_REG) {
                        // the record pass of the cs corrector rides on this launch where it can (whole-mesh
                        // slabs: the launch writes every node exactly once)
                        if (cf && fused && !warm_first && gl == 0 && gr == 0 && r.lo == 0 && r.hi == nrows &&
                            be.can_fuse_corrector(cf->iters) &&
                            be.strip_shape_carries_corrector(be.strip_shape_for(VAR, A.H, r.lo, r.hi, ncols, opt_strip_shape))) {
                            A.ncp = cf->ncp;
                            A.ycs = cf->ycs;
                            A.cs0 = cf->cs0;
                            A.cp_out = cf->cp_out;
                            A.cs_out = cf->cs_out;
                            A.cs_iters = cf->iters;
                            A.gmax = be.cs_record_max();
                            A.gmin = be.cs_record_min();
                            cf->done = true;
                        }
                    }
                    be.template solve_strip<VAR, REACT, true>(fam, A, ncols, opt_strip_shape);
                    mark(unew, r);
                } else {
                    be.template solve_strip<VAR, REACT, false>(fam, A, ncols, opt_strip_shape);
                    mark(dout, r);
                }
                din = dout;
            }
        }
    }

    template <int VAR, int CASE, int REACT>
    void newton_solve(const StepParams& p, const ForcingSrc& f1, const NewtonIn& a, double* unew, int nsweeps, int slot_id,
                      const double* u_prev = nullptr, CorrFuse* cf = nullptr) {
        if constexpr (Backend::kHasStripSolve) {
            if (strip_now()) {
                newton_solve_strip<VAR, CASE, REACT>(p, f1, a, unew, nsweeps, slot_id, u_prev, cf);
                return;
            }
        }
        const GridDev g = grid();
        double* rhs = B(RHS);
        double* da = B(DA);
        double* db = B(DB);
        const int fam_rhs = (VAR == VAR_T) ? FAM_RHS_T : (VAR == VAR_CL) ? FAM_RHS_CL : FAM_RHS_CD;
        const int fam_jac = (VAR == VAR_T) ? FAM_JACOBI_T : (VAR == VAR_CL) ? FAM_JACOBI_CL : FAM_JACOBI_CD;
        const double* uo = (VAR == VAR_T) ? a.To : (VAR == VAR_CL) ? a.clo : a.cdo;
        Range r;
        if (VAR == VAR_T)
            r = need({{a.cp1, false}, {a.To, true}, {a.YT, false}});
        else if (VAR == VAR_CL)
            r = need({{a.cp1, true}, {a.To, true}, {a.clo, true}, {a.Tn, true}, {a.Ycl, false}});
        else
            r = need({{a.cp1, true}, {a.To, true}, {a.cdo, true}, {a.clo, false}, {a.cs1, false}, {a.Tn, true},
                      {a.cln, false}, {a.Ycd, false}});
        NewtonRhsFn<VAR, CASE, REACT> fr{g, model, tabs(), f1, a, rhs, da, p.dt};
        be.nodes(fam_rhs, fr, r.lo, r.hi, ncols);
        mark(rhs, r);
        mark(da, r);
        double* din = da;
        double* dout = db;
        int k = 2;  // delta^1 = rhs/diag is the first sweep from a zero iterate
        if (u_prev) {
            // warm start: the first sweep runs from delta^0 = gscale (u_old - u_prev) and records
            // max|delta^1 - delta^0|
            Range rk = need({{uo, true}, {u_prev, true}, {rhs, false}, {a.cp1, VAR != VAR_T},
                             {VAR != VAR_T ? a.To : nullptr, true}});
            JacobiFn<VAR, REACT, JAC_FIRST> fj{g, model, a, rhs, nullptr, dout, p.dt, u_prev, p.gscale};
            be.nodes_max(fam_jac, fj, rk.lo, rk.hi, ncols, d_red + 8 + 2 * slot_id);
            mark(dout, rk);
            std::swap(din, dout);
        }
        for (; k <= nsweeps; ++k) {
            Range rk = need({{din, true}, {rhs, false}, {a.cp1, VAR != VAR_T}, {VAR != VAR_T ? a.To : nullptr, true}});
            if (k < nsweeps) {
                JacobiFn<VAR, REACT, JAC_PLAIN> fj{g, model, a, rhs, din, dout, p.dt};
                be.nodes(fam_jac, fj, rk.lo, rk.hi, ncols);
                mark(dout, rk);
                std::swap(din, dout);
            } else {
                JacobiFn<VAR, REACT, JAC_FINAL> fj{g, model, a, rhs, din, unew, p.dt};
                be.nodes_max(fam_jac, fj, rk.lo, rk.hi, ncols, d_red + 2 * slot_id);
                mark(unew, rk);
            }
        }
    }

    template <int CASE, int REACT>
    void step_pc_impl(const StepParams& p) {
        const GridDev g = grid();
        const MmsTabs tb = tabs();
        const double t0 = p.t0, dt = p.dt, t1 = p.t0 + p.dt;
        const ForcingSrc f0 = forcing_src(t0, 0), f1 = forcing_src(t1, 1);
        last.have = false;  // (what tedd_get_residuals needs of the step before goes back to the pool now)
        release_retained();

        // cp-forcing cell averages: reuse the array the last step computed for this time level
        const double* fcp_t0 = nullptr;
        double* fcp_t0_out = nullptr;
        double* fcp_t1_out = nullptr;
        if (CASE != CASE_STREAM) {
            int k0 = -1;
            for (int k = 0; k < 2; ++k) {
                const Range v = vof(B(FCPA + k)), o = owned();
                if (fcp_time[k] == t0 && v.lo <= o.lo && v.hi >= o.hi) k0 = k;
            }
            if (k0 >= 0) {
                fcp_t0 = B(FCPA + k0);
            } else {
                k0 = 0;
                fcp_t0_out = B(FCPA + k0);
            }
            const int k1 = 1 - k0;
            fcp_t1_out = B(FCPA + k1);
            fcp_time[k0] = t0;
            fcp_time[k1] = t1;
        }

        // Y_T is read by the right-hand side of the T system and by the Newton residual.  Where the predictor launch
        // makes b_T itself and the step has one Newton step, nothing in the step reads Y_T: it is not stored, and
        // remade from the old state if the residuals are asked for (whole-mesh contexts with library-owned forcing
        // arrays, which stay in place until the next step).
        const bool lazy_YT = Backend::kHasStripSolve && strip_now() && p.num_pc == 1 && p.num_newton == 1 && gl == 0 &&
                             gr == 0 && forcing_from_cache && CASE == CASE_STREAM && mms_case != CASE_STREAM;
        double* YT = lazy_YT ? nullptr : B(Y0);
        double* Ycl = B(Y0 + 1);
        double* Ycd = B(Y0 + 2);
        std::vector<double*> temps;  // everything acquired during this step
        auto grab = [&]() {
            double* q = acquire();
            temps.push_back(q);
            return q;
        };
        // a step that fails half-way (stale ghost rows, out of memory) hands its buffers back
        struct TempGuard {
            Ctx* c;
            std::vector<double*>* t;
            bool armed = true;
            ~TempGuard() {
                if (armed)
                    for (double* q : *t) c->release(q);
            }
        } temp_guard{this, &temps};
        double* cpP = grab();
        double* csP = grab();
        double* ncp = grab();  // t0 parts of the correctors
        double* ycs = grab();
        {
            auto pred_range = [&]() {
                return need({{cur[0], true}, {cur[1], true}, {cur[2], true}, {cur[3], true}, {cur[4], false}, {fcp_t0, false}});
            };
            Range r = pred_range();
            // Ghost rows in flight: the launch runs on the rows it can serve now, the hook brings the ghost rows,
            // then the launch runs on the bands that were left out (same functor, same values: the node results
            // do not depend on how the rows are dealt to launches).
            auto launch_split = [&](auto&& one_launch) {
                if (!ghost_pending) {
                    one_launch(r.lo, r.hi);
                    return;
                }
                const Range ri = r;
                if (ri.hi > ri.lo) one_launch(ri.lo, ri.hi);
                complete_pending_ghosts();
                r = pred_range();
                if (ri.hi <= ri.lo) {
                    one_launch(r.lo, r.hi);
                    return;
                }
                if (r.lo < ri.lo) one_launch(r.lo, ri.lo);
                if (ri.hi < r.hi) one_launch(ri.hi, r.hi);
            };
            // the register-resident solve iterates on b = rhs/diag; for the first T system the predictor
            // kernel has everything b needs
            double* bT = nullptr;
            if (Backend::kHasStripSolve && strip_now() && p.num_pc > 0 && p.num_newton > 0) bT = B(DA);
            bool launched = false;
            if constexpr (CASE == CASE_STREAM) {
                if (predict_makes_f1) {
                    // array-fed forcing whose t1 set is still to be made: this launch makes it
                    predict_makes_f1 = false;
                    Out5 fo;
                    for (int v = 0; v < 5; ++v) fo.v[v] = const_cast<double*>(f1.f[v]);
                    ForcingSrc f1c = f1;
                    f1c.tm = mms_time_fill(mms_case, model, mp, t1);
                    auto launch = [&](auto fcase) {
                        PredictFn<CASE_STREAM, REACT, decltype(fcase)::value> f{
                            g, model, tb, f0, f1c, fields(cur), YT, Ycl, Ycd, cpP, csP, fcp_t0, fcp_t0_out,
                            fcp_t1_out, dt, bT, ncp, ycs, fo};
                        launch_split([&](int lo, int hi) { launch_predictor(f, lo, hi); });
                    };
                    switch (mms_case) {
                        case CASE_POL: launch(std::integral_constant<int, CASE_POL>{}); break;
                        case CASE_EXPSIN: launch(std::integral_constant<int, CASE_EXPSIN>{}); break;
                        case CASE_PEAKS: launch(std::integral_constant<int, CASE_PEAKS>{}); break;
                        case CASE_NFS: launch(std::integral_constant<int, CASE_NFS>{}); break;
                        default: throw InvalidError("no closed-form case to evaluate the forcing from");
                    }
                    for (int v = 0; v < 5; ++v) mark(fo.v[v], r);
                    launched = true;
                }
            }
            if (!launched) {
                PredictFn<CASE, REACT> f{g, model, tb, f0, f1, fields(cur), YT, Ycl, Ycd, cpP, csP, fcp_t0, fcp_t0_out,
                                         fcp_t1_out, dt, bT, ncp, ycs, Out5{}};
                launch_split([&](int lo, int hi) { launch_predictor(f, lo, hi); });
            }
            for (double* q : {YT, Ycl, Ycd, cpP, csP, ncp, ycs})
                if (q) mark(q, r);
            if (bT) mark(bT, r);
            bT_ready = (bT != nullptr);
            if (fcp_t0_out) {
                mark(fcp_t0_out, r);
                fcp_t0 = fcp_t0_out;
            }
            if (fcp_t1_out) mark(fcp_t1_out, r);
        }

        // iterate: Newton on T -> cl -> cd with cp, cs frozen, then correct cp, cs
        double* To = cur[VAR_T];
        double* clo = cur[VAR_CL];
        double* cdo = cur[VAR_CD];
        double* cpU = cpP;  // cp, cs the Newton steps use
        double* csU = csP;
        double* cp_last_used = nullptr;
        double* cs_last_used = nullptr;
        for (int pc = 0; pc < p.num_pc; ++pc) {
            double* cpN = grab();
            double* csN = grab();
            CorrFuse cf;
            for (int nw = 0; nw < p.num_newton; ++nw) {
                double* Tn = grab();
                double* cln = grab();
                double* cdn = grab();
                NewtonIn a;
                a.cp1 = cpU;
                a.cs1 = csU;
                a.To = To;
                a.clo = clo;
                a.cdo = cdo;
                a.Tn = Tn;
                a.cln = cln;
                a.YT = YT;
                a.Ycl = Ycl;
                a.Ycd = Ycd;
                // (the increment of the step before is a starting iterate for the first Newton step only:
                // later ones correct a state that already moved)
                const bool first = (pc == 0 && nw == 0);
                newton_solve<VAR_T, CASE, REACT>(p, f1, a, Tn, p.sweeps[0], 0, first && p.warm[0] ? prev[0] : nullptr);
                newton_solve<VAR_CL, CASE, REACT>(p, f1, a, cln, p.sweeps[1], 1, first && p.warm[1] ? prev[1] : nullptr);
                CorrFuse* cfp = nullptr;
                if (Backend::kHasStripSolve && REACT == REACT_REG && nw == p.num_newton - 1 && opt_fuse_corrector) {
                    cf.ncp = ncp;
                    cf.ycs = ycs;
                    cf.cs0 = cur[VAR_CS];
                    cf.cp_out = cpN;
                    cf.cs_out = csN;
                    cf.iters = p.cs_iters;
                    cfp = &cf;
                }
                newton_solve<VAR_CD, CASE, REACT>(p, f1, a, cdn, p.sweeps[2], 2, first && p.warm[2] ? prev[2] : nullptr, cfp);
                To = Tn;
                clo = cln;
                cdo = cdn;
                cp_last_used = cpU;
                cs_last_used = csU;
            }
            CorrectIn<CASE, REACT> ci{g, model, ncp, ycs, cur[VAR_CS], To, clo, cdo, dt};
            Range r = need({{ncp, false}, {ycs, false}, {cur[VAR_CS], false}, {To, false}, {clo, false}, {cdo, false}});
            if constexpr (Backend::kHasStripSolve) {
                be.template correct<CASE, REACT>(ci, r.lo, r.hi, ncols, cpN, csN, p.cs_iters, p.cs_rtol, d_ints, cf.done);
            } else {
                be.template correct<CASE, REACT>(ci, r.lo, r.hi, ncols, cpN, csN, p.cs_iters, p.cs_rtol, d_ints);
            }
            mark(cpN, r);
            mark(csN, r);
            cpU = cpN;
            csU = csN;
        }

        double* nxt[5] = {cpU, To, clo, cdo, csU};
        double* const old_cp = cur[VAR_CP];
        double* const old_T = cur[VAR_T];
        const double* const old_fT = f0.f[VAR_T];
        commit_state(nxt, true, dt);
        temp_guard.armed = false;
        last.lazy_T0 = last.lazy_cp0 = last.lazy_fT0 = nullptr;
        if (lazy_YT) {
            // (the old cp and T stay out of the pool until the next step: commit_state has just handed them back)
            keep_until_next_step(old_cp);
            keep_until_next_step(old_T);
            last.lazy_T0 = old_T;
            last.lazy_cp0 = old_cp;
            last.lazy_fT0 = old_fT;
        }
        // keep what tedd_get_residuals needs; hand everything else back
        last.have = (cp_last_used != nullptr);
        last.t1 = t1;
        last.dt = dt;
        last.cp_used = cp_last_used;
        last.cs_used = cs_last_used;
        for (double* q : temps) {
            if (q == cp_last_used || q == cs_last_used) {
                bool is_state = false;
                for (int v = 0; v < 5; ++v) is_state = is_state || (cur[v] == q);
                if (!is_state) retained.push_back(q);
            } else {
                release(q);
            }
        }
        for (int v = 0; v < 3; ++v) last.sweeps[v] = p.sweeps[v];
        last.cs_iters_requested = p.cs_iters;
        last.cs_newton_ran = (REACT == REACT_REG) && p.num_pc > 0;
        stream_forcing_set[0] = stream_forcing_set[1] = false;
    }

    // ------------------------------------------------------------------ sub-steps of the step
    // The pieces of the predictor-corrector step the reference exposes as methods of the integrator
    // (src/prob1base.py:2946-3115).  The state set by tedd_set_state is their `at_t` / `at_t0`; further
    // inputs and the results are host arrays.  They run the same kernels as step_pc.
    struct SubstepOp {
        Ctx* c;
        int what;  // 0 predictors, 1 correctors, 2/3/4 Newton step for T/cl/cd
        double t0, dt;
        const double* in[3];
        double* out[2];
        int cs_iters;
        double cs_rtol;
        int sweeps;
        double jtol;
        int jmax;
        template <int CASE, int REACT>
        void run() {
            c->template substep_impl<CASE, REACT>(*this);
        }
    };
    struct Pooled {  // pool buffers handed back on scope exit
        Ctx* c;
        std::vector<double*> p;
        explicit Pooled(Ctx* c_) : c(c_) {}
        double* get() {
            p.push_back(c->acquire());
            return p.back();
        }
        double* upload(const double* h) {
            if (!h) throw InvalidError("missing input array");
            double* d = get();
            c->upload_field(d, h);
            return d;
        }
        ~Pooled() {
            for (double* q : p) c->release(q);
        }
    };
    template <int CASE, int REACT>
    void substep_impl(const SubstepOp& op) {
        const GridDev g = grid();
        const MmsTabs tb = tabs();
        const double t0 = op.t0, dt = op.dt, t1 = op.t0 + op.dt;
        Pooled pool(this);
        int flag = 0;
        if (op.what <= 1) {
            // predictors, and the t0 parts of the correctors, from the predictor kernel
            const ForcingSrc f0 = forcing_src(t0, 0), f1 = forcing_src(t1, 1);
            double *YT = pool.get(), *Ycl = pool.get(), *Ycd = pool.get(), *cpP = pool.get(), *csP = pool.get(),
                   *ncp = pool.get(), *ycs = pool.get();
            Range r = need({{cur[0], true}, {cur[1], true}, {cur[2], true}, {cur[3], true}, {cur[4], false}});
            PredictFn<CASE, REACT> f{g, model, tb, f0, f1, fields(cur), YT, Ycl, Ycd, cpP, csP, nullptr, nullptr,
                                     nullptr, dt, nullptr, ncp, ycs, Out5{}};
            be.nodes(FAM_PREDICT, f, r.lo, r.hi, ncols);
            for (double* q : {cpP, csP, ncp, ycs}) mark(q, r);
            if (op.what == 0) {
                if (op.out[0]) download_field(op.out[0], cpP);
                if (op.out[1]) download_field(op.out[1], csP);
            } else {
                double *T1 = pool.upload(op.in[0]), *cl1 = pool.upload(op.in[1]), *cd1 = pool.upload(op.in[2]);
                double *cpN = pool.get(), *csN = pool.get();
                CorrectIn<CASE, REACT> ci{g, model, ncp, ycs, cur[VAR_CS], T1, cl1, cd1, dt};
                Range rc = need({{ncp, false}, {ycs, false}, {cur[VAR_CS], false}});
                be.template correct<CASE, REACT>(ci, rc.lo, rc.hi, ncols, cpN, csN, op.cs_iters, op.cs_rtol, d_ints);
                if (op.out[0]) download_field(op.out[0], cpN);
                if (op.out[1]) download_field(op.out[1], csN);
                if (REACT == REACT_HCS) be.d2h(&flag, d_ints + 1, sizeof(int));
            }
        } else {
            // one Newton step of the trapezoidal rule for T, cl or cd with the state as lagged state
            const int var = op.what - 1;  // VAR_T, VAR_CL, VAR_CD
            const ForcingSrc f1 = forcing_src(t1, 1);
            NewtonIn a;
            a.cp1 = cur[VAR_CP];
            a.cs1 = cur[VAR_CS];
            a.To = cur[VAR_T];
            a.clo = cur[VAR_CL];
            a.cdo = cur[VAR_CD];
            double* Y = pool.upload(op.in[0]);
            a.YT = a.Ycl = a.Ycd = Y;
            a.Tn = (var != VAR_T) ? pool.upload(op.in[1]) : nullptr;
            a.cln = (var == VAR_CD) ? pool.upload(op.in[2]) : nullptr;
            const double dcoef[3] = {model.DT, model.Dl_max, model.Dd_max};
            const int sw = op.sweeps > 0 ? std::max(2, op.sweeps)
                                         : auto_sweeps(dt * std::fabs(dcoef[var - 1]) * q_metric, op.jtol, op.jmax);
            StepParams p{t0, dt, 1, 1, 0, 0.0, {sw, sw, sw}};
            fold_last_record();  // (the record of a step before this call is judged before it is cleared)
            record_unfolded = record_unjudged = false;
            chain_start = next_seq;
            be.memset0(d_red, 16 * sizeof(double));
            bT_ready = false;
            double* unew = pool.get();
            double* res = op.out[1] ? pool.get() : nullptr;
            if (res) be.memset0(res, field_elems() * sizeof(double));
            const double *cp = cur[VAR_CP], *cs = cur[VAR_CS];
            if (var == VAR_T) {
                newton_solve<VAR_T, CASE, REACT>(p, f1, a, unew, sw, 0);
                if (res) {
                    Range r = need({{cp, false}, {unew, true}, {Y, false}});
                    ResidualFn<VAR_T, CASE, REACT> f{g, model, tb, f1, cp, cs, unew, a.clo, a.cdo, Y, res, dt};
                    be.nodes(FAM_RESIDUAL, f, r.lo, r.hi, ncols);
                }
            } else if (var == VAR_CL) {
                newton_solve<VAR_CL, CASE, REACT>(p, f1, a, unew, sw, 1);
                if (res) {
                    Range r = need({{cp, true}, {a.Tn, true}, {unew, true}, {Y, false}});
                    ResidualFn<VAR_CL, CASE, REACT> f{g, model, tb, f1, cp, cs, a.Tn, unew, a.cdo, Y, res, dt};
                    be.nodes(FAM_RESIDUAL, f, r.lo, r.hi, ncols);
                }
            } else {
                newton_solve<VAR_CD, CASE, REACT>(p, f1, a, unew, sw, 2);
                if (res) {
                    Range r = need({{cp, true}, {a.Tn, true}, {unew, true}, {a.cln, false}, {cs, false}, {Y, false}});
                    ResidualFn<VAR_CD, CASE, REACT> f{g, model, tb, f1, cp, cs, a.Tn, a.cln, unew, Y, res, dt};
                    be.nodes(FAM_RESIDUAL, f, r.lo, r.hi, ncols);
                }
            }
            for (int v = 0; v < 3; ++v) last.sweeps[v] = sw;
            if (op.out[0]) download_field(op.out[0], unew);
            if (res) download_field(op.out[1], res);
        }
        be.sync();
        stream_forcing_set[0] = stream_forcing_set[1] = false;
        raise_corrector_flag(flag);
    }
    void substep(int what, double t0, double dt, const double* const in[3], double* const out[2], int cs_iters,
                 double cs_rtol, int sweeps, double jtol, int jmax) {
        require_state();
        refuse_pending_ghosts("a sub-step");
        if (!(dt > 0)) throw AssertError("dt must be > 0");
        if (what < 0 || what > 4) throw InvalidError("unknown sub-step");
        if (cs_iters < 0 || cs_iters > 4096) throw InvalidError("cs_newton_iterations must be in [0, 4096]");
        if (!(jtol > 0)) jtol = 1e-13;
        if (jmax < 2) jmax = 2;
        SubstepOp op{this, what, t0, dt, {in[0], in[1], in[2]}, {out[0], out[1]}, cs_iters, cs_rtol, sweeps, jtol, jmax};
        dispatch(op);
    }

    // ------------------------------------------------------------------ residuals of the last step
    struct ResOp {
        Ctx* c;
        double* out[3];
        double* lazy_out;  // where Y_T is remade if the step did not store it
        template <int CASE, int REACT>
        void run() {
            Ctx& C = *c;
            const GridDev g = C.grid();
            ForcingSrc f1;
            if (C.mms_case == CASE_PROG) {
                f1 = C.forcing_src(C.last.t1, 1);
            } else {
                f1.tm = mms_time_fill(C.mms_case, C.model, C.mp, C.last.t1);
                for (int v = 0; v < 5; ++v) f1.f[v] = (CASE == CASE_STREAM) ? C.slot[SF10 + v] : nullptr;
            }
            const double *cp = C.last.cp_used, *cs = C.last.cs_used;
            const double *T1 = C.cur[VAR_T], *cl1 = C.cur[VAR_CL], *cd1 = C.cur[VAR_CD];
            const double *YT = C.slot[Y0], *Ycl = C.slot[Y0 + 1], *Ycd = C.slot[Y0 + 2];
            if (C.last.lazy_T0) {
                // the step did not store Y_T: remake it from the old state (step_pc_impl)
                const Range ry = C.need({{C.last.lazy_T0, true}, {C.last.lazy_cp0, false}, {C.last.lazy_fT0, false}});
                LazyYTFn fy{g, C.model, C.last.lazy_T0, C.last.lazy_cp0, C.last.lazy_fT0, lazy_out, C.last.dt};
                C.be.nodes(FAM_RESIDUAL, fy, ry.lo, ry.hi, C.ncols);
                C.mark(lazy_out, ry);
                YT = lazy_out;
            }
            Range r = C.need({{cp, true}, {T1, true}, {cl1, true}, {cd1, true}, {cs, false}, {YT, false},
                              {Ycl, false}, {Ycd, false}});
            ResidualFn<VAR_T, CASE, REACT> a{g, C.model, C.tabs(), f1, cp, cs, T1, cl1, cd1, YT, out[0], C.last.dt};
            ResidualFn<VAR_CL, CASE, REACT> b{g, C.model, C.tabs(), f1, cp, cs, T1, cl1, cd1, Ycl, out[1], C.last.dt};
            ResidualFn<VAR_CD, CASE, REACT> d{g, C.model, C.tabs(), f1, cp, cs, T1, cl1, cd1, Ycd, out[2], C.last.dt};
            C.be.nodes(FAM_RESIDUAL, a, r.lo, r.hi, C.ncols);
            C.be.nodes(FAM_RESIDUAL, b, r.lo, r.hi, C.ncols);
            C.be.nodes(FAM_RESIDUAL, d, r.lo, r.hi, C.ncols);
        }
    };
    void get_residuals(double* const h[3]) {
        if (!last.have) throw InvalidError("no predictor-corrector step has been taken since the state was set");
        Tmp5 tmp(this);
        for (int v = 0; v < 3; ++v) be.memset0(tmp.p[v], field_elems() * sizeof(double));
        ResOp op{this, {tmp.p[0], tmp.p[1], tmp.p[2]}, tmp.p[3]};
        dispatch(op);
        for (int v = 0; v < 3; ++v)
            if (h[v]) download_field(h[v], op.out[v]);
        be.sync();
    }

    void set_option(const std::string& name, double value) {
        if (name == "strip_solve")
            opt_strip = value != 0.0;
        else if (name == "strip_hmax")
            opt_strip_hmax = (int)value;
        else if (name == "strip_shape")
            opt_strip_shape = (int)value;
        else if (name == "fuse_rhs")
            opt_fuse_rhs = value < 0.0 ? -1 : (value != 0.0);
        else if (name == "fuse_corrector")
            opt_fuse_corrector = value != 0.0;
        else if (name == "cache_forcing")
            opt_cache_forcing = value != 0.0;
        else if (name == "fuse_forcing")
            opt_fuse_forcing = value < 0.0 ? -1 : (value != 0.0);
        else if (name == "march" || name == "march_rows" || name == "march_occ" || name == "exact_faces" ||
                 name == "forcing_columns" || name == "strip_panel" || name == "march_ahead" || name == "march_lds" || name == "cs_lean" ||
                 name == "march_split") {
            if constexpr (Backend::kHasStripSolve) {
                if (name == "exact_faces")
                    opt_exact_faces = value != 0.0;
                else if (name == "march_ahead")
                    be.opt_march_ahead = value < 0.0 ? -1 : (value != 0.0);
                else if (name == "march_split")
                    be.opt_march_split = value < 0.0 ? -1 : (value != 0.0);
                else if (name == "cs_lean")
                    be.opt_cs_lean = value != 0.0;
                else if (name == "march_lds")
                    be.opt_march_lds = value < 0.0 ? -1 : (value != 0.0);
                else if (name == "strip_panel")
                    be.opt_strip_panel = std::max(0, std::min(1024, (int)value));
                else if (name == "forcing_columns")
                    be.opt_forcing_columns = value != 0.0;
                else if (name == "march")
                    be.opt_march = value < 0.0 ? -1 : (value != 0.0);
                else if (name == "march_rows")
                    be.opt_march_rows = (int)value;
                else
                    be.opt_march_occ = (int)value;
            }
        } else if (name == "persistent") {
            opt_persistent = (int)value;
            bT_ready = false;
        } else if (name == "prog_codegen") {
            opt_prog_codegen = value < 0.0 ? -1 : (value != 0.0);
        } else if (name == "warm_start") {
            opt_warm = value != 0.0;
            if (!opt_warm) drop_prev();
        } else if (name == "jacobi_floor") {
            if (!(value >= 0.0) || value > 1.0) throw InvalidError("jacobi_floor must be in [0, 1]");
            opt_floor = value;
        }
        else
            throw InvalidError("unknown option: " + name);
    }

    void get_stats(tedd_step_stats* s) {
        double red[20];
        int ints[4];
        fold_last_record();
        be.d2h(red, d_red, 20 * sizeof(double));
        be.d2h(ints, d_ints, 4 * sizeof(int));
        be.sync();
        const StepMeta& ml = meta[(next_seq - 1 + kRing) % kRing];
        const bool have_meta = next_seq > 0 && ml.seq == next_seq - 1;
        for (int v = 0; v < 3; ++v) {
            s->sweeps[v] = last.sweeps[v];
            s->last_update[v] = red[2 * v];
            s->increment[v] = red[2 * v + 1];
            s->warm_started[v] = (have_meta && record_unfolded && ml.warm[v]) ? 1 : 0;
            s->first_update[v] = red[8 + 2 * v];
            s->worst_ratio[v] = red[16 + v];
        }
        s->reserved = 0;
        s->cs_iterations = last.cs_newton_ran ? ints[0] : 0;
        raise_corrector_flag(ints[1]);
    }
    void set_reduce_hook(tedd_reduce_fn fn, void* user) {
        be.reduce_hook = fn;
        be.reduce_user = user;
    }
    // The reference raises ValueError inside step() (prob1base.py:3406-3409); the device records the
    // condition and the first call that synchronises reports it.
    void raise_corrector_flag(int flag) {
        if (flag == 0) return;
        be.memset0(d_ints + 1, sizeof(int));
        throw InvalidError("Denominator 2 - \xce\x94t Kd (Sd - Cd1) (1 + Cl1) below positiveness treshold.");  // UTF-8 Delta, the reference's text
    }

    // ------------------------------------------------------------------ multi-step with fused norms
    void advance(double t0, double dt, int nsteps, int integrator, int num_pc, int num_newton, int cs_iters,
                 double cs_rtol, const int32_t* sweeps, double jtol, int jmax, double* series) {
        require_state();
        refuse_pending_ghosts("tedd_advance");
        if (nsteps < 0) throw InvalidError("nsteps < 0");
        if (mms_case == CASE_STREAM) throw InvalidError("tedd_advance needs a closed-form MMS case");
        if ((gl > 0 || gr > 0) && nsteps > 1)
            throw InvalidError("tedd_advance on a partial slab: exchange ghost rows between steps instead");
        const size_t need_elems = (size_t)(nsteps + 1) * 8;
        if (series && need_elems > series_cap) {
            be.free(d_series);
            d_series = (double*)be.alloc(need_elems * sizeof(double));
            series_cap = need_elems;
        }
        // The sum launches of the norms leave partial sums; where they fit a modest array, their final passes
        // (one small launch each: 2.7 us of a 50 us step at 256^2) are all taken at the end of the run, as one launch.
        double* partials_all = nullptr;
        int sum_nb = 0;
        if constexpr (Backend::kHasPersist) {
            const Range o = owned();
            sum_nb = be.sum_blocks(o.lo, o.hi, ncols);
            const size_t need_p = (size_t)(nsteps + 1) * (size_t)sum_nb * 8;
            if (series && sum_nb > 0 && need_p * sizeof(double) <= ((size_t)256 << 20) &&
                persist_groups(num_pc, num_newton, cs_iters) == 0) {
                if (need_p > partials_cap) {
                    be.free(d_partials_all);
                    d_partials_all = nullptr;
                    partials_cap = 0;
                    d_partials_all = (double*)be.alloc(need_p * sizeof(double));
                    partials_cap = need_p;
                }
                partials_all = d_partials_all;
            }
        }
        auto norms_of = [&](double tt, int k) {
            if (!series) return;
            if (partials_all) {
                require_state();
                NormOp op{this, tt, nullptr, false, partials_all + (size_t)k * sum_nb * 8};
                dispatch(op);
            } else {
                error_norms_async(tt, d_series + (size_t)k * 8);
            }
        };
        double t = t0;
        norms_of(t, 0);
        for (int k = 0; k < nsteps; ++k) {
            if (integrator == 0)
                step_feuler(t, dt);
            else
                step_pc(t, dt, num_pc, num_newton, cs_iters, cs_rtol, sweeps, jtol, jmax);
            t += dt;
            norms_of(t, k + 1);
        }
        if constexpr (Backend::kHasPersist) {
            if (partials_all) be.sums_final_many(partials_all, sum_nb, 8, nsteps + 1, d_series);
        }
        if (series) be.d2h(series, d_series, need_elems * sizeof(double));
        int flag = 0;
        if (react == REACT_HCS) be.d2h(&flag, d_ints + 1, sizeof(int));
        be.sync();
        raise_corrector_flag(flag);
        // every solve of the run is judged: an under-converged Jacobi solve in any step is an error here,
        // where the reference's direct solve (src/prob1base.py:2103,2130) cannot fail that way
        if (integrator != 0) check_converged();
    }
};

}  // namespace tedd
