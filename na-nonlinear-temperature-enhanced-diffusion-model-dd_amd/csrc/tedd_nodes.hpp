// Per-node bodies of the kernels of the predictor-corrector step.  Each functor's
// operator()(li, j) updates local row li, column j.  The HIP launchers (tedd_kernels.hip) map
// one thread to one node; the host-compiled test twin loops over nodes with the same bodies.
//
// Reference sequencing restated (file:line into the reference's src/prob1base.py):
//   step driver 3117-3149, Heun predictors 2953-2965 / 3631-3645 / 3174-3188,
//   Newton steps T/cl/cd 2998-3115, correctors 2967-2996 / 3665-3702 / 3190-3219,
//   forward Euler 2885-2903, error norms 387-433 and mms_trial_utils.py:81-110.
#pragma once
#include "tedd_math.hpp"

namespace tedd {

// Local (slab) view of the mesh.  Rows are local: global i = i0 + li.
struct GridDev {
    int nrows, ncols, ld, i0, N, M;
    const double* rh;   // [nrows] 1/h_i   (0 at global i = 0)
    const double* rhm;  // [nrows] 1/hm_i  (0 at global i = 0, N)
    const double* rk;   // [ld]    1/k_j   (0 at j = 0)
    const double* rkm;  // [ld]    1/km_j  (0 at j = 0, M)
    const double* hm;   // [nrows] hm_i (0 at i = 0, N)  -- norm weights
    const double* km;   // [ld]
    const double* h;    // [nrows] h_i (0 at i = 0)
    const double* k;    // [ld]
};

TEDD_HD bool is_interior(const GridDev& g, int li, int j) {
    const int i = g.i0 + li;
    return i >= 1 && i <= g.N - 1 && j >= 1 && j <= g.M - 1;
}
// (column as unsigned: callers pass 0 <= j < ncols, and a zero-extended 32-bit column offset on top of a
// row base lets the device address the node from a scalar base register)
TEDD_HD size_t at(const GridDev& g, int li, int j) { return (size_t)li * g.ld + (unsigned)j; }
TEDD_HD Geo geo_at(const GridDev& g, int li, int j) {
    return Geo{g.rh[li], g.rh[li + 1], g.rhm[li], g.rk[j], g.rk[j + 1], g.rkm[j]};
}
TEDD_HD St5 load5(const double* u, const GridDev& g, int li, int j) {
    const size_t c = at(g, li, j);
    return St5{u[c], u[c + g.ld], u[c - g.ld], u[c + 1], u[c - 1]};
}
// The same at any node of the slab: neighbours outside it are replaced by the node itself (they are only
// used at interior nodes, which have all four), so the five loads need no condition around them.
TEDD_HD Geo geo_at_clamped(const GridDev& g, int li, int j) {
    const int le = (li + 1 < g.nrows) ? li + 1 : li, jn = (j + 1 < g.ncols) ? j + 1 : j;
    return Geo{g.rh[li], g.rh[le], g.rhm[li], g.rk[j], g.rk[jn], g.rkm[j]};
}
TEDD_HD St5 load5_clamped(const double* u, const GridDev& g, int li, int j) {
    const int le = (li + 1 < g.nrows) ? li + 1 : li, lw = (li > 0) ? li - 1 : li;
    const int jn = (j + 1 < g.ncols) ? j + 1 : j, js = (j > 0) ? j - 1 : j;
    return St5{u[at(g, li, j)], u[at(g, le, j)], u[at(g, lw, j)], u[at(g, li, jn)], u[at(g, li, js)]};
}
TEDD_HD void loads_done() {
#if defined(__HIP_DEVICE_COMPILE__)
    // Loads written before this point are issued before it: without it the compiler moves each load into the
    // conditional block that uses its value, and a thread then waits for its loads one block after the other.
    asm volatile("" ::: "memory");
#endif
}

struct Fields5 {  // cp, T, cl, cd, cs
    const double* v[5];
};
struct Out5 {
    double* v[5];
};

// Where the forcing terms come from: closed-form MMS evaluation (CASE >= 0) or arrays the
// caller uploaded (CASE_STREAM): f[0..4] = cell-averaged fcp, fT, fcl, fcd, fcs.
struct ForcingSrc {
    MmsTime tm;
    const double* f[5];
};

struct Forcing5 {
    double fcp, fT, fcl, fcd, fcs;
};

// All five forcing values at a node.  Unused members are dead-code-eliminated after inlining.
template <int CASE, int REACT, int QUAD = 0, class TB = MmsTabs>
TEDD_HD Forcing5 node_forcing(const ModelC& m, const TB& tb, const ForcingSrc& src, const GridDev& g,
                              int li, int j, bool interior, bool want_fcp) {
    Forcing5 r;
    if (CASE == CASE_STREAM) {
        const size_t c = at(g, li, j);
        r.fcp = get_once(src.f[0] + c);
        r.fT = get_once(src.f[1] + c);
        r.fcl = get_once(src.f[2] + c);
        r.fcd = get_once(src.f[3] + c);
        r.fcs = get_once(src.f[4] + c);
    } else {
        const MmsPt p = mms_node<CASE>(m, src.tm, tb, li, j);
        r.fT = mms_fT(m, p);
        r.fcl = mms_fcl(m, p);
        const double gain = reaction_gain<REACT>(m, p.cs);
        r.fcd = mms_fcd_gain(m, p, gain);
        r.fcs = mms_fcs_gain(m, p, gain);
        r.fcp = (interior && want_fcp) ? mms_fcp_cellavg<CASE, QUAD>(m, src.tm, tb, li, j) : 0.0;
    }
    return r;
}

struct F5 {
    double cp, T, cl, cd, cs;
};

// The semi-discrete field F(S, t) at a node given the forcing values (prob1base.py:2599-2672), from the
// node's stencil values (however the caller got them: memory, or the registers of a marching wave).
template <int REACT>
TEDD_HD F5 node_field_vals(const ModelC& m, const Geo& ge, const St5& cp, const St5& T, const St5& cl, const St5& cd,
                           double cs, const Forcing5& f, bool interior) {
    F5 F{f.fcp, f.fT, f.fcl, f.fcd, 0.0};
    if (interior) {
        const double R = reaction<REACT>(m, cl.c, cd.c, cs);
        F.cp += rhs_cp(m, cp.c, T.c, cl.c);
        F.T += rhs_T(m, ge, T, cp.c);
        F.cl += rhs_cl(m, ge, faces_Dl(m, cp), T, cl, cp.c);
        F.cd += rhs_cd_diffusion(ge, faces_Dd(m, cp, T), cd) + R;
        {
            // The sign of cs switches the sharp Heaviside gain, and manufactured solutions make cs pass
            // through zero exactly (reference tests/test_time_integrator_hcs_triple_full_step.py:133-283):
            // every operation on the cs path (here and in the cs predictor) is rounded on its own, as in the
            // reference's NumPy expressions, so that a cancellation to zero there is one here too.
#pragma clang fp contract(off)
            F.cs = f.fcs - R;
        }
    }
    return F;
}
template <int REACT>
TEDD_HD F5 node_field(const ModelC& m, const GridDev& g, const Fields5& S, const Forcing5& f, int li, int j,
                      bool interior) {
    if (!interior) return F5{f.fcp, f.fT, f.fcl, f.fcd, 0.0};
    const Geo ge = geo_at(g, li, j);
    const St5 cp = load5(S.v[VAR_CP], g, li, j), T = load5(S.v[VAR_T], g, li, j);
    const St5 cl = load5(S.v[VAR_CL], g, li, j), cd = load5(S.v[VAR_CD], g, li, j);
    return node_field_vals<REACT>(m, ge, cp, T, cl, cd, S.v[VAR_CS][at(g, li, j)], f, true);
}

// ------------------------------------------------------------------------------------
// K_F: field evaluation only (parity hook and forward Euler).  out = F, or S + dt F.
template <int CASE, int REACT, bool EULER>
struct FieldFn {
    GridDev g;
    ModelC m;
    MmsTabs tb;
    ForcingSrc f0;
    Fields5 S;
    Out5 out;
    double dt;
    TEDD_HD void operator()(int li, int j) const {
        const bool in = is_interior(g, li, j);
        const Forcing5 f = node_forcing<CASE, REACT>(m, tb, f0, g, li, j, in, true);
        const F5 F = node_field<REACT>(m, g, S, f, li, j, in);
        const size_t c = at(g, li, j);
        if (EULER) {
            out.v[0][c] = S.v[0][c] + dt * F.cp;
            out.v[1][c] = S.v[1][c] + dt * F.T;
            out.v[2][c] = S.v[2][c] + dt * F.cl;
            out.v[3][c] = S.v[3][c] + dt * F.cd;
            out.v[4][c] = S.v[4][c] + dt * F.cs;
        } else {
            out.v[0][c] = F.cp;
            out.v[1][c] = F.T;
            out.v[2][c] = F.cl;
            out.v[3][c] = F.cd;
            out.v[4][c] = F.cs;
        }
    }
};

// ------------------------------------------------------------------------------------
// K_pred: Y_v = dt F_v(S0,t0) + 2 v0 for v in {T,cl,cd}; Heun predictors for cp and cs.
// fcp_t0 (optional): cell-averaged cp-forcing at t0 computed by the previous step;
// fcp_t1 (optional): where to keep the one at t1 for the next step.
// FCASE (array-fed forcing only, CASE == CASE_STREAM): closed-form case whose forcing at t1 this
// launch evaluates itself and stores into the t1 arrays f1_out for the kernels that follow and for
// the next step (whose t0 it is) -- the arithmetic of the forcing hides under the memory traffic
// of the predictor instead of taking a launch of its own.
template <int CASE, int REACT, int FCASE = CASE_STREAM>
struct PredictFn {
    GridDev g;
    ModelC m;
    MmsTabs tb;
    ForcingSrc f0, f1;
    Fields5 S;
    double *YT, *Ycl, *Ycd, *cp1, *cs1;
    const double* fcp_t0;
    double* fcp_t0_out;
    double* fcp_t1_out;
    double dt;
    // optional: rhs/diag of the first Newton system for T (its lagged state is S with the predicted
    // cp, all of it known at this node), so that no separate right-hand-side launch is needed
    double* bT;
    // the parts of the correctors that depend on the t0 state and the forcing only:
    // ncp = (1 + dt/2 al0) cp0 + dt/2 (fcp0 + fcp1)           (prob1base.py:2967-2996)
    // ycs = 2 cs0 - dt R(S0) + dt (fcs0 + fcs1)   (RegHCsTriple, prob1base.py:3665-3702)
    //     = 2 cs0 + dt Fcs(S0,t0) + dt fcs1         (HCsTriple, prob1base.py:3376-3430)
    //     = (1 + dt/2 b0) cs0 + dt/2 (fcs0 + fcs1)    (CsTriple, prob1base.py:3190-3219)
    double *ncp, *ycs;
    Out5 f1_out;  // FCASE only
    TEDD_HD void operator()(int li, int j) const {
        const bool in = is_interior(g, li, j);
        const size_t c = at(g, li, j);
        if (CASE == CASE_STREAM) {
            // array-fed forcing: everything the node reads is requested at once, whatever kind of node it is
            // (on a small mesh the launch is as long as its longest chain of dependent loads), then the
            // arithmetic of the marching launcher's node body
            TEDD_STAMP(0);
            const St5 cp = load5_clamped(S.v[VAR_CP], g, li, j), T = load5_clamped(S.v[VAR_T], g, li, j);
            const St5 cl = load5_clamped(S.v[VAR_CL], g, li, j), cd = load5_clamped(S.v[VAR_CD], g, li, j);
            const double cs0 = S.v[VAR_CS][c];
            const Forcing5 fa{get_once(f0.f[0] + c), get_once(f0.f[1] + c), get_once(f0.f[2] + c),
                              get_once(f0.f[3] + c), get_once(f0.f[4] + c)};
            double fcp1 = 0.0, fcs1 = 0.0, fT1 = 0.0;
            if (FCASE == CASE_STREAM) {
                fcp1 = get_once(f1.f[0] + c);
                fcs1 = get_once(f1.f[4] + c);
                if (bT) fT1 = get_once(f1.f[1] + c);
            }
            const Geo ge = geo_at_clamped(g, li, j);
            loads_done();
            TEDD_STAMP_WAIT();
            TEDD_STAMP(1);
            if (FCASE != CASE_STREAM) {
                const Forcing5 fb = node_forcing<FCASE, REACT>(m, tb, f1, g, li, j, in, true);
                fcp1 = fb.fcp;
                fcs1 = fb.fcs;
                fT1 = fb.fT;
                put_once(f1_out.v[0] + c, fb.fcp);
                put_once(f1_out.v[1] + c, fb.fT);
                put_once(f1_out.v[2] + c, fb.fcl);
                put_once(f1_out.v[3] + c, fb.fcd);
                put_once(f1_out.v[4] + c, fb.fcs);
            }
            TEDD_STAMP_WAIT();
            TEDD_STAMP(2);
            const F5 F = node_field_vals<REACT>(m, ge, cp, T, cl, cd, cs0, fa, in);
            St5 Tb{T.c, 0.0, 0.0, 0.0, 0.0};
            if (bT && in) Tb = T;
            TEDD_STAMP(3);
            finish(c, in, ge, F, fa, cp.c, Tb, cl.c, cd.c, cs0, fcp1, fcs1, fT1);
            TEDD_STAMP(4);
            TEDD_STAMP_WAIT();
            TEDD_STAMP(5);
            return;
        }
        Forcing5 fa = node_forcing<CASE, REACT>(m, tb, f0, g, li, j, in, fcp_t0 == nullptr);
        if (CASE != CASE_STREAM && fcp_t0 != nullptr) fa.fcp = fcp_t0[c];
        const F5 F = node_field<REACT>(m, g, S, fa, li, j, in);
        const double cp0 = S.v[VAR_CP][c], T0 = S.v[VAR_T][c], cl0 = S.v[VAR_CL][c], cd0 = S.v[VAR_CD][c],
                     cs0 = S.v[VAR_CS][c];
        // forcing at t1: only fcp (cell average) and fcs are needed here
        double fcp1, fcs1, fT1 = 0.0;
        if (CASE == CASE_STREAM && FCASE != CASE_STREAM) {
            const Forcing5 fb = node_forcing<FCASE, REACT>(m, tb, f1, g, li, j, in, true);
            fcp1 = fb.fcp;
            fcs1 = fb.fcs;
            fT1 = fb.fT;
            put_once(f1_out.v[0] + c, fb.fcp);
            put_once(f1_out.v[1] + c, fb.fT);
            put_once(f1_out.v[2] + c, fb.fcl);
            put_once(f1_out.v[3] + c, fb.fcd);
            put_once(f1_out.v[4] + c, fb.fcs);
        } else if (CASE == CASE_STREAM) {
            fcp1 = get_once(f1.f[0] + c);
            fcs1 = get_once(f1.f[4] + c);
            if (bT) fT1 = get_once(f1.f[1] + c);
        } else {
            const MmsPt p1 = mms_node<CASE>(m, f1.tm, tb, li, j);
            fcs1 = mms_fcs<REACT>(m, p1);
            fT1 = mms_fT(m, p1);
            fcp1 = in ? mms_fcp_cellavg<CASE>(m, f1.tm, tb, li, j) : 0.0;
            if (fcp_t0_out) fcp_t0_out[c] = fa.fcp;
            if (fcp_t1_out) fcp_t1_out[c] = fcp1;
        }
        St5 T{T0, 0.0, 0.0, 0.0, 0.0};
        Geo ge{};
        if (bT && in) {
            ge = geo_at(g, li, j);
            T = load5(S.v[VAR_T], g, li, j);
        }
        finish(c, in, ge, F, fa, cp0, T, cl0, cd0, cs0, fcp1, fcs1, fT1);
    }
    // Everything after the field evaluation: Y, the Heun predictors, b_T and the corrector parts.
    // T: lagged T at the node and (where b_T is wanted, interior nodes) its four neighbours.
    // The values first, the stores after them: a launcher that has the next row's loads in flight can wait for
    // them between the two (NodeOut, node_vals, node_store below).
    struct Out {
        double YT, Ycl, Ycd, cp1, bT, cs1, ncp, ycs;
    };
    TEDD_HD Out finish_vals(bool in, const Geo& ge, const F5& F, const Forcing5& fa, double cp0, const St5& T, double cl0,
                            double cd0, double cs0, double fcp1, double fcs1, double fT1) const {
        Out o;
        const double T0 = T.c;
        o.YT = dt * F.T + 2.0 * T0;
        o.Ycl = dt * F.cl + 2.0 * cl0;
        o.Ycd = dt * F.cd + 2.0 * cd0;
        // Heun for cp (other fields frozen at t0)
        const double cps = cp0 + dt * F.cp;
        const double Fcps = fcp1 + (in ? rhs_cp(m, cps, T0, cl0) : 0.0);
        const double cp1v = cp0 + 0.5 * dt * (F.cp + Fcps);
        o.cp1 = cp1v;
        o.bT = 0.0;
        if (bT) {  // same arithmetic as newton_rhs_value<VAR_T> / newton_row<VAR_T>
            if (in) {
                const double r = o.YT - (2.0 * T.c - dt * (fT1 + rhs_T(m, ge, T, cp1v)));
                o.bT = tedd_div(r, 2.0 - dt * jac_T_FT(m, ge, cp1v).c);
            }
        }
        // Heun for cs
        double v;
        {
#pragma clang fp contract(off)  // (see node_field_vals)
            const double css = cs0 + dt * F.cs;
            const double Fcss = in ? (fcs1 - reaction<REACT>(m, cl0, cd0, css)) : 0.0;
            v = cs0 + (0.5 * dt) * (F.cs + Fcss);
        }
        o.cs1 = (REACT != REACT_CS3) ? (in ? v : 0.0 * v) : v;  // RegHCs/HCs: times the interior mask (:3645, :3374)
        // corrector inputs
        const double h = dt / 2;
        const double al0 = -m.K2 * T0 - m.K1 * (cl0 + 1.0);
        o.ncp = in ? (1.0 + h * al0) * cp0 + h * (fa.fcp + fcp1) : 0.0;
        if (REACT == REACT_REG) {
            o.ycs = 2.0 * cs0 - dt * m.Kd * (m.Sd - cd0) * (cl0 + 1.0) * heaviside_reg(cs0, m.eta) + dt * (fa.fcs + fcs1);
        } else if (REACT == REACT_HCS) {
            o.ycs = 2.0 * cs0 + dt * F.cs + dt * fcs1;  // Y0 of the sign-case corrector (:3412)
        } else {
            const double b0 = -m.Kd * (m.Sd - cd0) * (1.0 + cl0);
            o.ycs = in ? (1.0 + h * b0) * cs0 + h * (fa.fcs + fcs1) : 0.0;
        }
        return o;
    }
    TEDD_HD void finish_store(size_t c, const Out& o) const {
        if (YT) put_once(YT + c, o.YT);  // (null: nothing of this step reads it; LazyYTFn makes it if the residuals are asked for)
        put_once(Ycl + c, o.Ycl);
        put_once(Ycd + c, o.Ycd);
        put_once(cp1 + c, o.cp1);
        if (bT) put_once(bT + c, o.bT);
        put_once(cs1 + c, o.cs1);
        put_once(ncp + c, o.ncp);
        put_once(ycs + c, o.ycs);
    }
    TEDD_HD void finish(size_t c, bool in, const Geo& ge, const F5& F, const Forcing5& fa, double cp0, const St5& T,
                        double cl0, double cd0, double cs0, double fcp1, double fcs1, double fT1) const {
        finish_store(c, finish_vals(in, ge, F, fa, cp0, T, cl0, cd0, cs0, fcp1, fcs1, fT1));
    }

    // ---- interface of the marching launcher (tedd_march.hpp), array-fed forcing only: the arrays a node
    // reads with its four neighbours, the arrays it reads at the node alone (a null pointer reads as 0),
    // and the node body on values the wave holds in registers
    static constexpr bool kMarches = (CASE == CASE_STREAM && FCASE == CASE_STREAM);
    // measured best at 16384^2: two waves per SIMD (the body needs ~180 registers) and 32-row segments (alternated
    // inside one context: 8.6 ms with 32 rows, 8.65 with 24, 8.7 with 48, 8.8 with 16 or 128, 9.1 with 256)
    static constexpr int kMarchOcc = 2, kMarchRows = 32;
    // slabs from which the launch marches by itself: with one thread per node it is the faster one up to 8192^2
    // (2.25 against 2.30 ms there, 0.55 against 0.59 at 4096^2; 9.4 against 8.6 at 16384^2)
    static constexpr long long kMarchMinNodes = 1LL << 27;
    static constexpr bool kMarchAhead = false;
    // Looks ahead in registers, at two waves per SIMD.  With the next row travelling into LDS instead (k_march_lds:
    // 152 registers instead of 196, three waves per SIMD, 3 x 53 KB of LDS per compute unit) the launch takes the
    // same 8.78 ms at 16384^2, alternated with this one inside one context: what bounds it is the memory system.
    static constexpr bool kMarchLds = false;
    static constexpr int NS = 4, NC = 9;
    TEDD_HD const double* sptr(int k) const { return S.v[k]; }  // cp, T, cl, cd
    TEDD_HD const double* cptr(int k) const {
        return k == 0 ? S.v[VAR_CS] : k <= 5 ? f0.f[k - 1] : k == 6 ? f1.f[0] : k == 7 ? f1.f[4] : (bT ? f1.f[1] : nullptr);
    }
    TEDD_HD void node(int li, int j, const Geo& ge, const St5* s, const double* cv) const {
        node_store(li, j, node_vals(li, j, ge, s, cv));
    }
    // the same in two parts: everything the node computes, then its stores (the launcher waits for the loads it has
    // in flight in between, so that no wait of a later row ever has the stores of this one ahead of it)
    typedef Out NodeOut;
    TEDD_HD Out node_vals(int li, int j, const Geo& ge, const St5* s, const double* cv) const {
        const bool in = is_interior(g, li, j);
        const Forcing5 fa{cv[1], cv[2], cv[3], cv[4], cv[5]};
        const F5 F = node_field_vals<REACT>(m, ge, s[0], s[1], s[2], s[3], cv[0], fa, in);
        return finish_vals(in, ge, F, fa, s[0].c, s[1], s[2].c, s[3].c, cv[0], cv[6], cv[7], cv[8]);
    }
    TEDD_HD void node_store(int li, int j, const Out& o) const { finish_store(at(g, li, j), o); }
};

// ------------------------------------------------------------------------------------
// Newton systems (2 I - dt J_v) delta = rhs_v,  v in {T, cl, cd}, Jacobians at the lagged
// state S' = (cp1, T_old, cl_old, cd_old, cs1).

struct NewtonIn {
    const double *cp1, *cs1;           // predicted (or last corrected) cp, cs
    const double *To, *clo, *cdo;      // lagged T, cl, cd
    const double *Tn, *cln;            // new T (for cl, cd systems), new cl (for the cd system)
    const double *YT, *Ycl, *Ycd;
};

// One row of (2 I - dt J_v): returns diag and the off-diagonal multipliers of delta at e,w,n,s
// written so that  delta_c = (rhs + oe*d_e + ow*d_w + on*d_n + os*d_s) / diag.
// For v = cd the reference assembles the (0,-1) coefficients on the (i-1,j) diagonal and the
// (-1,0) ones on the (i,j-1) diagonal (prob1base.py:3097,3099); that is reproduced here.
// D_pre (optional): the face diffusivities (Dl for cl, Dd for cd) if the caller already has them.
template <int VAR, int REACT>
TEDD_HD void newton_row_vals(const ModelC& m, const Geo& ge, double dt, double cp_c, double T_e, double T_w,
                             const Face4& D, double cl_c, double cs_c, double& diag, double& oe, double& ow, double& on,
                             double& os) {
    Coef5 r;
    if (VAR == VAR_T) {
        r = jac_T_FT(m, ge, cp_c);
    } else if (VAR == VAR_CL) {
        St5 T;
        T.c = 0.0; T.n = 0.0; T.s = 0.0;
        T.e = T_e;
        T.w = T_w;
        r = jac_cl_Fcl(m, ge, D, T, cp_c, 1.0, 1.0);
    } else {
        r = jac_cd_Fcd<REACT>(m, ge, D, cl_c, cs_c);
        const double t = r.w;  // reference quirk: W and S coefficients exchanged in the cd matrix
        r.w = r.s;
        r.s = t;
    }
    diag = 2.0 - dt * r.c;
    oe = dt * r.e;
    ow = dt * r.w;
    on = dt * r.n;
    os = dt * r.s;
}
template <int VAR, int REACT>
TEDD_HD void newton_row(const ModelC& m, const GridDev& g, const NewtonIn& a, int li, int j, double dt,
                        double& diag, double& oe, double& ow, double& on, double& os, const Face4* D_pre = nullptr) {
    const Geo ge = geo_at(g, li, j);
    const size_t c = at(g, li, j);
    if (VAR == VAR_T) {
        newton_row_vals<VAR, REACT>(m, ge, dt, a.cp1[c], 0.0, 0.0, Face4{}, 0.0, 0.0, diag, oe, ow, on, os);
    } else if (VAR == VAR_CL) {
        const St5 cp = load5(a.cp1, g, li, j);
        newton_row_vals<VAR, REACT>(m, ge, dt, cp.c, a.To[c + g.ld], a.To[c - g.ld], D_pre ? *D_pre : faces_Dl(m, cp), 0.0,
                                    0.0, diag, oe, ow, on, os);
    } else {
        Face4 D;
        if (D_pre) {
            D = *D_pre;
        } else {
            const St5 cp = load5(a.cp1, g, li, j), T = load5(a.To, g, li, j);
            D = faces_Dd(m, cp, T);
        }
        newton_row_vals<VAR, REACT>(m, ge, dt, 0.0, 0.0, 0.0, D, a.clo[c], a.cs1[c], diag, oe, ow, on, os);
    }
}

// Right-hand side of the Newton system for v at an interior node (prob1base.py:3025-3036,
// 3068-3073, 3102-3108): rhs_T = Y_T - (2 T - dt F_T(S',t1)); rhs_cl = Y_cl - 2 cl + dt F_cl(S',t1)
// + dt (dFcl/dT)(Tn - T); rhs_cd = Y_cd - 2 cd + dt F_cd(S',t1) + dt (dFcd/dT)(Tn - T)
// + dt (dFcd/dcl)(cln - cl).
// D_out (optional): receives the face diffusivities the right-hand side evaluated (cl, cd).
// From values (cl and cd systems): fv forcing value, Y = Y_v, u the lagged unknown with its neighbours, Tn the
// new T with its neighbours (cl: only e, w are used), w_ok / e_ok: 0 where node (i-1,j) / (i+1,j) is a boundary
// node.  D: the face diffusivities (Dl for cl, Dd for cd); dD: dDd/dT on the faces (cd only).
template <int VAR, int REACT>
TEDD_HD double newton_rhs_from_faces(const ModelC& m, const Geo& ge, double dt, double fv, double Y, double w_ok,
                                     double e_ok, const Face4& D, const Face4& dD, double cp_c, const St5& T,
                                     const St5& u, const St5& Tn, double clc, double csc, double clnc) {
    double r;
    if (VAR == VAR_CL) {
        const St5& cl = u;
        const double F = fv + rhs_cl(m, ge, D, T, cl, cp_c);
        // coupling  dt * (dFcl/dT) (Tn - To), boundary entries of (Tn - To) dropped
        const Coef5 k = jac_T_Fcl(m, ge, cl, w_ok, e_ok);
        const double ww = Tn.w - T.w, we = Tn.e - T.e;
        r = Y - 2.0 * cl.c + dt * F + dt * (k.e * we + k.w * ww);
    } else {
        const St5& cd = u;
        const double F = fv + rhs_cd_diffusion(ge, D, cd) + reaction<REACT>(m, clc, cd.c, csc);
        // coupling with (Tn - To) keeps the boundary entries of the increment
        const Coef5 k = jac_T_Fcd(ge, dD, cd);
        const double JTw = k.c * (Tn.c - T.c) + k.e * (Tn.e - T.e) + k.w * (Tn.w - T.w) + k.n * (Tn.n - T.n) +
                           k.s * (Tn.s - T.s);
        const double Jclw = jac_cl_Fcd_diag<REACT>(m, cd.c, csc) * (clnc - clc);
        r = Y - 2.0 * cd.c + dt * F + dt * JTw + dt * Jclw;
    }
    return r;
}
template <int VAR, int REACT>
TEDD_HD double newton_rhs_vals(const ModelC& m, const Geo& ge, double dt, double fv, double Y, double w_ok, double e_ok,
                               const St5& cp, const St5& T, const St5& u, const St5& Tn, double clc, double csc,
                               double clnc, Face4* D_out) {
    Face4 D, dD{};
    if (VAR == VAR_CL)
        D = faces_Dl(m, cp);
    else
        faces_Dd_pair(m, cp, T, D, dD);
    if (D_out) *D_out = D;
    return newton_rhs_from_faces<VAR, REACT>(m, ge, dt, fv, Y, w_ok, e_ok, D, dD, cp.c, T, u, Tn, clc, csc, clnc);
}
template <int VAR, int CASE, int REACT>
TEDD_HD double newton_rhs_value(const GridDev& g, const ModelC& m, const MmsTabs& tb, const ForcingSrc& f1,
                                const NewtonIn& a, double dt, int li, int j, Face4* D_out = nullptr) {
    const size_t c = at(g, li, j);
    const Geo ge = geo_at(g, li, j);
    double fv;
    if (CASE == CASE_STREAM) {
        fv = f1.f[VAR][c];
    } else {
        const MmsPt p = mms_node<CASE>(m, f1.tm, tb, li, j);
        fv = (VAR == VAR_T) ? mms_fT(m, p) : (VAR == VAR_CL) ? mms_fcl(m, p) : mms_fcd<REACT>(m, p);
    }
    if (VAR == VAR_T) {
        const St5 T = load5(a.To, g, li, j);
        return a.YT[c] - (2.0 * T.c - dt * (fv + rhs_T(m, ge, T, a.cp1[c])));
    }
    const int i = g.i0 + li;
    const double w_ok = (i > 1) ? 1.0 : 0.0, e_ok = (i < g.N - 1) ? 1.0 : 0.0;
    const St5 cp = load5(a.cp1, g, li, j), T = load5(a.To, g, li, j);
    if (VAR == VAR_CL) {
        const St5 cl = load5(a.clo, g, li, j);
        St5 Tn{0.0, a.Tn[c + g.ld], a.Tn[c - g.ld], 0.0, 0.0};
        return newton_rhs_vals<VAR_CL, REACT>(m, ge, dt, fv, a.Ycl[c], w_ok, e_ok, cp, T, cl, Tn, 0.0, 0.0, 0.0, D_out);
    }
    const St5 cd = load5(a.cdo, g, li, j), Tn = load5(a.Tn, g, li, j);
    return newton_rhs_vals<VAR_CD, REACT>(m, ge, dt, fv, a.Ycd[c], w_ok, e_ok, cp, T, cd, Tn, a.clo[c], a.cs1[c], a.cln[c],
                                          D_out);
}

// K_rhs: right-hand side of the Newton system for v and the first Jacobi iterate
// delta^1 = rhs/diag (the iterate after one sweep from delta^0 = 0).
template <int VAR, int CASE, int REACT>
struct NewtonRhsFn {
    GridDev g;
    ModelC m;
    MmsTabs tb;
    ForcingSrc f1;
    NewtonIn a;
    double* rhs;
    double* d1;
    double dt;
    TEDD_HD void operator()(int li, int j) const {
        const size_t c = at(g, li, j);
        if (!is_interior(g, li, j)) {
            if (rhs) rhs[c] = 0.0;
            d1[c] = 0.0;
            return;
        }
        Face4 D;
        const double r = newton_rhs_value<VAR, CASE, REACT>(g, m, tb, f1, a, dt, li, j, &D);
        double diag, oe, ow, on, os;
        newton_row<VAR, REACT>(m, g, a, li, j, dt, diag, oe, ow, on, os, VAR == VAR_T ? nullptr : &D);
        if (rhs) rhs[c] = r;  // not needed by solvers that iterate on rhs/diag
        d1[c] = tedd_div(r, diag);
    }

    // ---- interface of the marching launcher (tedd_march.hpp): cl and cd systems with array-fed forcing
    static constexpr bool kMarches = (CASE == CASE_STREAM && VAR != VAR_T);
    static constexpr int kMarchOcc = (VAR == VAR_CL) ? 4 : 3, kMarchRows = 32;
    static constexpr bool kMarchAhead = false;  // (measured best at 16384^2)
    // marches wherever it is a launch of its own (slabs above the 4 M nodes up to which the solves evaluate their
    // right-hand sides themselves): rhs_cl 0.14 -> 0.11 ms at 3072^2, 0.23 -> 0.20 at 4096^2, 0.92 -> 0.78 at 8192^2
    static constexpr long long kMarchMinNodes = 1LL << 22;
    static constexpr bool kMarchLds = true;     // the next row travels into LDS while this one is computed (k_march_lds)
    static constexpr int NS = 4;                       // cp*, lagged T, lagged unknown, new T
    static constexpr int NC = (VAR == VAR_CL) ? 2 : 5;  // Y, forcing | + lagged cl, cs*, new cl
    TEDD_HD const double* sptr(int k) const {
        return k == 0 ? a.cp1 : k == 1 ? a.To : k == 2 ? (VAR == VAR_CL ? a.clo : a.cdo) : a.Tn;
    }
    TEDD_HD const double* cptr(int k) const {
        if (VAR == VAR_CL) return k == 0 ? a.Ycl : f1.f[VAR_CL];
        return k == 0 ? a.Ycd : k == 1 ? f1.f[VAR_CD] : k == 2 ? a.clo : k == 3 ? a.cs1 : a.cln;
    }
    TEDD_HD void node(int li, int j, const Geo& ge, const St5* s, const double* cv) const {
        node_store(li, j, node_vals(li, j, ge, s, cv));
    }
    // the same in two parts (see HasNodeVals in tedd_march.hpp): right-hand side and first iterate, then their stores
    struct NodeOut {
        double r, d;
    };
    TEDD_HD void node_store(int li, int j, const NodeOut& o) const {
        const size_t c = at(g, li, j);
        if (rhs) rhs[c] = o.r;  // not needed by solvers that iterate on rhs/diag
        d1[c] = o.d;
    }
    TEDD_HD NodeOut node_vals(int li, int j, const Geo& ge, const St5* s, const double* cv) const {
        if (!is_interior(g, li, j)) return NodeOut{0.0, 0.0};
        const int i = g.i0 + li;
        const double w_ok = (i > 1) ? 1.0 : 0.0, e_ok = (i < g.N - 1) ? 1.0 : 0.0;
        Face4 D;
        const double clc = (VAR == VAR_CD) ? cv[2] : 0.0, csc = (VAR == VAR_CD) ? cv[3] : 0.0;
        const double clnc = (VAR == VAR_CD) ? cv[4] : 0.0;
        const double r = newton_rhs_vals<VAR, REACT>(m, ge, dt, cv[1], cv[0], w_ok, e_ok, s[0], s[1], s[2], s[3], clc, csc,
                                                     clnc, &D);
        double diag, oe, ow, on, os;
        newton_row_vals<VAR, REACT>(m, ge, dt, s[0].c, s[1].e, s[1].w, D, clc, csc, diag, oe, ow, on, os);
        return NodeOut{r, tedd_div(r, diag)};
    }
};

// Arguments of one launch of the register-resident Newton solve chain (kernel in tedd_strip.hpp).
struct StripArgs {
    GridDev g;
    ModelC m;
    NewtonIn a;
    const double* b_in;   // rhs/diag = delta^1
    const double* d_in;   // current iterate (== b_in for the first launch of a chain)
    double* d_out;        // next iterate (all launches but the last)
    double* u_out;        // new unknown (last launch)
    double dt;
    int r0, r1;      // output rows (local)
    int H;           // halo depth of this launch
    int nsweeps;     // sweeps performed by this launch (<= H)
    int tiles_x;
    int tiles_y = 1;
    int panel = 0;   // tile rows per panel of the tile order (0: plain row-major order)
    unsigned long long* stats;
    // warm start (first launch of a chain only): the iteration starts from
    // delta^0 = gscale * (lagged unknown - u_prev), the increment of the time step before, instead of 0,
    // and max|delta^1 - delta^0| over the nodes written is recorded in stats_first[0]
    const double* u_prev = nullptr;
    double gscale = 0.0;
    unsigned long long* stats_first = nullptr;
    // fused right-hand side (single-launch chains of the cl and cd systems): the launch computes
    // b = rhs/diag itself from Y_v, the forcing array fv of the new time level and the fields in `a`
    // (b_in, d_in unused)
    const double* Y = nullptr;
    const double* fv = nullptr;
    // test hook (single-launch cl / cd chains that read b from an array): build the matrix rows from face
    // values evaluated as a fused launch evaluates them, so that the two can be compared bit for bit
    int xface = 0;
    // corrector carried by the launch (cd chain of RegHCsTriple with its right-hand side fused in, last Newton
    // step): every node the launch writes also gets the cp trapezoid and the record pass of the cs Newton
    // iteration (the six values it reads are the three below and the new T, cl, cd the thread holds)
    const double *ncp = nullptr, *ycs = nullptr, *cs0 = nullptr;
    double *cp_out = nullptr, *cs_out = nullptr;
    int cs_iters = 0;
    unsigned long long *gmax = nullptr, *gmin = nullptr;
};

// Convergence record of the Newton solves of one step (device memory, `red`):
//   red[2v], red[2v+1]   max|delta^S - delta^(S-1)| (last update) and max|delta^S| (increment), v = T, cl, cd
//   red[8+2v]            max|delta^1 - delta^0| of a warm-started solve (first update)
//   red[9+2v]            max|u| of the lagged unknown, recorded by the first sweep of a warm-started solve
//   red[16+v]            sticky: worst last_update / max(tol * increment, floor * max|u|) over the steps folded so far
// The prologue of the next step folds a record into the sticky ratios (systems with tol[v] > 0 only: sweep
// counts the caller fixed are the caller's responsibility) and publishes it to the host.
struct StepRecordArgs {
    double* red;
    double* ring_slot;  // host-visible {seq, last_update[3], increment[3], first_update[3]}; may be null
    double seq;
    double tol[3];
    int fold, clear, cs_iters;
    // An update below floor x max|u| is far below the rounding of u + delta (floor = 2^-62: 2^-10 of an ulp) and
    // counts as converged whatever the increment is: on meshes whose time step moves the unknown by less than
    // 1e-6 of its size, tol x increment alone would ask for updates 1e5 times smaller than an ulp of the unknown.
    double floor = 0.0;
};
TEDD_HD void fold_record(const StepRecordArgs& a, const double* rec, int v) {
    if (!(a.tol[v] > 0.0)) return;
    const double upd = rec[2 * v], inc = rec[2 * v + 1];
    double ratio = 0.0;
    if (upd != upd || inc != inc)
        ratio = INFINITY;
    else if (upd > 0.0) {
        const double umax = rec[9 + 2 * v];
        double den = a.tol[v] * inc;
        if (umax == umax && a.floor * umax > den) den = a.floor * umax;
        ratio = (den > 0.0) ? upd / den : INFINITY;
    }
    const double old = a.red[16 + v];
    a.red[16 + v] = (ratio > old || old != old) ? ratio : old;
}

// K_jacobi: one matrix-free Jacobi sweep; coefficients are rebuilt from the lagged fields.
// MODE JAC_FINAL: write the updated unknown u_new = u_old + delta instead of delta (boundary rule:
// T -> 0, cl/cd keep their boundary values, prob1base.py:3038-3039, 2102-2106) and report
// { max|delta_new - delta_in|, max|delta_new| } for the convergence record.
// MODE JAC_FIRST: first sweep of a warm-started solve: the incoming iterate is not an array but
// delta^0 = gscale * (u_old - u_prev), the increment of the time step before; reports
// { max|delta^1 - delta^0|, max|u_old| }.
enum : int { JAC_PLAIN = 0, JAC_FINAL = 1, JAC_FIRST = 2 };
template <int VAR, int REACT, int MODE>
struct JacobiFn {
    GridDev g;
    ModelC m;
    NewtonIn a;
    const double* rhs;
    const double* din;
    double* out;
    double dt;
    const double* u_prev = nullptr;  // JAC_FIRST
    double gscale = 0.0;
    static constexpr int NR = 2;
    TEDD_HD double iterate_in(const double* uo, int li, int j) const {
        const size_t c = at(g, li, j);
        if (MODE != JAC_FIRST) return din[c];
        // delta^0 lives on the nodes the system has unknowns at (0 on boundary nodes, like every iterate)
        return is_interior(g, li, j) ? gscale * (uo[c] - u_prev[c]) : 0.0;
    }
    TEDD_HD void operator()(int li, int j, double* red) const {
        const size_t c = at(g, li, j);
        const double* uo = (VAR == VAR_T) ? a.To : (VAR == VAR_CL) ? a.clo : a.cdo;
        if (!is_interior(g, li, j)) {
            const int i = g.i0 + li;
            const bool global_bd = (i == 0 || i == g.N || j == 0 || j == g.M);
            if (MODE == JAC_FINAL) out[c] = (VAR == VAR_T && global_bd) ? 0.0 : uo[c];
            else out[c] = 0.0;
            return;
        }
        double diag, oe, ow, on, os;
        newton_row<VAR, REACT>(m, g, a, li, j, dt, diag, oe, ow, on, os);
        const double de = iterate_in(uo, li + 1, j), dw = iterate_in(uo, li - 1, j);
        const double dnn = iterate_in(uo, li, j + 1), ds = iterate_in(uo, li, j - 1);
        const double dn = tedd_div(rhs[c] + oe * de + ow * dw + on * dnn + os * ds, diag);
        if (MODE == JAC_FINAL) {
            out[c] = uo[c] + dn;
            const double up = fabs(dn - din[c]), mag = fabs(dn);
            red[0] = up > red[0] ? up : red[0];
            red[1] = mag > red[1] ? mag : red[1];
            if (up != up || mag != mag) red[0] = INFINITY;
        } else {
            out[c] = dn;
            if (MODE == JAC_FIRST) {
                const double up = fabs(dn - iterate_in(uo, li, j));
                red[0] = (up > red[0] || up != up) ? up : red[0];
                if (up != up) red[0] = INFINITY;
                const double um = fabs(uo[c]);
                red[1] = um > red[1] ? um : red[1];
            }
        }
    }
    TEDD_HD void operator()(int li, int j) const {
        double red[2] = {0.0, 0.0};
        (*this)(li, j, red);
    }
};

// ------------------------------------------------------------------------------------
// Y_T = dt F_T(S0, t0) + 2 T0 after the fact, by the operations of PredictFn: a step whose predictor launch also
// made b_T, the only consumer of Y_T inside the step, does not store Y_T (one array pass of 68 fewer); the Newton
// residual of T needs it, and is asked for rarely.
struct LazyYTFn {
    GridDev g;
    ModelC m;
    const double *T0, *cp0, *fT0;
    double* out;
    double dt;
    TEDD_HD void operator()(int li, int j) const {
        const size_t c = at(g, li, j);
        double FT = fT0[c];
        if (is_interior(g, li, j)) FT += rhs_T(m, geo_at(g, li, j), load5(T0, g, li, j), cp0[c]);
        out[c] = dt * FT + 2.0 * T0[c];
    }
};

// ------------------------------------------------------------------------------------
// Newton residuals  2 v1 - dt F_v(S'' , t1) - Y_v  on the full array, where S'' is the lagged
// state with the already-updated unknowns substituted (prob1base.py:3041-3043, 3076-3078,
// 3111-3113).
template <int VAR, int CASE, int REACT>
struct ResidualFn {
    GridDev g;
    ModelC m;
    MmsTabs tb;
    ForcingSrc f1;
    const double *cp1, *cs1, *T1, *cl1, *cd1, *Y;
    double* out;
    double dt;
    TEDD_HD void operator()(int li, int j) const {
        const size_t c = at(g, li, j);
        const bool in = is_interior(g, li, j);
        double fv;
        if (CASE == CASE_STREAM) {
            fv = f1.f[VAR][c];
        } else {
            const MmsPt p = mms_node<CASE>(m, f1.tm, tb, li, j);
            fv = (VAR == VAR_T) ? mms_fT(m, p) : (VAR == VAR_CL) ? mms_fcl(m, p) : mms_fcd<REACT>(m, p);
        }
        double F = fv;
        const double* u = (VAR == VAR_T) ? T1 : (VAR == VAR_CL) ? cl1 : cd1;
        if (in) {
            const Geo ge = geo_at(g, li, j);
            if (VAR == VAR_T) {
                F += rhs_T(m, ge, load5(T1, g, li, j), cp1[c]);
            } else if (VAR == VAR_CL) {
                const St5 cp = load5(cp1, g, li, j);
                F += rhs_cl(m, ge, faces_Dl(m, cp), load5(T1, g, li, j), load5(cl1, g, li, j), cp.c);
            } else {
                const St5 cp = load5(cp1, g, li, j), T = load5(T1, g, li, j), cd = load5(cd1, g, li, j);
                F += rhs_cd_diffusion(ge, faces_Dd(m, cp, T), cd) + reaction<REACT>(m, cl1[c], cd.c, cs1[c]);
            }
        }
        out[c] = 2.0 * u[c] - dt * F - Y[c];
    }
};

// ------------------------------------------------------------------------------------
// Correctors.  cp: closed-form trapezoid (prob1base.py:2967-2996).  cs for CsTriple:
// closed-form trapezoid (3190-3219).  cs for RegHCsTriple: scalar Newton per node
// (3647-3702); the per-node inputs y, a are prepared by cs_newton_inputs and iterated by
// cs_newton_step; the kernel/loop around them reproduces the reference's global early exit.
// Everything that depends on the t0 state and the forcing only was folded into ncp, ycs by the
// predictor launch, so a corrector launch reads 6 arrays instead of 12.
template <int CASE, int REACT>
struct CorrectIn {
    GridDev g;
    ModelC m;
    const double *ncp, *ycs;  // from PredictFn
    const double* cs0;        // cs at t0: starting value of the scalar Newton iteration
    const double *T1, *cl1, *cd1;
    double dt;
};

template <int CASE, int REACT>
TEDD_HD double corrector_cp(const CorrectIn<CASE, REACT>& a, int li, int j) {
    const size_t c = at(a.g, li, j);
    if (!is_interior(a.g, li, j)) return 0.0;
    const ModelC& m = a.m;
    const double h = a.dt / 2;
    const double al1 = -m.K2 * a.T1[c] - m.K1 * (a.cl1[c] + 1.0);
    return tedd_div(a.ncp[c], 1.0 - h * al1);
}

template <int CASE, int REACT>
TEDD_HD double corrector_cs_closed_form(const CorrectIn<CASE, REACT>& a, int li, int j) {
    const size_t c = at(a.g, li, j);
    if (!is_interior(a.g, li, j)) return 0.0;
    const ModelC& m = a.m;
    const double h = a.dt / 2;
    const double b1 = -m.Kd * (m.Sd - a.cd1[c]) * (1.0 + a.cl1[c]);
    return tedd_div(a.ycs[c], 1.0 - h * b1);
}

// HCsTriple: 2 cs1 - dt R1 H(cs1) = Y0 with R1 = Kd (Sd - cd1)(1 + cl1) solved by the sign of Y0
// (prob1base.py:3376-3430).  *bad is set where 2 - dt R1 is not safely positive (the reference raises
// ValueError then); evaluated on every node, result masked to the interior like the reference.
template <int CASE, int REACT>
TEDD_HD double corrector_cs_sign_case(const CorrectIn<CASE, REACT>& a, int li, int j, bool* bad) {
    const size_t c = at(a.g, li, j);
    const ModelC& m = a.m;
    const double tol = 2.220446049250313e-16 * 100;
    const double R1 = (m.Sd - a.cd1[c]) * (1.0 + a.cl1[c]) * m.Kd;
    const double del = 2.0 - a.dt * R1;
    if (del < tol) *bad = true;
    const double Y0 = a.ycs[c];
    double v = 0.0;
    if (Y0 > tol) v = tedd_div(Y0, del);
    else if (Y0 < -tol) v = Y0 / 2.0;
    return is_interior(a.g, li, j) ? v : 0.0 * v;
}

// the six values the RegHCsTriple corrector reads at a node, and the same formulas on them
struct CsNodeIn {
    double ncp, ycs, cs0, T1, cl1, cd1;
};
template <int CASE, int REACT>
TEDD_HD CsNodeIn cs_node_load(const CorrectIn<CASE, REACT>& a, size_t c) {
    return CsNodeIn{get_once(a.ncp + c), get_once(a.ycs + c), a.cs0[c], a.T1[c], a.cl1[c], a.cd1[c]};
}
TEDD_HD double corrector_cp_vals(const ModelC& m, double dt, const CsNodeIn& v, bool interior) {
    if (!interior) return 0.0;
    const double h = dt / 2;
    const double al1 = -m.K2 * v.T1 - m.K1 * (v.cl1 + 1.0);
    return tedd_div(v.ncp, 1.0 - h * al1);
}
TEDD_HD void cs_newton_inputs_vals(const ModelC& m, double dt, const CsNodeIn& v, double& y, double& av, double& x0) {
    y = v.ycs;
    av = dt * m.Kd * (m.Sd - v.cd1) * (v.cl1 + 1.0);
    x0 = v.cs0;
}
// y and a of  2x + (2x - y) e^{-eta x} = y - a   (all nodes, boundary included)
template <int CASE, int REACT>
TEDD_HD void cs_newton_inputs(const CorrectIn<CASE, REACT>& a, int li, int j, double& y, double& av, double& x0) {
    const size_t c = at(a.g, li, j);
    const ModelC& m = a.m;
    y = a.ycs[c];
    av = a.dt * m.Kd * (m.Sd - a.cd1[c]) * (a.cl1[c] + 1.0);
    x0 = a.cs0[c];
}
// one Newton update; returns dx.  ex = e^{-eta x}
TEDD_HD double cs_newton_step(double x, double y, double av, double eta, double ex) {
    const double f = 2.0 * x + (2.0 * x - y) * ex - y + av;
    const double J = 2.0 + 2.0 * ex - eta * (2.0 * x - y) * ex;
    return tedd_div(-f, J);
}
TEDD_HD double cs_newton_exp(double x, double eta) { return exp(-eta * x); }
// Per-thread record of iteration k over the nodes a thread visits: rmx = max|dx| with a NaN counted as +inf,
// rmn = min|x| with a NaN counted as 0.  fmax / fmin drop NaNs, so the NaN is remembered apart, as the first
// iteration whose x was one (a compare, a select and an integer minimum instead of four compare-and-select
// pairs per iteration): a NaN in dx makes x = x + dx a NaN and every later x and dx with it, and x can become a
// NaN without dx being one only as inf - inf, where |dx| = inf is in the maximum already.
TEDD_HD void cs_record_update(double dx, double x, int k, double& rmx, double& rmn, int& first_bad) {
    rmx = fmax(rmx, fabs(dx));
    rmn = fmin(rmn, fabs(x));
    const int kb = (x != x) ? k : 0x7fffffff;
    first_bad = kb < first_bad ? kb : first_bad;
}
TEDD_HD void cs_record_finish(int k, double& rmx, double& rmn, int first_bad) {
    if (k >= first_bad) {
        rmx = INFINITY;
        rmn = 0.0;
    }
}
// e^{-eta (x + dx)} from ex = e^{-eta x}: the iteration converges quadratically, so after its first update
// |eta dx| is far below 2^-10 and the six-term series of e^{-eta dx} is exact to rounding; the library
// routine serves the values beyond that (value by value the result does not depend on the other lanes).
TEDD_HD double cs_newton_exp_next(double ex, double x_new, double dx, double eta) {
    const double z = -eta * dx;
    const bool small = fabs(z) < 0x1p-10;
    const double viaser = ex * (1.0 + z * (1.0 + z * (0.5 + z * (1.0 / 6.0 + z * (1.0 / 24.0 + z * (1.0 / 120.0))))));
#if defined(__HIP_DEVICE_COMPILE__)
    if (__all(small)) return viaser;
#endif
    return small ? viaser : exp(-eta * x_new);
}

// ------------------------------------------------------------------------------------
// Exact solution at the nodes (closed-form cases only).  what: 0 value, 1 d/dt, 2 d/dx, 3 d/dy,
// 4 Laplacian; entries the forcing terms never use (Laplacian of cp; space derivatives of cs)
// are returned as NaN.
template <int CASE>
struct ExactFn {
    GridDev g;
    ModelC m;
    MmsTabs tb;
    MmsTime tm;
    Out5 out;
    int what;
    TEDD_HD void operator()(int li, int j) const {
        const MmsPt p = mms_node<CASE>(m, tm, tb, li, j);
        const size_t c = at(g, li, j);
        const double nan = NAN;
        double r[5];
        switch (what) {
            case 1: r[0] = p.cp_t; r[1] = p.T_t; r[2] = p.cl_t; r[3] = p.cd_t; r[4] = p.cs_t; break;
            case 2: r[0] = p.cp_x; r[1] = p.T_x; r[2] = p.cl_x; r[3] = p.cd_x; r[4] = nan; break;
            case 3: r[0] = p.cp_y; r[1] = p.T_y; r[2] = p.cl_y; r[3] = p.cd_y; r[4] = nan; break;
            case 4: r[0] = nan; r[1] = p.T_l; r[2] = p.cl_l; r[3] = p.cd_l; r[4] = nan; break;
            default: r[0] = p.cp; r[1] = p.T; r[2] = p.cl; r[3] = p.cd; r[4] = p.cs; break;
        }
        for (int v = 0; v < 5; ++v) out.v[v][c] = r[v];
    }
};

// The five forcing arrays at one time level.  Per node (operator()), or for a thread that walks down a
// column with that column's table entries in registers (column / node_in_column): the same arithmetic.
template <int CASE, int REACT>
struct ForcingFn {
    GridDev g;
    ModelC m;
    MmsTabs tb;
    ForcingSrc f0;
    Out5 out;
    TEDD_HD void operator()(int li, int j) const {
        put(li, j, node_forcing<CASE, REACT>(m, tb, f0, g, li, j, is_interior(g, li, j), true));
    }
    template <int QUAD>
    TEDD_HD MmsColTabs<CASE> column(int j) const {
        return mms_col_tabs<CASE, QUAD>(tb, j);
    }
    template <int QUAD>
    TEDD_HD void node_in_column(const MmsColTabs<CASE>& ct, int li, int j) const {
        put(li, j, node_forcing<CASE, REACT, QUAD>(m, ct, f0, g, li, j, is_interior(g, li, j), true));
    }
    // which evaluation of the cell average the time level asks for (see mms_fcp_cellavg)
    int quad_mode() const { return (CASE == CASE_EXPSIN) ? (f0.tm.s[9] != 0.0 ? 1 : 2) : 2; }
    TEDD_HD void put(int li, int j, const Forcing5& f) const {
        const size_t c = at(g, li, j);
        put_once(out.v[0] + c, f.fcp);
        put_once(out.v[1] + c, f.fT);
        put_once(out.v[2] + c, f.fcl);
        put_once(out.v[3] + c, f.fcd);
        put_once(out.v[4] + c, f.fcs);
    }
};

// Jacobian stencil coefficient arrays (parity hook).  op: 0 dFT/dT, 1 dFcl/dT, 2 dFcl/dcl,
// 3 dFcd/dT, 4 dFcd/dcl, 5 dFcd/dcd.  out = coefficient at (0,0),(1,0),(-1,0),(0,1),(0,-1).
template <int REACT>
struct JacobianFn {
    GridDev g;
    ModelC m;
    Fields5 S;
    Out5 out;
    int op;
    TEDD_HD void operator()(int li, int j) const {
        const size_t c = at(g, li, j);
        Coef5 r{0, 0, 0, 0, 0};
        if (is_interior(g, li, j)) {
            const Geo ge = geo_at(g, li, j);
            const int i = g.i0 + li;
            const double w_ok = (i > 1) ? 1.0 : 0.0, e_ok = (i < g.N - 1) ? 1.0 : 0.0;
            const St5 cp = load5(S.v[VAR_CP], g, li, j), T = load5(S.v[VAR_T], g, li, j);
            const St5 cl = load5(S.v[VAR_CL], g, li, j), cd = load5(S.v[VAR_CD], g, li, j);
            const double cs = S.v[VAR_CS][c];
            switch (op) {
                case 0: r = jac_T_FT(m, ge, cp.c); break;
                case 1: r = jac_T_Fcl(m, ge, cl, w_ok, e_ok); break;
                case 2: r = jac_cl_Fcl(m, ge, faces_Dl(m, cp), T, cp.c, w_ok, e_ok); break;
                case 3: r = jac_T_Fcd(ge, faces_dDd_dT(m, cp, T), cd); break;
                case 4: r.c = jac_cl_Fcd_diag<REACT>(m, cd.c, cs); break;
                default: r = jac_cd_Fcd<REACT>(m, ge, faces_Dd(m, cp, T), cl.c, cs); break;
            }
        }
        out.v[0][c] = r.c;
        out.v[1][c] = r.e;
        out.v[2][c] = r.w;
        out.v[3][c] = r.n;
        out.v[4][c] = r.s;
    }
};

// ------------------------------------------------------------------------------------
// Error norms against the exact solution (mms_trial_utils.py:81-110 with prob1base.py:387-433):
// red[v] += e_v^2 hm_i km_j on interior nodes (v = cp,T,cl,cd,cs);
// red[5+q] += (D-x e)^2 h_i km_j [i>=1, 1<=j<=M-1] + (D-y e)^2 hm_i k_j [1<=i<=N-1, j>=1],
// q = 0,1,2 for T, cl, cd.  Only rows the caller owns are visited.
template <int CASE>
struct ErrNormFn {
    GridDev g;
    ModelC m;
    MmsTabs tb;
    MmsTime tm;
    Fields5 S;
    const double* ex[5];  // exact fields when CASE_STREAM (uploaded by the caller)
    static constexpr int NR = 8;
    TEDD_HD void exact5(int li, int j, double* e) const {
        if (CASE == CASE_STREAM) {
            const size_t c = at(g, li, j);
            for (int v = 0; v < 5; ++v) e[v] = ex[v][c];
        } else {
            const MmsPt p = mms_node<CASE>(m, tm, tb, li, j);
            e[0] = p.cp; e[1] = p.T; e[2] = p.cl; e[3] = p.cd; e[4] = p.cs;
        }
    }
    TEDD_HD void operator()(int li, int j, double* red) const {
        const int i = g.i0 + li;
        const size_t c = at(g, li, j);
        double ec[5];
        exact5(li, j, ec);
        double err[5];
        for (int v = 0; v < 5; ++v) err[v] = S.v[v][c] - ec[v];
        const bool in = is_interior(g, li, j);
        if (in) {
            const double w = g.hm[li] * g.km[j];
            for (int v = 0; v < 5; ++v) red[v] += err[v] * err[v] * w;
        }
        const bool xterm = (i >= 1 && j >= 1 && j <= g.M - 1);
        const bool yterm = (i >= 1 && i <= g.N - 1 && j >= 1);
        if (xterm) {
            double ew[5];
            exact5(li - 1, j, ew);
            const double wgt = g.h[li] * g.km[j];
            for (int q = 0; q < 3; ++q) {
                const int v = q + 1;
                const double dnum = (S.v[v][c] - S.v[v][c - g.ld]) * g.rh[li];
                const double dex = (ec[v] - ew[v]) * g.rh[li];
                const double d = dnum - dex;
                red[5 + q] += d * d * wgt;
            }
        }
        if (yterm) {
            double es[5];
            exact5(li, j - 1, es);
            const double wgt = g.hm[li] * g.k[j];
            for (int q = 0; q < 3; ++q) {
                const int v = q + 1;
                const double dnum = (S.v[v][c] - S.v[v][c - 1]) * g.rk[j];
                const double dex = (ec[v] - es[v]) * g.rk[j];
                const double d = dnum - dex;
                red[5 + q] += d * d * wgt;
            }
        }
    }
};

}  // namespace tedd
