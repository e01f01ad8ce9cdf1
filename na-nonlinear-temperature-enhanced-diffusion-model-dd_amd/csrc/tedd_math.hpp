// Pointwise arithmetic of the hot path: constitutive laws, the reghcstriple reaction, the
// 5-point flux-form stencil pieces, Jacobian stencil coefficients, MMS exact solutions and
// forcing terms.  Every function is `__host__ __device__` so that the HIP kernels and the
// host-compiled test twin (tests/cpu_twin) execute the same source.
//
// Reference semantics restated here (file:line into the reference's src/):
//   model laws            prob1base.py:96-217
//   reaction term         prob1base.py:2489-2509, 3452-3466, 3592-3593, 2875-2876, 3336-3337
//   F* right-hand sides   prob1base.py:2599-2672
//   Jacobian stencils     prob1base.py:2511-2597, 2674-2839
//   MMS forcing           prob1base.py:2313-2425, 3503-3551
//   MMS exact solutions   prob1_mms_cases.py:151-337, 406-511
#pragma once
// (the run-time compiler brings its own declarations of the math functions and fixed-width integers)
#if !defined(__HIPCC_RTC__)
#include <cmath>
#include <cstdint>
#else
#ifndef INFINITY
#define INFINITY __builtin_huge_val()
#endif
#ifndef NAN
#define NAN __builtin_nan("")
#endif
using __hip_internal::int8_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#endif

#if defined(__HIPCC__)
#define TEDD_HD __host__ __device__ __forceinline__
#else
#define TEDD_HD inline
#endif

// Diagnostic builds only (the stamp tools under tools/ define these): time stamps of a wave's progress.
#ifndef TEDD_STAMP
#define TEDD_STAMP(i)
#define TEDD_STAMP_WAIT()
#endif

namespace tedd {

// CASE_PROG is a run-time case (expression programs, tedd_prog.hpp); the kernels templated on a case
// never see it: its forcing reaches them as arrays (CASE_STREAM).
enum : int { CASE_STREAM = -1, CASE_POL = 0, CASE_EXPSIN = 1, CASE_PEAKS = 2, CASE_NFS = 3, CASE_PROG = 4 };
enum : int { REACT_CS3 = 0, REACT_REG = 1, REACT_HCS = 2 };
enum : int { VAR_CP = 0, VAR_T = 1, VAR_CL = 2, VAR_CD = 3, VAR_CS = 4 };

constexpr double kPi = 3.14159265358979323846;

// Model constants as the kernels use them.  T_shift = T_ref for DefaultModel02 and 0 for
// DefaultModel01 (prob1base.py:209-217).
struct ModelC {
    double K1, K2, K3, K4, DT, Dl_max, phi_l, gamma_T, Kd, Sd, Dd_max, phi_d, phi_T, r_sp, T_shift;
    double eta;
};

// exp(x).  For |x| < 2^-10 a degree-5 Taylor polynomial is exact to 2^-69 relative (x^6/720), i.e.
// at least as accurate as the library routine, at a sixth of its cost; almost every exponential
// on this path has such an argument (phi_l cp, phi_d cp, phi_T/T, the MMS time integrals).  The
// branch is wave-uniform in practice, so the library path is skipped entirely when no lane needs it.
TEDD_HD double tedd_exp_series(double x) {
    return 1.0 + x * (1.0 + x * (0.5 + x * (1.0 / 6.0 + x * (1.0 / 24.0 + x * (1.0 / 120.0)))));
}
TEDD_HD double tedd_exp(double x) {
    if (fabs(x) < 0x1p-10) return tedd_exp_series(x);
    return exp(x);
}

// a / b.  On the device, for |b| in (2^-500, 2^500): hardware reciprocal estimate, two Newton
// steps and one residual correction of the quotient (<= 1 ulp) -- half the instructions of the
// fully IEEE-scaled division sequence.  Anything else (0, inf, NaN, huge, tiny) takes the exact
// division, so 1/(1+inf) = 0 etc. keep their IEEE results.
TEDD_HD double tedd_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double ab = fabs(b);
    if (ab > 0x1p-500 && ab < 0x1p500) {
        double r = __builtin_amdgcn_rcp(b);
        r = fma(fma(-b, r, 1.0), r, r);
        r = fma(fma(-b, r, 1.0), r, r);
        const double q = a * r;
        return fma(fma(-b, q, a), r, q);
    }
#endif
    return a / b;
}

// Accesses to arrays a kernel touches exactly once per node (outputs, forcing values): marked
// non-temporal on the device so they do not displace the stencil rows that neighbouring tiles
// re-read from L2.
TEDD_HD void put_once(double* p, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
TEDD_HD double get_once(const double* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// ---------------------------------------------------------------- constitutive laws
TEDD_HD double law_Dl(const ModelC& m, double c) { return m.Dl_max * tedd_exp(-m.phi_l * c); }
TEDD_HD double law_V1(const ModelC& m, double T) { return m.gamma_T * T; }

// Dd(c,T) = Dd_max e^{-phi_d c} e^{-phi_T/(T+T_shift)}, and 0 where T+T_shift == 0.
TEDD_HD double law_Dd(const ModelC& m, double c, double T) {
    const double Te = T + m.T_shift;
    return (Te != 0.0) ? m.Dd_max * tedd_exp(-m.phi_d * c - tedd_div(m.phi_T, Te)) : 0.0;
}
// dDd/dT = Dd phi_T/(T+T_shift)^2
TEDD_HD double law_dDd_dT(const ModelC& m, double c, double T) {
    const double Te = T + m.T_shift;
    if (Te == 0.0) return 0.0;
    const double x = tedd_div(m.phi_T, Te);
    return m.Dd_max * tedd_exp(-m.phi_d * c - x) * tedd_div(x, Te);
}

// Dd and dDd/dT at once: one reciprocal and one exponential instead of three divisions and two
// exponentials (the right-hand side of the cd system needs both on every face)
// (law_Dd_pair's Dd alone, by the same operations)
TEDD_HD double law_Dd_p(const ModelC& m, double c, double T) {
    const double Te = T + m.T_shift;
    const double rTe = tedd_div(1.0, (Te != 0.0) ? Te : 1.0);
    const double x = m.phi_T * rTe;
    const double E = m.Dd_max * tedd_exp(-m.phi_d * c - x);
    return (Te != 0.0) ? E : 0.0;
}
TEDD_HD void law_Dd_pair(const ModelC& m, double c, double T, double& D, double& dT) {
    const double Te = T + m.T_shift;
    const double rTe = tedd_div(1.0, (Te != 0.0) ? Te : 1.0);
    const double x = m.phi_T * rTe;
    const double E = m.Dd_max * tedd_exp(-m.phi_d * c - x);
    D = (Te != 0.0) ? E : 0.0;
    dT = (Te != 0.0) ? E * x * rTe : 0.0;
}

TEDD_HD double heaviside_reg(double x, double eta) { return tedd_div(1.0, 1.0 + exp(-eta * x)); }

template <int REACT>
TEDD_HD double reaction_gain(const ModelC& m, double cs) {
    if (REACT == REACT_REG) return heaviside_reg(cs, m.eta);
    if (REACT == REACT_HCS) return (cs > 0.0) ? 1.0 : 0.0;  // sharp Heaviside, prob1base.py:3336-3337
    return cs;
}
// R = Kd g(cs) (1+cl) (Sd-cd)   (interior nodes only; callers apply the mask)
template <int REACT>
TEDD_HD double reaction(const ModelC& m, double cl, double cd, double cs) {
#pragma clang fp contract(off)  // see node_field: the cs path is evaluated without fused multiply-adds
    return (1.0 + cl) * (m.Sd - cd) * (m.Kd * reaction_gain<REACT>(m, cs));
}

// ---------------------------------------------------------------- geometry at a node
// Reciprocal metrics at node (i,j): rhw=1/h_i, rhe=1/h_{i+1}, rhm=1/hm_i, rks=1/k_j,
// rkn=1/k_{j+1}, rkm=1/km_j.  Only meaningful on interior nodes.
struct Geo {
    double rhw, rhe, rhm, rks, rkn, rkm;
};
struct St5 {  // a field at the node and its four neighbours: c=(i,j) e=(i+1,j) w=(i-1,j) n=(i,j+1) s=(i,j-1)
    double c, e, w, n, s;
};
struct Face4 {  // a face coefficient on the four faces of the node's cell
    double w, e, s, n;
};
struct Coef5 {  // one row of a 5-point Jacobian: dF_ij/du at c, e, w, n, s
    double c, e, w, n, s;
};

TEDD_HD Face4 faces_const(double v) { return Face4{v, v, v, v}; }

TEDD_HD Face4 faces_Dl(const ModelC& m, const St5& cp) {
    return Face4{law_Dl(m, 0.5 * (cp.c + cp.w)), law_Dl(m, 0.5 * (cp.e + cp.c)),
                 law_Dl(m, 0.5 * (cp.c + cp.s)), law_Dl(m, 0.5 * (cp.n + cp.c))};
}
TEDD_HD Face4 faces_Dd(const ModelC& m, const St5& cp, const St5& T) {
    return Face4{law_Dd(m, 0.5 * (cp.c + cp.w), 0.5 * (T.c + T.w)),
                 law_Dd(m, 0.5 * (cp.e + cp.c), 0.5 * (T.e + T.c)),
                 law_Dd(m, 0.5 * (cp.c + cp.s), 0.5 * (T.c + T.s)),
                 law_Dd(m, 0.5 * (cp.n + cp.c), 0.5 * (T.n + T.c))};
}
TEDD_HD void faces_Dd_pair(const ModelC& m, const St5& cp, const St5& T, Face4& D, Face4& dT) {
    law_Dd_pair(m, 0.5 * (cp.c + cp.w), 0.5 * (T.c + T.w), D.w, dT.w);
    law_Dd_pair(m, 0.5 * (cp.e + cp.c), 0.5 * (T.e + T.c), D.e, dT.e);
    law_Dd_pair(m, 0.5 * (cp.c + cp.s), 0.5 * (T.c + T.s), D.s, dT.s);
    law_Dd_pair(m, 0.5 * (cp.n + cp.c), 0.5 * (T.n + T.c), D.n, dT.n);
}
TEDD_HD Face4 faces_dDd_dT(const ModelC& m, const St5& cp, const St5& T) {
    return Face4{law_dDd_dT(m, 0.5 * (cp.c + cp.w), 0.5 * (T.c + T.w)),
                 law_dDd_dT(m, 0.5 * (cp.e + cp.c), 0.5 * (T.e + T.c)),
                 law_dDd_dT(m, 0.5 * (cp.c + cp.s), 0.5 * (T.c + T.s)),
                 law_dDd_dT(m, 0.5 * (cp.n + cp.c), 0.5 * (T.n + T.c))};
}

// D*x(D D-x u) + D*y(D D-y u) at an interior node
TEDD_HD double div_flux(const Geo& g, const Face4& D, const St5& u) {
    const double qw = D.w * ((u.c - u.w) * g.rhw), qe = D.e * ((u.e - u.c) * g.rhe);
    const double qs = D.s * ((u.c - u.s) * g.rks), qn = D.n * ((u.n - u.c) * g.rkn);
    return (qe - qw) * g.rhm + (qn - qs) * g.rkm;
}
// -D*x(Mx(V1(T)(cl+1)))  (V2 == 0)
TEDD_HD double div_advection(const ModelC& m, const Geo& g, const St5& T, const St5& cl) {
    const double vc = law_V1(m, T.c) * (cl.c + 1.0);
    const double ve = law_V1(m, T.e) * (cl.e + 1.0);
    const double vw = law_V1(m, T.w) * (cl.w + 1.0);
    return -((0.5 * (ve + vc) - 0.5 * (vc + vw)) * g.rhm);
}

// interior parts of the right-hand sides (forcing added by the caller)
TEDD_HD double rhs_cp(const ModelC& m, double cp, double T, double cl) {
    return -m.K1 * (cl + 1.0) * cp - m.K2 * T * cp;
}
TEDD_HD double rhs_T(const ModelC& m, const Geo& g, const St5& T, double cp) {
    return div_flux(g, faces_const(m.DT), T) - m.K3 * cp * T.c;
}
TEDD_HD double rhs_cl(const ModelC& m, const Geo& g, const Face4& Dl, const St5& T, const St5& cl, double cp) {
    return div_flux(g, Dl, cl) + div_advection(m, g, T, cl) - m.K4 * cp * (cl.c + 1.0);
}
TEDD_HD double rhs_cd_diffusion(const Geo& g, const Face4& Dd, const St5& cd) { return div_flux(g, Dd, cd); }

// ---------------------------------------------------------------- Jacobian rows
// Diffusion part: u -> D*x(D D-x u)+D*y(D D-y u)
TEDD_HD Coef5 jac_diffusion(const Geo& g, const Face4& D) {
    Coef5 r;
    r.w = D.w * (g.rhm * g.rhw);
    r.e = D.e * (g.rhm * g.rhe);
    r.s = D.s * (g.rkm * g.rks);
    r.n = D.n * (g.rkm * g.rkn);
    r.c = -(r.w + r.e + r.s + r.n);
    return r;
}
// dFT/dT  (prob1base.py:2674-2684)
TEDD_HD Coef5 jac_T_FT(const ModelC& m, const Geo& g, double cp) {
    Coef5 r = jac_diffusion(g, faces_const(m.DT));
    r.c -= m.K3 * cp;
    return r;
}
// dFcl/dcl (prob1base.py:2716-2750).  w_ok/e_ok are 0 where node (i-1,j)/(i+1,j) is a boundary node.
TEDD_HD Coef5 jac_cl_Fcl(const ModelC& m, const Geo& g, const Face4& Dl, const St5& T, double cp,
                         double w_ok, double e_ok) {
    Coef5 r = jac_diffusion(g, Dl);
    r.w += w_ok * (0.5 * g.rhm) * law_V1(m, T.w);
    r.e -= e_ok * (0.5 * g.rhm) * law_V1(m, T.e);
    r.c -= m.K4 * cp;
    return r;
}
// dFcl/dT  (prob1base.py:2686-2714): only the x-advection couples
TEDD_HD Coef5 jac_T_Fcl(const ModelC& m, const Geo& g, const St5& cl, double w_ok, double e_ok) {
    Coef5 r;
    r.c = 0.0;
    r.n = 0.0;
    r.s = 0.0;
    r.w = w_ok * (1.0 + cl.w) * m.gamma_T * (0.5 * g.rhm);
    r.e = -(e_ok * (1.0 + cl.e) * m.gamma_T * (0.5 * g.rhm));
    return r;
}
// dFcd/dT  (prob1base.py:2752-2800); dDdT = dDd/dT on the faces
TEDD_HD Coef5 jac_T_Fcd(const Geo& g, const Face4& dDdT, const St5& cd) {
    const double gw = ((cd.c - cd.w) * g.rhw) * dDdT.w, ge = ((cd.e - cd.c) * g.rhe) * dDdT.e;
    const double gs = ((cd.c - cd.s) * g.rks) * dDdT.s, gn = ((cd.n - cd.c) * g.rkn) * dDdT.n;
    Coef5 r;
    r.w = -(0.5 * g.rhm) * gw;
    r.e = (0.5 * g.rhm) * ge;
    r.s = -(0.5 * g.rkm) * gs;
    r.n = (0.5 * g.rkm) * gn;
    r.c = r.w + r.e + r.s + r.n;
    return r;
}
// dFcd/dcl: diagonal only (prob1base.py:2802-2809, 2541-2568)
template <int REACT>
TEDD_HD double jac_cl_Fcd_diag(const ModelC& m, double cd, double cs) {
    return (m.Sd - cd) * (m.Kd * reaction_gain<REACT>(m, cs));
}
// dFcd/dcd (prob1base.py:2811-2839, 2570-2597)
template <int REACT>
TEDD_HD Coef5 jac_cd_Fcd(const ModelC& m, const Geo& g, const Face4& Dd, double cl, double cs) {
    Coef5 r = jac_diffusion(g, Dd);
    r.c -= (1.0 + cl) * (m.Kd * reaction_gain<REACT>(m, cs));
    return r;
}

// ---------------------------------------------------------------- MMS exact solutions
// Values of the exact solution and the derivatives the forcing terms need, at one point.
struct MmsPt {
    double cp, cp_t, cp_x, cp_y;
    double T, T_t, T_x, T_y, T_l;
    double cl, cl_t, cl_x, cl_y, cl_l;
    double cd, cd_t, cd_x, cd_y, cd_l;
    double cs, cs_t;
};
// Case-specific scalars that depend on time only; filled on the host per time level.
struct MmsTime {
    double s[20];
};
// 1-D tables of the basis functions each case is built from, at the nodes (bx: [nb][nrows],
// by: [nb][ld]) and at the 3 Gauss abscissae of each node's cell (bxq: [nbq][3][nrows],
// byq: [nbq][3][ld]).  Rows are local (slab) rows.  Behind the nbq*3 quadrature rows of bxq / byq stand
// mms_num_moments(case) rows of Gauss-weighted moments of the first quadrature basis function over the
// node's cell:  moment n = sum_a w_a f(p_a)^(n+2).
struct MmsTabs {
    const double* bx;
    const double* by;
    const double* bxq;
    const double* byq;
    int nrows;
    int ld;
};
TEDD_HD double tab_x(const MmsTabs& t, int k, int li) { return t.bx[(size_t)k * t.nrows + li]; }
TEDD_HD double tab_y(const MmsTabs& t, int k, int j) { return t.by[(size_t)k * t.ld + j]; }
TEDD_HD double tab_xq(const MmsTabs& t, int k, int a, int li) { return t.bxq[((size_t)k * 3 + a) * t.nrows + li]; }
TEDD_HD double tab_yq(const MmsTabs& t, int k, int b, int j) { return t.byq[((size_t)k * 3 + b) * t.ld + j]; }

constexpr int mms_num_basis(int c) { return c == CASE_POL ? 2 : c == CASE_EXPSIN ? 2 : c == CASE_PEAKS ? 3 : c == CASE_NFS ? 15 : 0; }
constexpr int mms_num_qbasis(int c) { return c == CASE_POL ? 1 : c == CASE_EXPSIN ? 1 : c == CASE_PEAKS ? 2 : c == CASE_NFS ? 3 : 0; }
constexpr int kExpSinMoments = 6;
constexpr int mms_num_moments(int c) { return c == CASE_EXPSIN ? kExpSinMoments : 0; }
TEDD_HD double tab_xmom(const MmsTabs& t, int nbq, int n, int li) { return t.bxq[((size_t)nbq * 3 + n) * t.nrows + li]; }
TEDD_HD double tab_ymom(const MmsTabs& t, int nbq, int n, int j) { return t.byq[((size_t)nbq * 3 + n) * t.ld + j]; }

// The same tables for a thread that walks down ONE column: the column's entries of the y tables are held
// in registers (loaded once), the row's entries still come from the x tables (the same address in every
// lane of a wave that covers one row: scalar loads).  The index arguments k, a, b, n are compile-time
// constants wherever the evaluation uses them, so y[], yq[], ym[] stay in registers.
template <int CASE>
struct MmsColTabs {
    static constexpr int NB = mms_num_basis(CASE), NBQ = mms_num_qbasis(CASE), NM = mms_num_moments(CASE);
    const double* bx;
    const double* bxq;
    int nrows;
    double y[NB > 0 ? NB : 1];
    double yq[NBQ > 0 ? NBQ * 3 : 1];
    double ym[NM > 0 ? NM : 1];
};
// QUAD: 1 the moment rows only, 2 the quadrature rows only, 0 both
template <int CASE, int QUAD>
TEDD_HD MmsColTabs<CASE> mms_col_tabs(const MmsTabs& t, int j) {
    MmsColTabs<CASE> c;
    c.bx = t.bx;
    c.bxq = t.bxq;
    c.nrows = t.nrows;
#pragma unroll
    for (int k = 0; k < MmsColTabs<CASE>::NB; ++k) c.y[k] = tab_y(t, k, j);
#pragma unroll
    for (int k = 0; k < MmsColTabs<CASE>::NBQ * 3; ++k) c.yq[k] = (QUAD != 1) ? tab_yq(t, k / 3, k % 3, j) : 0.0;
#pragma unroll
    for (int n = 0; n < MmsColTabs<CASE>::NM; ++n) c.ym[n] = (QUAD != 2) ? tab_ymom(t, MmsColTabs<CASE>::NBQ, n, j) : 0.0;
    return c;
}
// Row entries of a column walker: one address per wave.  On the device they are read through the constant
// address space (the tables are written before the launch and never during it), which makes them scalar loads
// whatever stores the kernel performs in between: a vector load of a uniform address would queue up behind, and
// be waited for together with, everything else the wave has in flight.
TEDD_HD double tab_row_entry(const double* p, size_t idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    return ((const __attribute__((address_space(4))) double*)p)[idx];
#else
    return p[idx];
#endif
}
template <int CASE>
TEDD_HD double tab_x(const MmsColTabs<CASE>& t, int k, int li) { return tab_row_entry(t.bx, (size_t)k * t.nrows + li); }
template <int CASE>
TEDD_HD double tab_y(const MmsColTabs<CASE>& t, int k, int) { return t.y[k]; }
template <int CASE>
TEDD_HD double tab_xq(const MmsColTabs<CASE>& t, int k, int a, int li) { return tab_row_entry(t.bxq, ((size_t)k * 3 + a) * t.nrows + li); }
template <int CASE>
TEDD_HD double tab_yq(const MmsColTabs<CASE>& t, int k, int b, int) { return t.yq[k * 3 + b]; }
template <int CASE>
TEDD_HD double tab_xmom(const MmsColTabs<CASE>& t, int nbq, int n, int li) { return tab_row_entry(t.bxq, ((size_t)nbq * 3 + n) * t.nrows + li); }
template <int CASE>
TEDD_HD double tab_ymom(const MmsColTabs<CASE>& t, int, int n, int) { return t.ym[n]; }

TEDD_HD void mms_set_all(MmsPt& p, double u, double ut, double ux, double uy, double lap) {
    p.cp = p.T = p.cl = p.cd = p.cs = u;
    p.cp_t = p.T_t = p.cl_t = p.cd_t = p.cs_t = ut;
    p.cp_x = p.T_x = p.cl_x = p.cd_x = ux;
    p.cp_y = p.T_y = p.cl_y = p.cd_y = uy;
    p.T_l = p.cl_l = p.cd_l = lap;
}

// ExpSin pieces shared by node and quadrature evaluation.  tm.s: 0 e^{-at}, 1 e^{-t}, 2 a,
// 3 beta, 4 beta', 5 -K1 t, 6 q0, 7 q1, 8 q2  with  cp = W exp(-K1 t + beta W),
// cs = r_sp W exp(q0 + q1 W + q2 W^2);  9: 1 if the cell average of the cp-forcing is taken from the
// moment tables, 10: its factor  e^{-K1 t} (beta' - K1 e^{-t} + K2 e^{-at}) / 4,  11..16: beta^n / n!.
TEDD_HD double expsin_cp(const MmsTime& tm, double W) { return W * tedd_exp(tm.s[5] + tm.s[3] * W); }

template <int CASE, class TB>
TEDD_HD MmsPt mms_node(const ModelC& m, const MmsTime& tm, const TB& tb, int li, int j) {
    MmsPt p;
    if (CASE == CASE_POL) {  // u = X(x) X(y)/(1+t), X = x(1-x); basis 0: X, 1: X'
        const double fx = tab_x(tb, 0, li), dfx = tab_x(tb, 1, li);
        const double fy = tab_y(tb, 0, j), dfy = tab_y(tb, 1, j);
        const double phi = tm.s[0], dphi = tm.s[1];
        mms_set_all(p, phi * fx * fy, dphi * fx * fy, phi * dfx * fy, phi * fx * dfy,
                    phi * (-2.0 * fy - 2.0 * fx));
    } else if (CASE == CASE_PEAKS) {  // u = C e^{-at} (x^2+y^2)^3 sin(pi x) sin(pi y); basis 0 sin, 1 cos, 2 x
        const double sx = tab_x(tb, 0, li), cx = tab_x(tb, 1, li), x = tab_x(tb, 2, li);
        const double sy = tab_y(tb, 0, j), cy = tab_y(tb, 1, j), y = tab_y(tb, 2, j);
        const double amp = tm.s[0];
        const double R = x * x + y * y, S = sx * sy, R2 = R * R, R3 = R * R * R;
        const double u = amp * R3 * S;
        const double ux = amp * (6.0 * x * R2 * S + R3 * kPi * cx * sy);
        const double uy = amp * (6.0 * y * R2 * S + R3 * kPi * sx * cy);
        const double uxx = amp * (6.0 * R2 * S + 24.0 * x * x * R * S + 12.0 * x * R2 * kPi * cx * sy - kPi * kPi * R3 * S);
        const double uyy = amp * (6.0 * R2 * S + 24.0 * y * y * R * S + 12.0 * y * R2 * kPi * sx * cy - kPi * kPi * R3 * S);
        mms_set_all(p, u, tm.s[1] * u, ux, uy, uxx + uyy);
    } else if (CASE == CASE_NFS) {  // u_v = f_v(x) f_v(y)/(1+t); basis 3v+d: d-th derivative of f_v
        const double phi = tm.s[0], dphi = tm.s[1];
        double u[5], ut[5], ux[5], uy[5], ul[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const double fx = tab_x(tb, 3 * v, li), dfx = tab_x(tb, 3 * v + 1, li), ddfx = tab_x(tb, 3 * v + 2, li);
            const double fy = tab_y(tb, 3 * v, j), dfy = tab_y(tb, 3 * v + 1, j), ddfy = tab_y(tb, 3 * v + 2, j);
            u[v] = phi * fx * fy;
            ut[v] = dphi * fx * fy;
            ux[v] = phi * dfx * fy;
            uy[v] = phi * fx * dfy;
            ul[v] = phi * (ddfx * fy + fx * ddfy);
        }
        p.cp = u[0]; p.cp_t = ut[0]; p.cp_x = ux[0]; p.cp_y = uy[0];
        p.T = u[1]; p.T_t = ut[1]; p.T_x = ux[1]; p.T_y = uy[1]; p.T_l = ul[1];
        p.cl = u[2]; p.cl_t = ut[2]; p.cl_x = ux[2]; p.cl_y = uy[2]; p.cl_l = ul[2];
        p.cd = u[3]; p.cd_t = ut[3]; p.cd_x = ux[3]; p.cd_y = uy[3]; p.cd_l = ul[3];
        p.cs = u[4]; p.cs_t = ut[4];
    } else {  // CASE_EXPSIN; basis 0 sin(pi x), 1 cos(pi x)
        const double sx = tab_x(tb, 0, li), cx = tab_x(tb, 1, li);
        const double sy = tab_y(tb, 0, j), cy = tab_y(tb, 1, j);
        const double W = sx * sy, Wx = kPi * cx * sy, Wy = kPi * sx * cy, lapW = -2.0 * kPi * kPi * W;
        const double ea = tm.s[0], e1 = tm.s[1], a = tm.s[2], beta = tm.s[3], dbeta = tm.s[4];
        p.T = ea * W; p.T_t = -a * ea * W; p.T_x = ea * Wx; p.T_y = ea * Wy; p.T_l = ea * lapW;
        p.cl = -e1 * W; p.cl_t = e1 * W; p.cl_x = -e1 * Wx; p.cl_y = -e1 * Wy; p.cl_l = -e1 * lapW;
        p.cd = e1 * W; p.cd_t = -e1 * W; p.cd_x = e1 * Wx; p.cd_y = e1 * Wy; p.cd_l = e1 * lapW;
        const double E = tedd_exp(tm.s[5] + beta * W);
        const double cpW = E * (1.0 + beta * W);
        p.cp = W * E;
        p.cp_t = p.cp * (-m.K1 + dbeta * W);
        p.cp_x = cpW * Wx;
        p.cp_y = cpW * Wy;
        const double cs = m.r_sp * W * tedd_exp(tm.s[6] + tm.s[7] * W + tm.s[8] * W * W);
        p.cs = cs;
        p.cs_t = cs * (-m.Kd * (m.Sd - e1 * W) * (1.0 - e1 * W));
    }
    return p;
}

// Pointwise cp-forcing  d_t cp + cp (K1 (1+cl) + K2 T)  of the exact solution at the Gauss point
// (a,b) of node (li,j)'s cell  (prob1base.py:2313-2323).
template <int CASE, class TB>
TEDD_HD double mms_fcp_quad(const ModelC& m, const MmsTime& tm, const TB& tb, int li, int a, int j, int b) {
    double cp, cp_t, cl, T;
    if (CASE == CASE_POL) {
        const double ff = tab_xq(tb, 0, a, li) * tab_yq(tb, 0, b, j);
        cp = cl = T = tm.s[0] * ff;
        cp_t = tm.s[1] * ff;
    } else if (CASE == CASE_PEAKS) {
        const double x = tab_xq(tb, 1, a, li), y = tab_yq(tb, 1, b, j);
        const double R = x * x + y * y;
        const double u = tm.s[0] * (R * R * R) * (tab_xq(tb, 0, a, li) * tab_yq(tb, 0, b, j));
        cp = cl = T = u;
        cp_t = tm.s[1] * u;
    } else if (CASE == CASE_NFS) {  // q-basis 0,1,2: f_cp, f_T, f_cl
        const double fcp = tab_xq(tb, 0, a, li) * tab_yq(tb, 0, b, j);
        cp = tm.s[0] * fcp;
        cp_t = tm.s[1] * fcp;
        T = tm.s[0] * (tab_xq(tb, 1, a, li) * tab_yq(tb, 1, b, j));
        cl = tm.s[0] * (tab_xq(tb, 2, a, li) * tab_yq(tb, 2, b, j));
    } else {
        const double W = tab_xq(tb, 0, a, li) * tab_yq(tb, 0, b, j);
        cp = expsin_cp(tm, W);
        cp_t = cp * (-m.K1 + tm.s[4] * W);
        cl = -tm.s[1] * W;
        T = tm.s[0] * W;
    }
    return cp_t - (-cp * (m.K1 * (1.0 + cl) + m.K2 * T));
}

// 3x3 Gauss-Legendre cell average of the pointwise cp-forcing (prob1base.py:493-598).
// QUAD (ExpSin): 0 the time level says which evaluation applies (tm.s[9]); 1 / 2: the caller has looked,
// and only the moment / only the nine-point evaluation is compiled in.
template <int CASE, int QUAD = 0, class TB = MmsTabs>
TEDD_HD double mms_fcp_cellavg(const ModelC& m, const MmsTime& tm, const TB& tb, int li, int j) {
    const double w[3] = {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0};
    double acc = 0.0;
    if (CASE == CASE_EXPSIN && QUAD != 2 && (QUAD == 1 || tm.s[9] != 0.0)) {
        // With W = sin(pi x) sin(pi y) the pointwise forcing is  cp_t + cp (K1 (1+cl) + K2 T)
        //   = W^2 e^{-K1 t} e^{beta W} (beta' - K1 e^{-t} + K2 e^{-at}),
        // and for |beta| < 2^-10 (the host's test behind s[9]; |W| <= 1) the six-term series of e^{beta W}
        // is exact to rounding.  The 3x3 Gauss sum of W^(n+2) = X^(n+2) Y^(n+2) is the product of two
        // per-axis moments, so the cell average is six products of table entries instead of nine
        // exponentials.
        constexpr int nbq = mms_num_qbasis(CASE_EXPSIN);
#pragma unroll
        for (int n = kExpSinMoments - 1; n >= 0; --n) acc += tm.s[11 + n] * (tab_xmom(tb, nbq, n, li) * tab_ymom(tb, nbq, n, j));
        return tm.s[10] * acc;
    }
    if (CASE == CASE_EXPSIN) {
        // The nine exponentials of the exact cp in one block: the short series of tedd_exp for all of
        // them, and the library routine only if some lane of the wave has an argument outside the
        // range of the series (value by value the result is what tedd_exp returns; nine separate
        // range checks would cut the quadrature into as many branch regions).
        double W[9], E[9];
        bool small = true;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            W[q] = tab_xq(tb, 0, q / 3, li) * tab_yq(tb, 0, q % 3, j);
            E[q] = tm.s[5] + tm.s[3] * W[q];
            small = small && fabs(E[q]) < 0x1p-10;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        const bool all_small = __all(small);
#else
        const bool all_small = small;
#endif
        double P[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const double x = E[q];
            P[q] = 1.0 + x * (1.0 + x * (0.5 + x * (1.0 / 6.0 + x * (1.0 / 24.0 + x * (1.0 / 120.0)))));
        }
        if (!all_small) {
#pragma unroll
            for (int q = 0; q < 9; ++q) P[q] = (fabs(E[q]) < 0x1p-10) ? P[q] : exp(E[q]);
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const double cp = W[q] * P[q];
            const double cp_t = cp * (-m.K1 + tm.s[4] * W[q]);
            const double cl = -tm.s[1] * W[q], T = tm.s[0] * W[q];
            acc += w[q / 3] * w[q % 3] * (cp_t - (-cp * (m.K1 * (1.0 + cl) + m.K2 * T)));
        }
        return 0.25 * acc;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += w[a] * w[b] * mms_fcp_quad<CASE>(m, tm, tb, li, a, j, b);
    return 0.25 * acc;
}

// Nodal forcing terms from the exact solution (prob1base.py:2331-2425, 3503-3551).
TEDD_HD double mms_fT(const ModelC& m, const MmsPt& p) { return p.T_t - (m.DT * p.T_l - m.K3 * p.cp * p.T); }
TEDD_HD double mms_fcl(const ModelC& m, const MmsPt& p) {
    const double Dl = law_Dl(m, p.cp), dDl = -m.phi_l * Dl;
    return p.cl_t - (dDl * (p.cp_x * p.cl_x + p.cp_y * p.cl_y) + Dl * p.cl_l - law_V1(m, p.T) * p.cl_x -
                     (p.cl + 1.0) * (m.gamma_T * p.T_x) - m.K4 * p.cp * (p.cl + 1.0));
}
// gain = reaction_gain<REACT>(m, p.cs), shared by fcd and fcs
TEDD_HD double mms_fcd_gain(const ModelC& m, const MmsPt& p, double gain) {
    double Dd, dT;
    law_Dd_pair(m, p.cp, p.T, Dd, dT);
    const double dc = -m.phi_d * Dd;
    return p.cd_t - ((dc * p.cp_x + dT * p.T_x) * p.cd_x + (dc * p.cp_y + dT * p.T_y) * p.cd_y + Dd * p.cd_l +
                     m.Kd * (m.Sd - p.cd) * (p.cl + 1.0) * gain);
}
TEDD_HD double mms_fcs_gain(const ModelC& m, const MmsPt& p, double gain) {
    return p.cs_t - (-m.Kd * gain * (1.0 + p.cl) * (m.Sd - p.cd));
}
template <int REACT>
TEDD_HD double mms_fcd(const ModelC& m, const MmsPt& p) {
    return mms_fcd_gain(m, p, reaction_gain<REACT>(m, p.cs));
}
template <int REACT>
TEDD_HD double mms_fcs(const ModelC& m, const MmsPt& p) {
    return mms_fcs_gain(m, p, reaction_gain<REACT>(m, p.cs));
}

}  // namespace tedd
