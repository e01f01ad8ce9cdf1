// Host-side preparation of the closed-form MMS cases: 1-D basis tables at the nodes and at
// the Gauss abscissae of each cell, and the per-time-level scalars (see MmsTabs / MmsTime in
// tedd_math.hpp).  Exact solutions follow the reference's src/prob1_mms_cases.py:151-337,406-511;
// the quadrature abscissae follow src/prob1base.py:507-570.
#pragma once
#include <cmath>
#include <vector>

#include "tedd_math.hpp"

namespace tedd {

struct MmsParams {
    double p[8];  // PEAKS: [C, a];  NFS: [gamma_cp, gamma_T, gamma_cl, gamma_cd, gamma_cs, theta]
};

// f(x) = x(1-x)|x-theta|^gamma and its first two derivatives
inline void nfs_axis(double x, double gamma, double theta, double& f, double& df, double& ddf) {
    const double d = x - theta, ad = std::fabs(d);
    const double sg = (d > 0) - (d < 0);
    const double p0 = std::pow(ad, gamma);
    const double p1 = gamma * std::pow(ad, gamma - 1.0) * sg;
    const double p2 = gamma * (gamma - 1.0) * std::pow(ad, gamma - 2.0);
    const double w = x * (1.0 - x), dw = 1.0 - 2.0 * x;
    f = w * p0;
    df = dw * p0 + w * p1;
    ddf = -2.0 * p0 + 2.0 * dw * p1 + w * p2;
}

// basis functions of `mms_case` at coordinate x; out has mms_num_basis(case) entries
inline void mms_basis_at(int mms_case, const MmsParams& mp, double x, double* out) {
    switch (mms_case) {
        case CASE_POL:
            out[0] = x * (1.0 - x);
            out[1] = 1.0 - 2.0 * x;
            break;
        case CASE_EXPSIN:
            out[0] = std::sin(kPi * x);
            out[1] = std::cos(kPi * x);
            break;
        case CASE_PEAKS:
            out[0] = std::sin(kPi * x);
            out[1] = std::cos(kPi * x);
            out[2] = x;
            break;
        case CASE_NFS:
            for (int v = 0; v < 5; ++v) nfs_axis(x, mp.p[v], mp.p[5], out[3 * v], out[3 * v + 1], out[3 * v + 2]);
            break;
        default:
            break;
    }
}
inline void mms_qbasis_at(int mms_case, const MmsParams& mp, double x, double* out) {
    double d1, d2;
    switch (mms_case) {
        case CASE_POL:
            out[0] = x * (1.0 - x);
            break;
        case CASE_EXPSIN:
            out[0] = std::sin(kPi * x);
            break;
        case CASE_PEAKS:
            out[0] = std::sin(kPi * x);
            out[1] = x;
            break;
        case CASE_NFS:
            nfs_axis(x, mp.p[0], mp.p[5], out[0], d1, d2);  // cp
            nfs_axis(x, mp.p[1], mp.p[5], out[1], d1, d2);  // T
            nfs_axis(x, mp.p[2], mp.p[5], out[2], d1, d2);  // cl
            break;
        default:
            break;
    }
}

// Tables along one axis.  coords: the n+1 global coordinates; first: global index of table
// entry 0; count: entries to fill; stride: table row stride (>= count).
inline void build_axis_tables(int mms_case, const MmsParams& mp, const std::vector<double>& coords, int first,
                              int count, int stride, std::vector<double>& nodal, std::vector<double>& quad) {
    const int nb = mms_num_basis(mms_case), nbq = mms_num_qbasis(mms_case), nmom = mms_num_moments(mms_case);
    const int n = (int)coords.size() - 1;
    nodal.assign((size_t)nb * stride, 0.0);
    quad.assign((size_t)(nbq * 3 + nmom) * stride, 0.0);
    const double gl[3] = {-std::sqrt(3.0 / 5.0), 0.0, std::sqrt(3.0 / 5.0)};
    const double gw[3] = {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0};
    double tmp[16];
    for (int l = 0; l < count; ++l) {
        const int i = first + l;
        if (i < 0 || i > n) continue;
        mms_basis_at(mms_case, mp, coords[i], tmp);
        for (int k = 0; k < nb; ++k) nodal[(size_t)k * stride + l] = tmp[k];
        if (i >= 1 && i <= n - 1) {
            const double base = 0.5 * (coords[i - 1] + coords[i]);
            const double width = 0.5 * ((coords[i] - coords[i - 1]) + (coords[i + 1] - coords[i]));
            for (int a = 0; a < 3; ++a) {
                const double p = base + (gl[a] + 1.0) * 0.5 * width;
                mms_qbasis_at(mms_case, mp, p, tmp);
                for (int k = 0; k < nbq; ++k) quad[((size_t)k * 3 + a) * stride + l] = tmp[k];
                double pw = tmp[0];  // moment m: sum_a w_a f(p_a)^(m+2)
                for (int mo = 0; mo < nmom; ++mo) {
                    pw *= tmp[0];
                    quad[((size_t)nbq * 3 + mo) * stride + l] += gw[a] * pw;
                }
            }
        }
    }
}

inline MmsTime mms_time_fill(int mms_case, const ModelC& m, const MmsParams& mp, double t) {
    MmsTime tm;
    for (double& v : tm.s) v = 0.0;
    switch (mms_case) {
        case CASE_POL:
        case CASE_NFS:
            tm.s[0] = 1.0 / (1.0 + t);
            tm.s[1] = -1.0 / ((1.0 + t) * (1.0 + t));
            break;
        case CASE_PEAKS:
            tm.s[0] = mp.p[0] * std::exp(-mp.p[1] * t);
            tm.s[1] = -mp.p[1];
            break;
        case CASE_EXPSIN: {
            const double a = 2.0 * kPi * kPi * m.DT;
            const double ea = std::exp(-a * t), e1 = std::exp(-t);
            const double em1a = (a != 0.0) ? std::expm1(-a * t) / a : -t;
            const double em11 = std::expm1(-t), em12 = std::expm1(-2.0 * t);
            tm.s[0] = ea;
            tm.s[1] = e1;
            tm.s[2] = a;
            tm.s[3] = -m.K1 * em11 + m.K2 * em1a;       // beta
            tm.s[4] = m.K1 * e1 - m.K2 * ea;            // beta'
            tm.s[5] = -m.K1 * t;
            tm.s[6] = -m.Kd * m.Sd * t;                 // q0
            tm.s[7] = -m.Kd * (m.Sd + 1.0) * em11;      // q1
            tm.s[8] = 0.5 * m.Kd * em12;                // q2
            // cell-averaged cp-forcing from the moment tables while the series of e^{beta W} is exact
            tm.s[9] = (std::fabs(tm.s[3]) < 0x1p-10) ? 1.0 : 0.0;
            tm.s[10] = 0.25 * std::exp(tm.s[5]) * (tm.s[4] - m.K1 * e1 + m.K2 * ea);
            tm.s[11] = 1.0;
            for (int k = 1; k < kExpSinMoments; ++k) tm.s[11 + k] = tm.s[10 + k] * tm.s[3] / k;
            break;
        }
        default:
            break;
    }
    return tm;
}

}  // namespace tedd
