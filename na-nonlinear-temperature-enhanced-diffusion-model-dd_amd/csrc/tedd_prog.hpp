// Manufactured solutions given as expression programs.
//
// The reference's MMSCaseSymbolic (src/prob1base.py:1225-1347) takes the exact solution as SymPy
// expressions of (t, x, y), derives the needed derivatives symbolically and lambdifies them for
// NumPy; its ForcingTerms_* classes (src/prob1base.py:2296-2425, 3222-3295, 3468-3551) combine those
// arrays into the five forcing terms.  Here the host lowers the same expressions (after common
// subexpression elimination and register allocation) to a straight-line program of three-address
// operations on double registers, and every node runs that program on the device: no run-time
// compiler, and no forcing arrays crossing PCIe at every time level.
//
// Two programs per case:
//   node program: 21 outputs, the members of MmsPt in declaration order
//                 (cp cp_t cp_x cp_y | T T_t T_x T_y T_lap | cl ... | cd ... | cs cs_t);
//   cell program:  4 outputs (cp, cp_t, cl, T) for the pointwise cp-forcing at the 3x3 Gauss
//                 points of a node's cell (src/prob1base.py:493-598, 2313-2328).
//
// Word format (32 bits): op | dst << 8 | a << 16 | b << 24.  Registers 0, 1, 2 hold t, x, y.
//   CONST: dst = consts[a | b << 8];  POWI: dst = reg[a] ** int8(b);  OUT: out[dst] = reg[a].
// The registers of a thread live in LDS (one column per lane, so register k of all lanes is one
// conflict-free row), because a register file indexed by program words cannot live in VGPRs.
#pragma once
// (the run-time compiler brings its own declarations of the math functions and fixed-width integers)
#if !defined(__HIPCC_RTC__)
#include <cmath>
#include <cstdint>
#include <vector>
#endif

#include "tedd_nodes.hpp"

namespace tedd {

// Version of the contract between the prebuilt library and the kernels it generates and compiles at run time against
// THIS header tree (argument structures passed by value, prog_forcing_node / prog_exact_node): raise it with any
// change to that contract.  The generated source asserts the value and the structure sizes the library was built
// with, so that a header tree that does not belong to the library fails to compile (and the interpreter is used).
constexpr int kProgContractVersion = 1;


enum ProgOp : int {
    POP_CONST = 0, POP_ADD, POP_SUB, POP_MUL, POP_DIV, POP_NEG, POP_POWI, POP_POW, POP_EXP, POP_SIN, POP_COS,
    POP_SQRT, POP_LOG, POP_ABS, POP_SIGN, POP_TANH, POP_OUT, POP_MOV, POP_COUNT
};
constexpr int kProgMaxRegs = 96;      // 96 x 64 lanes x 8 B = 48 KB of LDS per wave-sized workgroup
constexpr int kProgNodeOutputs = 21;  // MmsPt
constexpr int kProgCellOutputs = 4;   // cp, cp_t, cl, T

struct MmsProg {
    const uint32_t* code;
    const double* consts;
    int n_ops;
};

#if !defined(__HIPCC_RTC__)
// host copy of a validated program (what the run-time code generator reads)
struct HostProg {
    std::vector<uint32_t> code;
    std::vector<double> consts;
    int n_regs = 0;
};
#endif

// reg[k] of this thread is regs[k * stride]
TEDD_HD void prog_run(const MmsProg& P, double* regs, int stride, double t, double x, double y, double* out) {
    regs[0] = t;
    regs[stride] = x;
    regs[2 * stride] = y;
    for (int pc = 0; pc < P.n_ops; ++pc) {
        const uint32_t w = P.code[pc];
        const int op = (int)(w & 0xffu), d = (int)((w >> 8) & 0xffu), ia = (int)((w >> 16) & 0xffu), ib = (int)(w >> 24);
        if (op == POP_CONST) {
            regs[d * stride] = P.consts[ia | (ib << 8)];
            continue;
        }
        const double a = regs[ia * stride];
        if (op == POP_OUT) {
            out[d] = a;
            continue;
        }
        double r;
        switch (op) {
            case POP_ADD: r = a + regs[ib * stride]; break;
            case POP_SUB: r = a - regs[ib * stride]; break;
            case POP_MUL: r = a * regs[ib * stride]; break;
            case POP_DIV: r = a / regs[ib * stride]; break;
            case POP_NEG: r = -a; break;
            case POP_POWI: {
                int n = (int)(int8_t)ib;
                const bool inv = n < 0;
                n = inv ? -n : n;
                double base = a;
                r = 1.0;
                while (n) {
                    if (n & 1) r *= base;
                    base *= base;
                    n >>= 1;
                }
                if (inv) r = 1.0 / r;
                break;
            }
            case POP_POW: r = pow(a, regs[ib * stride]); break;
            case POP_EXP: r = exp(a); break;
            case POP_SIN: r = sin(a); break;
            case POP_COS: r = cos(a); break;
            case POP_SQRT: r = sqrt(a); break;
            case POP_LOG: r = log(a); break;
            case POP_ABS: r = fabs(a); break;
            case POP_SIGN: r = (a > 0.0) ? 1.0 : (a < 0.0) ? -1.0 : a; break;  // sign(0) = 0, sign(nan) = nan
            case POP_TANH: r = tanh(a); break;
            default: r = a; break;  // POP_MOV
        }
        regs[d * stride] = r;
    }
}

// coordinates of the local rows / the columns, and of the 3 Gauss points of each interior node's cell
struct ProgAxes {
    const double *x, *y;    // [nrows], [ld]
    const double *xq, *yq;  // [3][nrows], [3][ld]
    int nrows, ld;
};

// How a thread evaluates the two programs.  ProgInterp runs the words (registers in LDS or, on the host, in a
// small array); source generated from the words at run time (tedd_codegen.hpp) defines its own evaluator with
// the same two members, the programs spelled out as straight-line code.
struct ProgInterp {
    MmsProg node_p, cell_p;
    double* regs;
    int stride;
    TEDD_HD void node(double t, double x, double y, double* out) const { prog_run(node_p, regs, stride, t, x, y, out); }
    TEDD_HD void cell(double t, double x, double y, double* out) const { prog_run(cell_p, regs, stride, t, x, y, out); }
};

// Exact solution (or one kind of derivative) at a node: same output convention as ExactFn.
template <class Eval>
TEDD_HD void prog_exact_node(const Eval& ev, const GridDev& g, const ProgAxes& ax, double t, const Out5& out, int what, int li,
                             int j) {
    double p[kProgNodeOutputs];
    ev.node(t, ax.x[li], ax.y[j], p);
    const size_t c = at(g, li, j);
    const double nan = NAN;
    double r[5];
    switch (what) {
        case 1: r[0] = p[1]; r[1] = p[5]; r[2] = p[10]; r[3] = p[15]; r[4] = p[20]; break;
        case 2: r[0] = p[2]; r[1] = p[6]; r[2] = p[11]; r[3] = p[16]; r[4] = nan; break;
        case 3: r[0] = p[3]; r[1] = p[7]; r[2] = p[12]; r[3] = p[17]; r[4] = nan; break;
        case 4: r[0] = nan; r[1] = p[8]; r[2] = p[13]; r[3] = p[18]; r[4] = nan; break;
        default: r[0] = p[0]; r[1] = p[4]; r[2] = p[9]; r[3] = p[14]; r[4] = p[19]; break;
    }
    for (int v = 0; v < 5; ++v) out.v[v][c] = r[v];
}

// The five forcing values of a node at one time: nodal fT, fcl, fcd, fcs from the node program through the same
// formulas as the closed-form cases (mms_fT ... mms_fcs_gain), fcp as the 3x3 Gauss cell average of the
// pointwise cp-forcing from the cell program (0 on the boundary).
template <int REACT, class Eval>
TEDD_HD void prog_forcing_node(const Eval& ev, const GridDev& g, const ModelC& m, const ProgAxes& ax, double t, const Out5& out,
                               int li, int j) {
    double q[kProgNodeOutputs];
    ev.node(t, ax.x[li], ax.y[j], q);
    MmsPt p;
    p.cp = q[0]; p.cp_t = q[1]; p.cp_x = q[2]; p.cp_y = q[3];
    p.T = q[4]; p.T_t = q[5]; p.T_x = q[6]; p.T_y = q[7]; p.T_l = q[8];
    p.cl = q[9]; p.cl_t = q[10]; p.cl_x = q[11]; p.cl_y = q[12]; p.cl_l = q[13];
    p.cd = q[14]; p.cd_t = q[15]; p.cd_x = q[16]; p.cd_y = q[17]; p.cd_l = q[18];
    p.cs = q[19]; p.cs_t = q[20];
    const size_t c = at(g, li, j);
    const double gain = reaction_gain<REACT>(m, p.cs);
    out.v[1][c] = mms_fT(m, p);
    out.v[2][c] = mms_fcl(m, p);
    out.v[3][c] = mms_fcd_gain(m, p, gain);
    out.v[4][c] = mms_fcs_gain(m, p, gain);
    double fcp = 0.0;
    if (is_interior(g, li, j)) {
        const double w[3] = {5.0 / 9.0, 8.0 / 9.0, 5.0 / 9.0};
        double acc = 0.0;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double u[kProgCellOutputs];
                ev.cell(t, ax.xq[a * ax.nrows + li], ax.yq[b * ax.ld + j], u);
                const double cp = u[0], cp_t = u[1], cl = u[2], T = u[3];
                acc += w[a] * w[b] * (cp_t - (-cp * (m.K1 * (1.0 + cl) + m.K2 * T)));
            }
        fcp = 0.25 * acc;
    }
    out.v[0][c] = fcp;
}

// the interpreting node functors
struct ProgExactFn {
    GridDev g;
    MmsProg P;
    ProgAxes ax;
    double t;
    Out5 out;
    int what;
    TEDD_HD void operator()(int li, int j, double* regs, int stride) const {
        prog_exact_node(ProgInterp{P, P, regs, stride}, g, ax, t, out, what, li, j);
    }
};
template <int REACT>
struct ProgForcingFn {
    GridDev g;
    ModelC m;
    MmsProg node, cell;
    ProgAxes ax;
    double t;
    Out5 out;
    TEDD_HD void operator()(int li, int j, double* regs, int stride) const {
        prog_forcing_node<REACT>(ProgInterp{node, cell, regs, stride}, g, m, ax, t, out, li, j);
    }
};

}  // namespace tedd
