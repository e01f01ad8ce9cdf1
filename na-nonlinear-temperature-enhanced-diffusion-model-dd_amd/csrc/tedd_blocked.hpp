// LDS-blocked Newton solve for gfx950: the right-hand side, the stencil rows and up to H Jacobi
// sweeps of one 5-diagonal system (2 I - dt J_v) delta = rhs_v in ONE launch.
//
// A 256-thread workgroup owns a 64 x 64 tile of nodes (64 KB of LDS holds the two Jacobi
// iterates of the tile).  Each thread keeps, in registers, the scaled row of the matrix for its
// 16 nodes:  delta_c <- b + ce*delta_e + cw*delta_w + cn*delta_n + cs*delta_s  with b = rhs/diag
// and c* = (dt J_*)/diag.  Every sweep is then 4 LDS reads, 4 FMAs and 1 LDS write per node and
// one workgroup barrier; a sweep invalidates one ring of the tile, so after s sweeps the nodes at
// distance >= s from the tile edge hold exactly the s-th global Jacobi iterate.  Only the
// (64-2H)^2 interior is written back: HBM sees the inputs once and the result once per H sweeps
// instead of 4-7 arrays per sweep.  Lanes map to consecutive columns j, so all four neighbour
// reads are conflict-free row reads of the LDS tile.
//
// Launch chain for S sweeps: FIRST computes rhs and delta^1 = b and does up to H sweeps;
// middle launches reload (rhs, delta) and continue; LAST adds delta to the lagged unknown,
// applies the boundary rule and records max|delta^S - delta^(S-1)|, max|delta^S|.
//
// Replaces reference src/prob1base.py:2998-3115 (Newton steps incl. assembly :601-710 and
// spsolve :2088-2130) together with tedd_nodes.hpp's NewtonRhsFn / JacobiFn.
#pragma once
#include <hip/hip_runtime.h>

#include "tedd_nodes.hpp"

namespace tedd {

constexpr int kBT = 64;           // tile edge (nodes), halo included
constexpr int kBNPT = kBT / 4;    // nodes per thread (rows r = wave + 4 k)

template <int VAR, int CASE, int REACT, bool FIRST, bool LAST>
__global__ void __launch_bounds__(256) k_solve_blocked(const BlockedArgs A) {
    __shared__ double lds[2 * kBT * kBT];  // [2][kBT][kBT]: the two Jacobi iterates of the tile (64 KB)
    double(*buf)[kBT][kBT] = reinterpret_cast<double(*)[kBT][kBT]>(lds);
    const int lane = threadIdx.x, wave = threadIdx.y;
    const GridDev& g = A.g;
    const int H = A.H, TI = kBT - 2 * H;

    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2); give each XCD
    // a contiguous run of tiles so that neighbouring tiles' halos hit the same L2.
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb / 8, rem = nb % 8, xcd = bid % 8;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    const int ty = t / A.tiles_x, tx = t % A.tiles_x;
    const int row_base = A.r0 + ty * TI - H;  // local row of tile row 0
    const int col_base = tx * TI - H;
    const int j = col_base + lane;

    double b[kBNPT], ce[kBNPT], cw[kBNPT], cn[kBNPT], cs[kBNPT];
    const double* uo = (VAR == VAR_T) ? A.a.To : (VAR == VAR_CL) ? A.a.clo : A.a.cdo;

#pragma unroll
    for (int k = 0; k < kBNPT; ++k) {
        const int r = wave + 4 * k;
        const int li = row_base + r;
        double bb = 0.0, e = 0.0, w = 0.0, n = 0.0, s = 0.0, d0 = 0.0;
        const bool in_array = (li >= 0 && li < g.nrows && j >= 0 && j < g.ncols);
        const bool active = in_array && li >= 1 && li <= g.nrows - 2 && is_interior(g, li, j);
        if (active) {
            const size_t c = at(g, li, j);
            double rv;
            if (FIRST)
                rv = newton_rhs_value<VAR, CASE, REACT>(g, A.m, A.tb, A.f1, A.a, A.dt, li, j);
            else
                rv = A.rhs_in[c];
            double diag, oe, ow, on, os;
            newton_row<VAR, REACT>(A.m, g, A.a, li, j, A.dt, diag, oe, ow, on, os);
            const double inv = tedd_div(1.0, diag);
            bb = rv * inv;
            e = oe * inv;
            w = ow * inv;
            n = on * inv;
            s = os * inv;
            if (FIRST) {
                d0 = bb;  // delta^1 = rhs/diag (one sweep from delta^0 = 0)
                if (!LAST && A.rhs_out != nullptr) {
                    // rhs is needed again by the next launch of the chain
                    const bool mine = (r >= H && r < kBT - H && lane >= H && lane < kBT - H && li >= A.r0 && li < A.r1);
                    if (mine) A.rhs_out[c] = rv;
                }
            } else {
                d0 = A.d_in[c];
            }
        }
        b[k] = bb;
        ce[k] = e;
        cw[k] = w;
        cn[k] = n;
        cs[k] = s;
        buf[0][r][lane] = d0;
    }
    __syncthreads();

    int cur = 0;
    for (int sw = 1; sw <= A.nsweeps; ++sw) {
        const int nxt = cur ^ 1;
        const bool lane_on = (lane >= sw && lane < kBT - sw);
#pragma unroll
        for (int k = 0; k < kBNPT; ++k) {
            const int r = wave + 4 * k;
            if (r >= sw && r < kBT - sw) {  // wave-uniform
                if (lane_on) {
                    const double v = b[k] + ce[k] * buf[cur][r + 1][lane] + cw[k] * buf[cur][r - 1][lane] +
                                     cn[k] * buf[cur][r][lane + 1] + cs[k] * buf[cur][r][lane - 1];
                    buf[nxt][r][lane] = v;
                }
            }
        }
        __syncthreads();
        cur = nxt;
    }

    // write back the tile interior
    double up_max = 0.0, mag_max = 0.0;
    const bool lane_in = (lane >= H && lane < kBT - H && j >= 0 && j < g.ncols);
#pragma unroll
    for (int k = 0; k < kBNPT; ++k) {
        const int r = wave + 4 * k;
        const int li = row_base + r;
        if (r >= H && r < kBT - H && li >= A.r0 && li < A.r1 && lane_in) {
            const size_t c = at(g, li, j);
            const double dn = buf[cur][r][lane];
            if (LAST) {
                if (is_interior(g, li, j)) {
                    A.u_out[c] = uo[c] + dn;
                    const double up = (A.nsweeps > 0) ? fabs(dn - buf[cur ^ 1][r][lane]) : fabs(dn);
                    const double mag = fabs(dn);
                    up_max = (up > up_max || up != up) ? up : up_max;
                    mag_max = (mag > mag_max || mag != mag) ? mag : mag_max;
                } else {
                    A.u_out[c] = (VAR == VAR_T) ? 0.0 : uo[c];
                }
            } else {
                A.d_out[c] = dn;
            }
        }
    }
    if (LAST) {
        __syncthreads();  // everyone is done reading the iterates: reuse the first 8 doubles
        double* red = lds;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o1 = __shfl_xor(up_max, off, 64), o2 = __shfl_xor(mag_max, off, 64);
            up_max = (o1 > up_max || o1 != o1) ? o1 : up_max;
            mag_max = (o2 > mag_max || o2 != o2) ? o2 : mag_max;
        }
        if (lane == 0) {
            red[2 * wave] = up_max;
            red[2 * wave + 1] = mag_max;
        }
        __syncthreads();
        if (wave == 0 && lane < 2) {
            double v = red[lane];
            for (int w2 = 1; w2 < 4; ++w2) {
                const double o = red[2 * w2 + lane];
                v = (o > v || o != o) ? o : v;
            }
            if (v != v) v = INFINITY;
            atomicMax(A.stats + lane, (unsigned long long)__double_as_longlong(v));
        }
    }
}

}  // namespace tedd
