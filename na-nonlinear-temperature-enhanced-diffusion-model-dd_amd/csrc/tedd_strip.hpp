// Register-resident Newton solve for gfx950: the stencil rows and up to H Jacobi sweeps of one
// 5-diagonal system (2 I - dt J_v) delta = rhs_v in ONE launch, with the iterate kept in VGPRs.
//
// A workgroup owns a tile of 64 columns x W*R rows (see StripShape for the three shapes).  Lanes map to consecutive columns j; each wave owns a
// strip of R consecutive rows, so a thread holds an R-node column segment: the iterate delta[R] and
// the scaled matrix row (b, ce, cw, cn, cs)[R] with b = rhs/diag and c* = (dt J_*)/diag.  One sweep
// delta_c <- b + ce*delta_e + cw*delta_w + cn*delta_n + cs*delta_s  takes the row neighbours
// (i -/+ 1) from the thread's own registers, the column neighbours (j -/+ 1) from the adjacent
// lanes with DPP wave shifts (no LDS traffic), and only the two seam rows of every strip cross
// waves through a few KB of LDS and one barrier per sweep.
//
// The T system (constant conductivity: off-diagonals depend on the mesh only) keeps just b and
// 1/diag per node (<= 128 VGPRs, 4 tiles in flight per CU; the per-row off-diagonals are read from
// LDS as broadcasts); the cl and cd systems
// keep the full scaled row (<= 256 VGPRs, 2 tiles in flight per CU).
//
// Every node of the tile is updated in every sweep, with zeros standing in for the neighbours
// outside the tile, so each sweep spoils one more ring of the tile: after s sweeps the nodes at
// distance >= s from the tile edge hold exactly the s-th global Jacobi iterate (their dependency
// cone lies inside the tile) and only the interior (H rows / columns in from every edge) is written back.  All global loads of the set-up are unconditional (clamped coordinates, results masked) so
// that they are issued back to back instead of one dependent round trip per node, and every face
// coefficient is evaluated once and shared with the neighbouring node (registers for the x faces,
// a lane shift for the y faces).
//
// Launch chain for S sweeps: b = delta^1 = rhs/diag comes from NewtonRhsFn; each launch reloads
// (b, delta) and continues; LAST adds delta to the lagged unknown, applies the
// boundary rule and records max|delta^S - delta^(S-1)|, max|delta^S|.
//
// Replaces reference src/prob1base.py:2998-3115 (Newton steps incl. assembly :601-710 and
// spsolve :2088-2130).
#pragma once
#include <hip/hip_runtime.h>

#include "tedd_nodes.hpp"

namespace tedd {

constexpr int kSTile = 64;  // tile width (nodes = lanes), halo included


// SHAPE_LARGE: 4 waves x 16 rows (64-row tiles): least halo overhead per node; the T system on large
//   meshes (deep chains of 10 sweeps, lean rows that fit 4 waves per SIMD).
// SHAPE_MID: 12 waves x 4 rows (48-row tiles): a quarter of the registers per thread, so 6 waves per SIMD
//   instead of 2 for the full matrix rows of the cl and cd systems (their chains are 4-5 sweeps deep, so
//   the shorter tile costs little): 4.52 -> 3.74 ms (cl) and 5.39 -> 4.08 ms (cd) at 16384^2.
// SHAPE_SMALL: 8 waves x 4 rows (32-row tiles): meshes with too few large tiles to occupy the compute
//   units, where the latency of one tile is the run time (a quarter of the set-up and sweep work per wave,
//   more and smaller workgroups): 9.6k -> 14.7k steps/s at 256^2.
// SHAPE_TINY: 8 waves x 2 rows (16-row tiles), cl and cd systems on meshes with fewer small tiles than compute
//   units.  What a launch takes there is the chain of one thread (its loads, the faces and rows it sets up, its
//   share of every sweep): two rows per thread instead of four shorten it, on twice as many compute units
//   (256^2, in-kernel time of the cd launch with its right-hand side 11.7 -> 7.9 us, cl 8.9 -> 6.5).  The T
//   system keeps the 32-row tiles: its deep chains leave no interior in a 16-row tile, and its time is the
//   per-sweep exchange (0.37 us a sweep, the same with 16 waves x 2 rows: measured, 8.2 -> 8.7 us).
// SHAPE_OCT: 8 waves x 8 rows (64-row tiles), the T system on large meshes.  The tile of SHAPE_LARGE -- the same
//   halo overhead per node -- with half the rows per thread: 78 registers instead of 127, so six waves per SIMD
//   instead of four, and eight instead of sixteen per-row lane masks and row offsets in scalar registers (56
//   instead of 373 instructions that move scalars in and out of vector lanes), for twice the seam rows through
//   LDS.  The instructions executed per node are the same (389); what changes is how many waves are there to
//   issue them.  Warm 7-sweep chain at 16384^2, alternated inside one context: 3.42 -> 2.98 ms; 16 waves x 4
//   rows were measured too: 3.88.
// All shapes perform the same arithmetic on every node they write: results do not depend on the shape.
enum : int { SHAPE_LARGE = 0, SHAPE_SMALL = 1, SHAPE_MID = 2, SHAPE_TINY = 3, SHAPE_OCT = 4 };

template <int VAR, int SHAPE>
struct StripShape {
    static constexpr int R = (SHAPE == SHAPE_LARGE) ? 16 : (SHAPE == SHAPE_OCT) ? 8 : (SHAPE == SHAPE_TINY) ? 2 : 4;  // rows per thread
    static constexpr int W = (SHAPE == SHAPE_LARGE) ? 4 : (SHAPE == SHAPE_MID) ? 12 : 8;  // waves per workgroup
    static constexpr int TR = R * W;                            // tile height (nodes), halo included
    // waves per SIMD the register budget is set for
    static constexpr int OCC = (SHAPE != SHAPE_LARGE) ? 6 : (VAR == VAR_T) ? 4 : 2;
    // ... with the right-hand side fused in (more live columns during the set-up)
    static constexpr int OCC_FUSED = (SHAPE == SHAPE_MID) ? 3 : (VAR == VAR_CD) ? 2 : 3;
};
__host__ __device__ inline int strip_tile_rows(int shape) {
    return shape == SHAPE_TINY ? 16 : shape == SHAPE_SMALL ? 32 : shape == SHAPE_MID ? 48 : 64;  // (SHAPE_LARGE, SHAPE_OCT)
}

// column halo of a launch of depth H: H rounded up to even, so that the row segments a wave reads and
// the interior it writes stay 16-byte aligned.  (A fixed 8-node halo would keep them on 64-byte lines,
// but measured slower: the tile interior it gives up costs more than the split lines.)
__host__ __device__ inline int strip_col_halo(int H) { return (H + 1) & ~1; }

// 1/x without branches: hardware estimate, two Newton steps and a final residual correction
// (<= 1 ulp for normal x of moderate magnitude; a zero, huge or tiny x gives inf/NaN, which the
// convergence record of the solve then reports).  The diagonals 2 - dt J_c it is used on are >= 2
// whenever the step is well posed.
__device__ __forceinline__ double rcp_straight(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

// value of v held by the lane below / above (column j-1 / j+1); 0 at the end of the wave
__device__ __forceinline__ double lane_below(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_above(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Maximum / minimum over the 64 lanes of a wave, for values that are not NaN, returned in every lane.  Six
// steps of lane moves inside the vector unit (DPP: the pairs of a quad, the quads of 8 lanes, the halves of a
// row of 16, then lane 15 and lane 31 into the rows above) with one v_max_f64 / v_min_f64 each: about 20
// instructions and no LDS traffic.  The same reduction as a butterfly of __shfl_xor costs six dependent trips
// through the LDS crossbar with NaN-aware selects around them: ten of those at the end of the cs corrector's
// record pass took 3.9 of its 9.8 microseconds on a 256^2 mesh.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);  // (lanes the row mask leaves out keep v)
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_max_nn(double v) {
    v = fmax(v, dpp_move_f64<0xB1, 0xf>(v));   // quad_perm [1,0,3,2]
    v = fmax(v, dpp_move_f64<0x4E, 0xf>(v));   // quad_perm [2,3,0,1]
    v = fmax(v, dpp_move_f64<0x141, 0xf>(v));  // row_half_mirror
    v = fmax(v, dpp_move_f64<0x140, 0xf>(v));  // row_mirror
    v = fmax(v, dpp_move_f64<0x142, 0xa>(v));  // row_bcast:15 into rows 1 and 3
    v = fmax(v, dpp_move_f64<0x143, 0xc>(v));  // row_bcast:31 into rows 2 and 3
    return lane63_f64(v);
}
__device__ __forceinline__ double wave_min_nn(double v) {
    v = fmin(v, dpp_move_f64<0xB1, 0xf>(v));
    v = fmin(v, dpp_move_f64<0x4E, 0xf>(v));
    v = fmin(v, dpp_move_f64<0x141, 0xf>(v));
    v = fmin(v, dpp_move_f64<0x140, 0xf>(v));
    v = fmin(v, dpp_move_f64<0x142, 0xa>(v));
    v = fmin(v, dpp_move_f64<0x143, 0xc>(v));
    return lane63_f64(v);
}
// Sum over the 64 lanes in one fixed order (the same on every run: pairs of a quad, quads of 8 lanes, halves of
// a row, rows 0+1 and 2+3, then the two halves of the wave), returned in every lane.
__device__ __forceinline__ double wave_sum_fixed(double v) {
    v += dpp_move_f64<0xB1, 0xf>(v);
    v += dpp_move_f64<0x4E, 0xf>(v);
    v += dpp_move_f64<0x141, 0xf>(v);
    v += dpp_move_f64<0x140, 0xf>(v);
    {  // lane 15 of rows 0 and 2 into rows 1 and 3 (the other rows add 0)
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);
        v += __hiloint2double(hi, lo);
    }
    {  // lane 31 into rows 2 and 3
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false);
        v += __hiloint2double(hi, lo);
    }
    return lane63_f64(v);
}
// ... with a NaN anywhere reported as +inf (what every record of a maximum stores for it)
__device__ __forceinline__ double wave_max_nan_inf(double v) { return wave_max_nn(v != v ? INFINITY : v); }

// WARM: first launch of a warm-started chain (StripArgs::u_prev, gscale, stats_first); a separate
// instantiation so that the launches from zero iterates carry none of its registers and branches.
// FUSED (cl and cd systems, the only launch of a chain, 4-row strips): the launch evaluates the right-hand
// side of the Newton system itself -- b = rhs/diag by the arithmetic of NewtonRhsFn, from the column
// segments of cp*, the lagged T and unknown and the new T that the thread holds anyway (row neighbours in
// registers, column neighbours by lane shifts) -- instead of reading b from an array a launch of its own
// wrote: five (cl) / seven (cd) array passes and a launch fewer per solve.
// XFACE: the matrix rows are built from face diffusivities evaluated exactly as the right-hand-side kernel
// evaluates them (always so with FUSED, which shares them between right-hand side and rows).  Without it the
// cl rows use Dl at a face = Dl_max e_a e_b with one exponential e = exp(-phi_l c / 2) per node, and the cd
// rows law_Dd: cheaper, equal to rounding, and what the launches that read b from an array use; the exact
// variant of those launches exists so that a fused launch can be checked against them bit for bit.
// CORR (cd chain of RegHCsTriple with FUSED, its last launch): the launch carries the corrector of cp and the
// record pass of the cs corrector for the nodes it writes (the arithmetic of k_cs_newton's record pass on the
// same values: the new T, cl and cd are in the thread's registers), one launch fewer per step on small slabs.
constexpr int kStripCsRegs = 8;  // cs Newton iterations a fused corrector keeps the record of (as k_cs_newton's registers)
template <int VAR, int REACT, bool LAST, int SHAPE, bool WARM = false, bool FUSED = false, bool XFACE = false, bool CORR = false>
__global__ void __launch_bounds__((64 * StripShape<VAR, SHAPE>::W))
__attribute__((amdgpu_waves_per_eu((FUSED ? StripShape<VAR, SHAPE>::OCC_FUSED : StripShape<VAR, SHAPE>::OCC))))
k_solve_strip(const StripArgs A) {
    static_assert(!CORR || (FUSED && VAR == VAR_CD && REACT == REACT_REG && !WARM), "fused corrector: cd chain of RegHCsTriple");
    constexpr int R = StripShape<VAR, SHAPE>::R, W = StripShape<VAR, SHAPE>::W, TR = StripShape<VAR, SHAPE>::TR;
    constexpr bool LEAN = (VAR == VAR_T);
    constexpr int kBatch = (R > 4) ? 4 : R;  // rows whose late loads (warm iterate, lagged unknown) are in flight together
    static_assert(!FUSED || (LAST && !LEAN && R <= 4), "fused right-hand side: last launch of a cl/cd chain, short strips");
    __shared__ double seam[2][W][2][kSTile];  // [parity][wave][first/last row of the strip][lane]
    __shared__ double2 rowc[LEAN ? TR : 1];  // lean rows: (dt J_e, dt J_w) of each tile row
    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
    TEDD_STAMP(0);
    const GridDev& g = A.g;
    const ModelC& m = A.m;
    // (fused right-hand side: b at the two end lanes of the wave lacks a column neighbour, so the column
    // halo is one deeper than the sweeps alone need)
    const int H = A.H, HC = strip_col_halo(H + (FUSED ? 1 : 0));
    const int TIR = TR - 2 * H, TIC = kSTile - 2 * HC;

    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2); give each XCD
    // a contiguous run of tiles so that neighbouring tiles' halos hit the same L2.
    // The tiles are numbered in panels of A.panel tile rows, inside a panel down the columns first: a tile's
    // neighbour below is the next workgroup of the XCD and its neighbour to the right follows A.panel
    // workgroups later, so the rows and columns two tiles share (2H of a tile's 64 rows, re-read by the tile
    // below) are found in the L2 while both are in flight, instead of a whole tile row later.
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb / 8, rem = nb % 8, xcd = bid % 8;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    int ty, tx;
    if (A.panel > 0) {
        const int per_panel = A.panel * A.tiles_x;
        const int p = t / per_panel, u = t - p * per_panel;
        const int rows_p = min(A.panel, A.tiles_y - p * A.panel);  // (the last panel may be shorter)
        tx = u / rows_p;
        ty = p * A.panel + (u - tx * rows_p);
    } else {
        ty = t / A.tiles_x;
        tx = t % A.tiles_x;
    }
    const int row0 = A.r0 + ty * TIR - H + R * wave;  // local row of this thread's node k = 0
    const int j = tx * TIC - HC + lane;
    const int jc = min(max(j, 0), g.ncols - 1);
    const bool col_interior = (j >= 1 && j <= g.M - 1);
    const int rmax = g.nrows - 1;

    // ---- loads: clamped coordinates, no branches.  b_in holds b = rhs/diag = delta^1; the first
    // launch of a chain starts from it (d_in == b_in)
    double d[R], b[R];
    double cpv[R + 2];  // cp at rows row0-1 .. row0+R
    double Tv[R + 2];   // lagged T at the same rows (cl, cd systems)
    double clv[R], csv[R];
    // fused right-hand side: the lagged unknown and the new T at rows row0-1 .. row0+R, Y_v, the forcing
    // array and (cd) the new cl at the thread's own rows
    double uov[FUSED ? R + 2 : 1], Tnv[FUSED ? R + 2 : 1], Yv[FUSED ? R : 1], fvv[FUSED ? R : 1];
    double clnv[(FUSED && VAR == VAR_CD) ? R : 1];
    // Short strips of the T system keep the lagged unknown from here on (4 values) instead of asking for it
    // at the write-back, where nothing else is left to hide that round trip behind.
    constexpr bool UOLD_EARLY = LAST && LEAN && R <= 4 && !FUSED;
    double uold_e[UOLD_EARLY ? R : 1];
    // A warm-started chain of long strips that is a single launch reads the lagged unknown twice: for its
    // starting iterate and, after the sweeps, to add the increment to.  There are no registers to carry 16 values
    // across the sweeps, so each thread parks them in LDS (the rows the tile writes back only: dynamic shared
    // memory of (TR - 2H + 1) x 64 doubles, 26 KB at depth 7, four tiles per compute unit still fit) and reads its own
    // words back at the write-back: one array pass fewer (no barrier: a thread reads what it wrote itself).
    constexpr bool USTASH = LAST && WARM && LEAN && R > 4 && !FUSED;
    extern __shared__ double ustash[];
    // The record words this launch may raise at its very end, read now: most tiles find at the end that they do
    // not raise them, without a round trip there (a value read too early is too small: one atomic more, never
    // one fewer).
    // (the tiles of small meshes only: the others have no registers to carry it across their sweeps)
    constexpr bool STAT_EARLY = LAST && (SHAPE == SHAPE_SMALL || SHAPE == SHAPE_TINY);
    unsigned long long stat_seen = 0ull;
    if (STAT_EARLY) stat_seen = __atomic_load_n(A.stats + (lane & 1), __ATOMIC_RELAXED);
    // fused corrector: its three inputs at the thread's nodes, and the record words lane k may move at the end
    double ncpv[CORR ? R : 1], ycsv[CORR ? R : 1], cs0v[CORR ? R : 1];
    unsigned long long cs_seen_mx = 0ull, cs_seen_mn = 0xFFFFFFFFFFFFFFFFull;
    if (CORR && wave == 0 && lane < A.cs_iters) {
        cs_seen_mx = __atomic_load_n(A.gmax + lane, __ATOMIC_RELAXED);
        cs_seen_mn = __atomic_load_n(A.gmin + lane, __ATOMIC_RELAXED);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const size_t c = (size_t)min(max(row0 + k, 0), rmax) * g.ld + jc;
        if (!FUSED) b[k] = A.b_in[c];
        if (UOLD_EARLY) uold_e[k] = A.a.To[c];
        if (CORR) {
            ncpv[k] = get_once(A.ncp + c);
            ycsv[k] = get_once(A.ycs + c);
            cs0v[k] = A.cs0[c];
        }
        if (LEAN && !WARM) d[k] = A.d_in[c];  // (the same line as b[k] in the first launch of a cold chain)
        if (VAR == VAR_CD) {
            clv[k] = A.a.clo[c];
            csv[k] = A.a.cs1[c];
        }
        if (FUSED) {
            Yv[k] = A.Y[c];
            fvv[k] = A.fv[c];
            if (VAR == VAR_CD) clnv[k] = A.a.cln[c];
        }
    }
#pragma unroll
    for (int k = 0; k < R + 2; ++k) {
        if (LEAN && (k == 0 || k == R + 1)) continue;
        const size_t c = (size_t)min(max(row0 + k - 1, 0), rmax) * g.ld + jc;
        cpv[k] = A.a.cp1[c];
        if (!LEAN) Tv[k] = A.a.To[c];
        if (FUSED) {
            uov[k] = (VAR == VAR_CL ? A.a.clo : A.a.cdo)[c];
            Tnv[k] = A.a.Tn[c];
        }
    }
    const double rks = g.rk[jc], rkn = g.rk[min(jc + 1, g.ncols - 1)], rkm = g.rkm[jc];
    TEDD_STAMP_WAIT();
    TEDD_STAMP(1);

    // ---- matrix rows
    // full rows (cl, cd): ce..cs scaled by 1/diag.  Every face coefficient is evaluated once, by the
    // arithmetic of faces_Dl / faces_Dd_pair (the right-hand-side kernel's): the R+1 x faces of the
    // thread's column segment and its R y faces towards column j+1; the y face towards j-1 is the lane
    // below's.
    // lean rows (T): inv = 1/diag per node, the e/w off-diagonals per row (LDS), on/os per lane.
    double ce[LEAN ? 1 : R], cw[LEAN ? 1 : R], cn[LEAN ? 1 : R], cs[LEAN ? 1 : R];
    double inv[LEAN ? R : 1];
    double on = 0.0, os = 0.0;
    // (short strips keep the per-row off-diagonals of the T system in registers too: the LDS read they replace
    // stands at the head of every node's update)
    constexpr bool ROWC_REGS = LEAN && R <= 4;
    double rce[ROWC_REGS ? R : 1], rcw[ROWC_REGS ? R : 1];
    constexpr bool EXACT = FUSED || XFACE;
    constexpr int NF = LEAN ? 1 : 2 * R + 1;       // [0..R] x faces (between rows k-1 and k), [R+1..2R] y faces of row k
    constexpr bool WANT_DT = FUSED && VAR == VAR_CD;  // dDd/dT on the faces (coupling with the new T)
    double Df[NF], dDf[WANT_DT ? NF : 1];
    if (VAR == VAR_CL && EXACT) {
        // Dl_max exp(-phi_l c) at the face means.  One wave-uniform choice between the short series and the
        // series with the library routine patched in where a value needs it (value by value the result is
        // tedd_exp's; its per-value branch would split the set-up into dozens of basic blocks)
        bool small = true;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double cm = (f <= R) ? 0.5 * (cpv[f + 1] + cpv[f])
                                       : 0.5 * (lane_above(cpv[f - R]) + cpv[f - R]);
            Df[f] = -m.phi_l * cm;
            small = small && fabs(Df[f]) < 0x1p-10;
        }
        const bool all_small = __all(small);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const double x = Df[f];
            double E = tedd_exp_series(x);
            if (!all_small) E = (fabs(x) < 0x1p-10) ? E : exp(x);
            Df[f] = m.Dl_max * E;
        }
    } else if (VAR == VAR_CL) {
        // Dl at a face = Dl_max exp(-phi_l (c_a + c_b)/2) = Dl_max e_a e_b with e = exp(-phi_l c/2) per node:
        // one exponential per node instead of one per face (wave-uniform choice of the routine, as above)
        double ev[R + 2];
        bool small = true;
#pragma unroll
        for (int k = 0; k < R + 2; ++k) {
            ev[k] = (-0.5 * m.phi_l) * cpv[k];
            small = small && fabs(ev[k]) < 0x1p-10;
        }
        if (__all(small)) {
#pragma unroll
            for (int k = 0; k < R + 2; ++k) ev[k] = tedd_exp_series(ev[k]);
        } else {
#pragma unroll
            for (int k = 0; k < R + 2; ++k) ev[k] = exp(ev[k]);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f)
            Df[f] = (f <= R) ? m.Dl_max * (ev[f + 1] * ev[f]) : m.Dl_max * (lane_above(ev[f - R]) * ev[f - R]);
    } else if (VAR == VAR_CD) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            double cm, Tm;
            if (f <= R) {
                cm = 0.5 * (cpv[f + 1] + cpv[f]);
                Tm = 0.5 * (Tv[f + 1] + Tv[f]);
            } else {
                cm = 0.5 * (lane_above(cpv[f - R]) + cpv[f - R]);
                Tm = 0.5 * (lane_above(Tv[f - R]) + Tv[f - R]);
            }
            if (WANT_DT)
                law_Dd_pair(m, cm, Tm, Df[f], dDf[f]);
            else if (EXACT)
                Df[f] = law_Dd_p(m, cm, Tm);
            else
                Df[f] = law_Dd(m, cm, Tm);
        }
    }
    auto is_active = [&](int li) {  // nodes the system has an unknown at
        const int i = g.i0 + li;
        return col_interior && li >= 1 && li <= g.nrows - 2 && i >= 1 && i <= g.N - 1;
    };
    // (short strips: the row metrics of all R rows are requested together, ahead of the loop that uses them
    // one row at a time)
    constexpr bool GEO_AHEAD = (R <= 4);
    double rhv[GEO_AHEAD ? R + 1 : 1], rhmv[GEO_AHEAD ? R : 1];
    if (GEO_AHEAD) {
#pragma unroll
        for (int k = 0; k <= R; ++k) rhv[GEO_AHEAD ? k : 0] = g.rh[min(max(row0 + k, 0), rmax)];
#pragma unroll
        for (int k = 0; k < R; ++k) rhmv[GEO_AHEAD ? k : 0] = g.rhm[min(max(row0 + k, 0), rmax)];
        loads_done();
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int li = row0 + k;
        const int lic = min(max(li, 0), rmax);
        // (rhv[k + 1] is g.rh[min(lic + 1, rmax)] at every row that has an unknown)
        const Geo ge = GEO_AHEAD ? Geo{rhv[GEO_AHEAD ? k : 0], rhv[GEO_AHEAD ? k + 1 : 0], rhmv[GEO_AHEAD ? k : 0], rks, rkn, rkm}
                                 : Geo{g.rh[lic], g.rh[min(lic + 1, rmax)], g.rhm[lic], rks, rkn, rkm};
        const bool active = is_active(li);
        Coef5 r;
        if (LEAN) {
            r = jac_T_FT(m, ge, cpv[k + 1]);
            const double iv = tedd_div(1.0, 2.0 - A.dt * r.c);
            inv[k] = active ? iv : 0.0;
            if (ROWC_REGS) {
                rce[ROWC_REGS ? k : 0] = A.dt * r.e;
                rcw[ROWC_REGS ? k : 0] = A.dt * r.w;
            } else if (lane == 0) {
                rowc[R * wave + k] = make_double2(A.dt * r.e, A.dt * r.w);  // read back by this wave only
            }
            on = A.dt * r.n;
            os = A.dt * r.s;
        } else {
            const double Dn = Df[R + 1 + k];
            const Face4 D{Df[k], Df[k + 1], lane_below(Dn), Dn};
            if (VAR == VAR_CL) {
                St5 T;
                T.c = 0.0; T.n = 0.0; T.s = 0.0;
                T.e = Tv[k + 2];
                T.w = Tv[k];
                r = jac_cl_Fcl(m, ge, D, T, cpv[k + 1], 1.0, 1.0);
            } else {
                r = jac_cd_Fcd<REACT>(m, ge, D, clv[k], csv[k]);
                const double tmp = r.w;  // reference quirk: W and S coefficients exchanged in the cd matrix
                r.w = r.s;
                r.s = tmp;
            }
            const double diag = 2.0 - A.dt * r.c;
            if (FUSED) {
                // right-hand side at this node (newton_rhs_from_faces: the arithmetic of NewtonRhsFn)
                const int i = g.i0 + li;
                const double w_ok = (i > 1) ? 1.0 : 0.0, e_ok = (i < g.N - 1) ? 1.0 : 0.0;
                St5 T, u, Tn;
                T.c = Tv[k + 1]; T.e = Tv[k + 2]; T.w = Tv[k];
                u.c = uov[k + 1]; u.e = uov[k + 2]; u.w = uov[k];
                u.n = lane_above(uov[k + 1]);
                u.s = lane_below(uov[k + 1]);
                Tn.c = Tnv[k + 1]; Tn.e = Tnv[k + 2]; Tn.w = Tnv[k];
                Face4 dD{};
                if (VAR == VAR_CD) {
                    T.n = lane_above(Tv[k + 1]);
                    T.s = lane_below(Tv[k + 1]);
                    Tn.n = lane_above(Tnv[k + 1]);
                    Tn.s = lane_below(Tnv[k + 1]);
                    const double dDn = dDf[WANT_DT ? R + 1 + k : 0];
                    dD = Face4{dDf[WANT_DT ? k : 0], dDf[WANT_DT ? k + 1 : 0], lane_below(dDn), dDn};
                } else {
                    T.n = 0.0; T.s = 0.0; Tn.n = 0.0; Tn.s = 0.0;
                }
                const double rr = newton_rhs_from_faces<VAR, REACT>(
                    m, ge, A.dt, fvv[k], Yv[k], w_ok, e_ok, D, dD, cpv[k + 1], T, u, Tn, (VAR == VAR_CD) ? clv[k] : 0.0,
                    (VAR == VAR_CD) ? csv[k] : 0.0, (VAR == VAR_CD) ? clnv[k] : 0.0);
                b[k] = tedd_div(rr, diag);
            }
            const double iv = rcp_straight(diag);
            ce[k] = active ? (A.dt * r.e) * iv : 0.0;
            cw[k] = active ? (A.dt * r.w) * iv : 0.0;
            cn[k] = active ? (A.dt * r.n) * iv : 0.0;
            cs[k] = active ? (A.dt * r.s) * iv : 0.0;
        }
        b[k] = active ? b[k] : 0.0;
        if (LEAN && !WARM) d[k] = active ? d[k] : 0.0;
    }
    // full rows: the iterate is loaded last, until here the registers hold the inputs of the rows.
    // (The empty asm ties the load addresses to the end of the set-up so that the loads are not hoisted.)
    if (WARM) {  // warm start: the increment of the time step before
        const double* uo_ = (VAR == VAR_T) ? A.a.To : (VAR == VAR_CL) ? A.a.clo : A.a.cdo;
        int jd = jc;
        asm volatile("" : "+v"(jd) : "v"(b[R - 1]));
        // (a batch of loads first, unconditionally, then the masking: a select next to its load is compiled
        // into a branch around the load, and the loads of a thread then wait for each other, one round trip
        // each; batches of kBatch rows keep the loaded values within the register budget)
        // (parked values: byte offset of the thread's row k = 0 in the stash; rows the tile does not write back go
        // to a spare row -- a select, no branch that would split the batch of loads)
        const int sbase = (R * wave - H) * (kSTile * 8) + lane * 8;
        const int sdummy = TIR * (kSTile * 8) + lane * 8;
#pragma unroll
        for (int k0 = 0; k0 < R; k0 += kBatch) {
#pragma unroll
            for (int k = k0; k < k0 + kBatch; ++k) {
                const size_t c = (size_t)min(max(row0 + k, 0), rmax) * g.ld + jd;
                const double uo_v = FUSED ? uov[FUSED ? k + 1 : 0] : uo_[c];
                d[k] = uo_v - A.u_prev[c];
                if (USTASH) {
                    const int r = R * wave + k;
                    const int off = (r >= H && r < TR - H) ? sbase + k * (kSTile * 8) : sdummy;
                    *reinterpret_cast<double*>(reinterpret_cast<char*>(ustash) + off) = uo_v;
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = k0; k < k0 + kBatch; ++k) d[k] = is_active(row0 + k) ? A.gscale * d[k] : 0.0;
        }
    } else if (LEAN) {
    } else if (FUSED || A.d_in == A.b_in) {
#pragma unroll
        for (int k = 0; k < R; ++k) d[k] = b[k];
    } else {
        int jd = jc;
        asm volatile("" : "+v"(jd) : "v"(b[R - 1]));
#pragma unroll
        for (int k = 0; k < R; ++k) d[k] = A.d_in[(size_t)min(max(row0 + k, 0), rmax) * g.ld + jd];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < R; ++k) d[k] = is_active(row0 + k) ? d[k] : 0.0;
    }

    TEDD_STAMP_WAIT();
    TEDD_STAMP(2);
    // which of this thread's nodes are written back
    const bool lane_in = (lane >= HC && lane < kSTile - HC && j >= 0 && j < g.ncols);
    double up_max = 0.0, mag_max = 0.0, u_max = 0.0;

    // ---- sweeps.  The last launch of a chain keeps its final sweep for the write-back below.
    int par = 0;
    double above = 0.0, below = 0.0;
    auto exchange_seams = [&]() {
        seam[par][wave][0][lane] = d[0];
        seam[par][wave][1][lane] = d[R - 1];
        __syncthreads();
        above = seam[par][max(wave - 1, 0)][1][lane];      // row row0-1 (unused for wave 0)
        below = seam[par][min(wave + 1, W - 1)][0][lane];  // row row0+R (unused for the last wave)
        par ^= 1;
    };
    // (explicit fused multiply-adds in a fixed order: left to the compiler, the contraction of these sums
    // differs between instantiations of the kernel, and the tile shapes would no longer agree bit for bit)
    auto update = [&](int k, double prev, double cur, double nxt) {
        if (LEAN) {
            const double2 rc = ROWC_REGS ? make_double2(rce[ROWC_REGS ? k : 0], rcw[ROWC_REGS ? k : 0]) : rowc[R * wave + k];
            double acc = rc.y * prev;
            acc = fma(rc.x, nxt, acc);
            acc = fma(on, lane_above(cur), acc);
            acc = fma(os, lane_below(cur), acc);
            return fma(inv[k], acc, b[k]);
        }
        double acc = fma(ce[k], nxt, b[k]);
        acc = fma(cw[k], prev, acc);
        acc = fma(cn[k], lane_above(cur), acc);
        return fma(cs[k], lane_below(cur), acc);
    };
    // The last launch of a chain runs its final sweep apart from the others (it alone keeps the record
    // max|delta^S - delta^(S-1)|), in place like them: fused with the store it would put each row's load of
    // the lagged unknown between its update and its store, one memory round trip per row.
    // (full rows of 16 nodes have no registers for a second copy of the sweep: their final sweep stays in the
    // loop, told apart by a flag)
    constexpr bool PEEL_FINAL = LAST && (LEAN || R <= 4);
    const int plain_end = PEEL_FINAL ? max(A.nsweeps - 1, 0) : A.nsweeps;
    int sw0 = 1;
    if (WARM && plain_end >= 1) {
        // first sweep of a warm chain (a warm chain performs at least two): records max|delta^1 - delta^0|
        // over the rows and columns this launch writes back (nodes without an unknown contribute 0), reduced
        // per wave right away so that nothing of it stays live across the other sweeps
        exchange_seams();
        double first_max = 0.0;
        double prev = above;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double cur = d[k];
            const double v = update(k, prev, cur, (k < R - 1) ? d[k + 1] : below);
            d[k] = v;
            const int r = R * wave + k, li = row0 + k;
            const double up = (lane_in && r >= H && r < TR - H && li >= A.r0 && li < A.r1) ? fabs(v - cur) : 0.0;
            first_max = (up > first_max || up != up) ? up : first_max;
            prev = cur;
        }
        first_max = wave_max_nan_inf(first_max);
        if (lane == 0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(first_max);
            if (bits > __atomic_load_n(A.stats_first, __ATOMIC_RELAXED)) atomicMax(A.stats_first, bits);
        }
        sw0 = 2;
    }
    for (int sw = sw0; sw <= plain_end; ++sw) {
        exchange_seams();
        const bool track = LAST && !PEEL_FINAL && sw == A.nsweeps;
        double prev = above;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double cur = d[k];
            const double v = update(k, prev, cur, (k < R - 1) ? d[k + 1] : below);
            d[k] = v;
            if (LAST && !PEEL_FINAL && track) {
                const int r = R * wave + k, li = row0 + k;
                const bool counted = lane_in && r >= H && r < TR - H && li >= A.r0 && li < A.r1 && is_interior(g, li, j);
                const double up = counted ? fabs(v - cur) : 0.0;
                up_max = (up > up_max || up != up) ? up : up_max;
            }
            prev = cur;
        }
    }
    if (PEEL_FINAL && A.nsweeps >= 1) {
        exchange_seams();
        double prev = above;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double cur = d[k];
            const double v = update(k, prev, cur, (k < R - 1) ? d[k + 1] : below);
            d[k] = v;
            const int r = R * wave + k, li = row0 + k;
            const bool counted = lane_in && r >= H && r < TR - H && li >= A.r0 && li < A.r1 && is_interior(g, li, j);
            const double up = counted ? fabs(v - cur) : 0.0;
            up_max = (up > up_max || up != up) ? up : up_max;
            prev = cur;
        }
    }

    TEDD_STAMP(3);
    // ---- write back the tile interior.  The coordinates are laundered through empty asm so that
    // the store addresses are rebuilt here instead of being kept in registers across the sweeps.
    const double* uo = (VAR == VAR_T) ? A.a.To : (VAR == VAR_CL) ? A.a.clo : A.a.cdo;
    int row0w = row0, jw = j;
    asm volatile("" : "+s"(row0w));
    asm volatile("" : "+v"(jw));
    int wave_w = wave, lane_w = lane;
    asm volatile("" : "+s"(wave_w));
    asm volatile("" : "+v"(lane_w));
    double cs_rmx[CORR ? kStripCsRegs : 1], cs_rmn[CORR ? kStripCsRegs : 1];
    int cs_first_bad = 0x7fffffff;
    if (CORR) {
#pragma unroll
        for (int q = 0; q < kStripCsRegs; ++q) {
            cs_rmx[CORR ? q : 0] = 0.0;
            cs_rmn[CORR ? q : 0] = INFINITY;
        }
    }
    if (LAST) {
        // the lagged unknown of a batch of rows first (clamped coordinates, unconditional), then the stores
#pragma unroll
        for (int k0 = 0; k0 < R; k0 += kBatch) {
            double uold[kBatch];
#pragma unroll
            for (int k = k0; k < k0 + kBatch; ++k) {
                if (FUSED)
                    uold[k - k0] = uov[FUSED ? k + 1 : 0];
                else if (UOLD_EARLY)
                    uold[k - k0] = uold_e[UOLD_EARLY ? k : 0];
                else if (USTASH) {
                    // (addresses rebuilt from laundered coordinates: kept from the set-up they would be sixteen
                    // registers across the sweeps)
                    const int r = R * wave_w + k;
                    const int off = (r >= H && r < TR - H) ? (r - H) * (kSTile * 8) + lane_w * 8 : TIR * (kSTile * 8) + lane_w * 8;
                    uold[k - k0] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(ustash) + off);
                } else
                    uold[k - k0] = uo[(size_t)min(max(row0w + k, 0), rmax) * g.ld + min(max(jw, 0), g.ncols - 1)];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = k0; k < k0 + kBatch; ++k) {
                const int r = R * wave + k;
                const int li = row0w + k;
                if (r >= H && r < TR - H && li >= A.r0 && li < A.r1 && lane_in) {
                    const size_t c = at(g, li, jw);
                    const bool in = is_interior(g, li, jw);
                    double unew;
                    if (in) {
                        unew = uold[k - k0] + d[k];
                        const double mag = fabs(d[k]);
                        mag_max = (mag > mag_max || mag != mag) ? mag : mag_max;
                        if (WARM) u_max = fmax(u_max, fabs(unew));  // (size of the unknown: floor of the stopping rule)
                    } else {
                        unew = (VAR == VAR_T) ? 0.0 : uold[k - k0];
                    }
                    A.u_out[c] = unew;
                    if (CORR) {
                        const CsNodeIn v{ncpv[CORR ? k : 0], ycsv[CORR ? k : 0], cs0v[CORR ? k : 0], Tnv[FUSED ? k + 1 : 0],
                                         clnv[(FUSED && VAR == VAR_CD) ? k : 0], unew};
                        put_once(A.cp_out + c, corrector_cp_vals(m, A.dt, v, in));
                        double x, y, av;
                        cs_newton_inputs_vals(m, A.dt, v, y, av, x);
                        double ex = cs_newton_exp(x, m.eta);
#pragma unroll
                        for (int q = 0; q < kStripCsRegs; ++q) {
                            if (q < A.cs_iters) {
                                const double dx = cs_newton_step(x, y, av, m.eta, ex);
                                x = x + dx;
                                if (q + 1 < A.cs_iters) ex = cs_newton_exp_next(ex, x, dx, m.eta);
                                cs_record_update(dx, x, q, cs_rmx[CORR ? q : 0], cs_rmn[CORR ? q : 0], cs_first_bad);
                            }
                        }
                        put_once(A.cs_out + c, in ? x : 0.0 * x);
                    }
                }
            }
        }
        if (A.nsweeps == 0) up_max = mag_max;
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int r = R * wave + k;
            const int li = row0w + k;
            if (r >= H && r < TR - H && li >= A.r0 && li < A.r1 && lane_in) A.d_out[at(g, li, jw)] = d[k];
        }
    }
    TEDD_STAMP(4);
    // fused corrector: the wave's per-iteration record (lane q keeps iteration q), one set of plain LDS words per wave
    __shared__ unsigned long long cs_wrec[CORR ? W : 1][2 * kStripCsRegs];
    if (CORR) {
        double mine_mx = 0.0, mine_mn = INFINITY;
#pragma unroll
        for (int q = 0; q < kStripCsRegs; ++q) {
            if (q < A.cs_iters) {
                cs_record_finish(q, cs_rmx[CORR ? q : 0], cs_rmn[CORR ? q : 0], cs_first_bad);
                const double wmx = wave_max_nn(cs_rmx[CORR ? q : 0]), wmn = wave_min_nn(cs_rmn[CORR ? q : 0]);
                if (lane == q) {
                    mine_mx = wmx;
                    mine_mn = wmn;
                }
            }
        }
        if (lane < A.cs_iters) {
            cs_wrec[CORR ? wave : 0][lane] = (unsigned long long)__double_as_longlong(mine_mx);
            cs_wrec[CORR ? wave : 0][kStripCsRegs + lane] = (unsigned long long)__double_as_longlong(mine_mn);
        }
    }
    if (LAST) {
        __syncthreads();  // the seam rows are dead: reuse their first doubles
        if (CORR && wave == 0 && lane < A.cs_iters) {
            unsigned long long mx = cs_wrec[0][lane], mn = cs_wrec[0][kStripCsRegs + lane];
#pragma unroll
            for (int w2 = 1; w2 < (CORR ? W : 1); ++w2) {
                mx = cs_wrec[CORR ? w2 : 0][lane] > mx ? cs_wrec[CORR ? w2 : 0][lane] : mx;
                mn = cs_wrec[CORR ? w2 : 0][kStripCsRegs + lane] < mn ? cs_wrec[CORR ? w2 : 0][kStripCsRegs + lane] : mn;
            }
            if (mx > cs_seen_mx) atomicMax(A.gmax + lane, mx);
            if (mn < cs_seen_mn) atomicMin(A.gmin + lane, mn);
        }
        double* red = &seam[0][0][0][0];
        up_max = wave_max_nan_inf(up_max);
        mag_max = wave_max_nan_inf(mag_max);
        if (lane == 0) {
            red[2 * wave] = up_max;
            red[2 * wave + 1] = mag_max;
        }
        if (WARM) {
            // a single-launch warm chain also records the size of the unknown, next to its first update
            // (fmax drops NaNs, which show in the other records)
            u_max = wave_max_nn(u_max);
            if (lane == 0) red[2 * W + wave] = u_max;
        }
        __syncthreads();
        if (WARM && wave == 0 && lane == 2) {
            double v = red[2 * W];
            for (int w2 = 1; w2 < W; ++w2) v = fmax(v, red[2 * W + w2]);
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            if (bits > __atomic_load_n(A.stats_first + 1, __ATOMIC_RELAXED)) atomicMax(A.stats_first + 1, bits);
        }
        if (wave == 0 && lane < 2) {
            double v = red[lane];
            for (int w2 = 1; w2 < W; ++w2) {
                const double o = red[2 * w2 + lane];
                v = (o > v || o != o) ? o : v;
            }
            if (v != v) v = INFINITY;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
            // most tiles do not raise the maximum: skip their atomics
            if (!STAT_EARLY) stat_seen = __atomic_load_n(A.stats + lane, __ATOMIC_RELAXED);
            if (bits > stat_seen) atomicMax(A.stats + lane, bits);
        }
    }
    TEDD_STAMP_WAIT();
    TEDD_STAMP(5);
}

}  // namespace tedd
