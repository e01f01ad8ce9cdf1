// Marching launcher of the 5-point node functors for gfx950.
//
// The plain launcher (k_nodes) gives every node a thread that loads the node and its four neighbours
// of every stencil array: each row of such an array is requested three times (as row i-1, i, i+1 of
// three different workgroup rows), and whether the second and third request is served on chip is left
// to the caches -- on a mesh far larger than them a quarter of it is not.
//
// Here a wavefront owns 64 consecutive columns and marches down a segment of rows.  Lanes are columns.
// Of every stencil array the wave keeps rows i-1, i, i+1 in registers and rolls them: each row is loaded
// from memory ONCE per segment (plus the two rows that border the segment), as one 512-byte request per
// wave.  The neighbours at j-1 / j+1 come from the adjacent lanes by DPP wave shifts; only lanes 0 and 63
// need a value from outside the wave, which one two-lane load per row and array supplies.  Arrays read at
// the node alone (forcing values, cs) are loaded once, non-temporally.  W waves of a workgroup stand side
// by side (64 W columns), and workgroups are dealt to the XCDs in contiguous runs of tiles, so that the
// lines shared across wave and workgroup edges are found in the same L2.
//
// A functor F that marches provides
//   static constexpr int NS, NC;          number of stencil arrays / of arrays read at the node alone
//   const double* sptr(k), cptr(k);       those arrays (cptr may be null: reads as 0)
//   GridDev g;
//   void node(li, j, Geo, const St5 s[NS], const double cv[NC]);   the node body on register values
// and performs exactly the arithmetic of its operator()(li, j): results are bit-identical.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "tedd_nodes.hpp"

namespace tedd {

// value of the lane below / above with `edge` (the wave's own lane 0 / lane 63 copy) standing in at the
// end of the wave: lane 0 holds u[j-1] in edge, lane 63 holds u[j+1]
__device__ __forceinline__ double lane_below_or(double v, double edge) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(edge), lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_above_or(double v, double edge) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(edge), lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

constexpr int kMarchWaves = 4;  // waves side by side in a workgroup: 256 columns

// A functor may give its node body in two parts -- NodeOut node_vals(...) and node_store(li, j, NodeOut) -- and the
// look-ahead march then waits for the loads it has in flight between the two.  Why: the counter of outstanding vector
// memory operations counts loads and stores alike, in order.  The compiler waits for the next row's loads where
// their registers are first read, at the top of the next row, and because the row's stores sit behind branches it
// cannot count them: it waits for everything, the stores it has just issued included, and a store's round trip
// to memory joins every row's critical path.  With the wait placed before the stores the loads have had the whole
// row's arithmetic to arrive, and the stores have the whole next row to complete.
template <class F, class = void>
struct HasNodeVals : std::false_type {};
template <class F>
struct HasNodeVals<F, std::void_t<typename F::NodeOut>> : std::true_type {};

// OCC: waves per SIMD the register budget is set for.  AHEAD: request everything the next row needs before
// the present row is computed (more registers; for kernels that run at two waves per SIMD).
template <class F, int OCC, bool AHEAD, bool SPLIT = true>
__global__ void __launch_bounds__(64 * kMarchWaves) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
k_march(const F f, int r0, int r1, int ncols, int seg_rows, int tiles_x) {
    constexpr int NS = F::NS, NC = F::NC;
    const GridDev& g = f.g;
    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);

    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2); every XCD gets a
    // contiguous run of tiles (x fastest), so that the lines two neighbouring tiles share are found there
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb / 8, rem = nb % 8, xcd = bid % 8;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    const int ty = t / tiles_x, tx = t % tiles_x;

    const int j = (tx * kMarchWaves + wave) * 64 + lane;
    const int ra = r0 + ty * seg_rows;
    const int rb = min(ra + seg_rows, r1);
    if ((tx * kMarchWaves + wave) * 64 >= ncols) return;  // (whole wave out of the mesh)
    const bool col_ok = j < ncols;
    const unsigned jc = (unsigned)min(j, ncols - 1);  // (unsigned: 32-bit lane offsets on scalar row bases)
    const int rmax = g.nrows - 1;
    // column metrics are the same for every row of the march
    const double rks = g.rk[jc], rkn = g.rk[min(jc + 1u, (unsigned)ncols - 1u)], rkm = g.rkm[jc];
    // the one value per row and array a wave needs from outside: column j-1 for lane 0, j+1 for lane 63
    const bool edge_lane = (lane == 0 && jc >= 1u) || (lane == 63 && jc + 1u < (unsigned)ncols);
    const unsigned je = (lane == 0) ? (jc >= 1u ? jc - 1u : 0u) : jc + 1u;

    const double* sp[NS];
    const double* cp[NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) sp[k] = f.sptr(k);
#pragma unroll
    for (int k = 0; k < NC; ++k) cp[k] = f.cptr(k);

    // (row bases are wave-uniform: scalar registers; the lanes add a 32-bit column offset)
    auto row_of = [&](int li) { return (size_t)min(max(li, 0), rmax) * g.ld; };
    double w[NS], c[NS], e[NS], edge[NS], cv[NC];
    {
        const size_t ow = row_of(ra - 1), oc = row_of(ra), oe = row_of(ra + 1);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            w[k] = (sp[k] + ow)[jc];
            c[k] = (sp[k] + oc)[jc];
            e[k] = (sp[k] + oe)[jc];
            edge[k] = edge_lane ? (sp[k] + oc)[je] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < NC; ++k) cv[k] = cp[k] ? __builtin_nontemporal_load(cp[k] + oc + jc) : 0.0;
    }
    if constexpr (AHEAD && SPLIT && HasNodeVals<F>::value) {
        // the first rows are waited for here, once: a wait at the top of the loop that has to serve the first entry
        // would also be executed on every later one, where all it could wait for is the row's stores
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
    }
    for (int li = ra; li < rb; ++li) {
        // AHEAD: everything the NEXT row needs is requested before this row is computed: row li+2 of the
        // stencil arrays, the edge values and the node-only arrays of row li+1 (clamped past the end of the
        // segment: the values are then unused); the wave waits for them only when it rolls its registers.
        // Otherwise row li+1 of the stencil arrays is requested at the top of row li, which needs it.
        const size_t on = row_of(li + 1), on2 = row_of(li + 2);
        double e2[NS], edge_n[NS], cv_n[NC];
        if (AHEAD) {
#pragma unroll
            for (int k = 0; k < NS; ++k) e2[k] = (sp[k] + on2)[jc];
#pragma unroll
            for (int k = 0; k < NS; ++k) edge_n[k] = edge_lane ? (sp[k] + on)[je] : 0.0;
#pragma unroll
            for (int k = 0; k < NC; ++k) cv_n[k] = cp[k] ? __builtin_nontemporal_load(cp[k] + on + jc) : 0.0;
        } else if (li > ra) {
            const size_t oc = row_of(li);
#pragma unroll
            for (int k = 0; k < NS; ++k) e[k] = (sp[k] + on)[jc];
#pragma unroll
            for (int k = 0; k < NS; ++k) edge[k] = edge_lane ? (sp[k] + oc)[je] : 0.0;
#pragma unroll
            for (int k = 0; k < NC; ++k) cv[k] = cp[k] ? __builtin_nontemporal_load(cp[k] + oc + jc) : 0.0;
        }
        St5 s[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            s[k].c = c[k];
            s[k].e = e[k];
            s[k].w = w[k];
            s[k].n = lane_above_or(c[k], edge[k]);
            s[k].s = lane_below_or(c[k], edge[k]);
        }
        // Row metrics: one address per wave, written before the launch and never during it, read through the
        // constant address space (scalar loads).  As vector loads they would be the youngest loads of the row, and
        // the wait for them in the middle of the row's arithmetic would be a wait for the whole look-ahead batch.
        const int lic = min(li, rmax);
        const Geo ge{tab_row_entry(g.rh, lic), tab_row_entry(g.rh, min(lic + 1, rmax)), tab_row_entry(g.rhm, lic), rks, rkn, rkm};
        if constexpr (AHEAD && SPLIT && HasNodeVals<F>::value) {
            typename F::NodeOut o{};
            if (col_ok) o = f.node_vals(li, j, ge, s, cv);
            // (vmcnt 0, expcnt and lgkmcnt left alone; the builtin, so that the compiler's own bookkeeping sees it)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
            if (col_ok) f.node_store(li, j, o);
        } else {
            if (col_ok) f.node(li, j, ge, s, cv);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            w[k] = c[k];
            c[k] = e[k];
            if (AHEAD) {
                e[k] = e2[k];
                edge[k] = edge_n[k];
            }
        }
        if (AHEAD) {
#pragma unroll
            for (int k = 0; k < NC; ++k) cv[k] = cv_n[k];
        }
    }
}

// The same march with the next row's loads travelling into LDS instead of into registers.
//
// The functors that run at three or four waves per SIMD have no registers to hold a row that is still on its
// way, so in k_march they ask for row li+1 at the top of row li and wait for it there: one memory round trip per
// row, hidden only by the other waves of the SIMD.  Here a wave owns two slots of LDS and asks for everything
// row li+1 will need while it computes row li, with loads that write straight into LDS (16 bytes per lane, lane
// l at byte 16 l of the destination: global_load_lds_dwordx4): of every stencil array the 64 doubles of its
// columns in row li+2 and the two granules on either side of them (lanes 0..33: columns -2 .. 65; columns -1 and
// 64 are what lanes 0 and 63 need from outside the wave once that row is the centre, one row later, so those two
// values are carried in registers for a row); of the arrays read at the node alone, two per instruction, the 64
// doubles of row li+1.  At the top of the next row the wave
// waits for that one batch, reads it from LDS (consecutive lanes, consecutive doubles: no bank conflicts) and
// asks for the next.  Registers as in k_march without look-ahead; LDS 2 x (68 NS + 64 NC) doubles per wave.
// Same node bodies on the same values: bit-identical results.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;

template <class F, int OCC, bool SPLIT = true>
__global__ void __launch_bounds__(64 * kMarchWaves) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
k_march_lds(const F f, int r0, int r1, int ncols, int seg_rows, int tiles_x) {
    constexpr int NS = F::NS, NC = F::NC;
    constexpr int kSlot = 68 * NS + 64 * NC;  // doubles
    __shared__ __attribute__((aligned(16))) double ring[kMarchWaves][2][kSlot];
    const GridDev& g = f.g;
    const int lane = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);

    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb / 8, rem = nb % 8, xcd = bid % 8;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    const int ty = t / tiles_x, tx = t % tiles_x;

    const int col0 = (tx * kMarchWaves + wave) * 64;
    const int j = col0 + lane;
    const int ra = r0 + ty * seg_rows;
    const int rb = min(ra + seg_rows, r1);
    if (col0 >= ncols) return;  // (whole wave out of the mesh)
    const bool col_ok = j < ncols;
    const unsigned jc = (unsigned)min(j, ncols - 1);
    const int rmax = g.nrows - 1;
    const double rks = g.rk[jc], rkn = g.rk[min(jc + 1u, (unsigned)ncols - 1u)], rkm = g.rkm[jc];
    const bool edge_lane = (lane == 0 && jc >= 1u) || (lane == 63 && jc + 1u < (unsigned)ncols);
    // first double of the 16-byte granule a lane asks for (rows are ld doubles long, ld a multiple of 16, so a
    // granule clamped to ld - 2 stays inside its row; clamped granules hold columns no node reads)
    const int gmax = g.ld - 2;
    const unsigned gs = (unsigned)min(max(col0 - 2 + 2 * min(lane, 33), 0), gmax);  // stencil arrays, lanes 0..33
    const unsigned gn = (unsigned)min(col0 + 2 * (lane & 31), gmax);               // node arrays, two per instruction
    const bool s_lane = lane < 34;
    const unsigned je = (lane == 0) ? (jc >= 1u ? jc - 1u : 0u) : jc + 1u;  // (as k_march: the first row's outside value)

    const double* sp[NS];
    const double* cp[NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) sp[k] = f.sptr(k);
#pragma unroll
    for (int k = 0; k < NC; ++k) cp[k] = f.cptr(k);

    auto row_of = [&](int li) { return (size_t)min(max(li, 0), rmax) * g.ld; };
    // everything row li needs that the wave does not hold yet: row li + 1 of the stencil arrays (and the outer
    // granules of row li), row li of the node arrays
    auto request = [&](int slot, int li) {
        const size_t oc = row_of(li), on = row_of(li + 1);
        double* base = &ring[wave][slot][0];
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (s_lane) __builtin_amdgcn_global_load_lds((global_void_ptr)(sp[k] + on + gs), (lds_void_ptr)(base + 68 * k), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < NC; k += 2) {
            const double* src = (lane < 32 || k + 1 >= NC) ? cp[k] : cp[k + 1 < NC ? k + 1 : k];
            const bool on_duty = (lane < 32 || k + 1 < NC) && src != nullptr;
            if (on_duty)
                __builtin_amdgcn_global_load_lds((global_void_ptr)(src + oc + gn), (lds_void_ptr)(base + 68 * NS + 64 * k), 16, 0, 0);
        }
    };

    double w[NS], c[NS], e[NS], edge[NS], edge_e[NS], cv[NC];
    {
        const size_t ow = row_of(ra - 1), oc = row_of(ra);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            w[k] = (sp[k] + ow)[jc];
            c[k] = (sp[k] + oc)[jc];
            edge[k] = edge_lane ? (sp[k] + oc)[je] : 0.0;
        }
    }
    // Row metrics of the row being computed, requested a row ahead like everything else and BEFORE the batch of
    // that row: vector loads complete in order, so the wait for these never waits for the batch behind them.
    double m_rh0 = g.rh[min(ra, rmax)], m_rh1 = g.rh[min(min(ra, rmax) + 1, rmax)], m_rhm = g.rhm[min(ra, rmax)];
    request(0, ra);
    if constexpr (SPLIT && HasNodeVals<F>::value) {  // (the first batch and the first rows: waited for once, here)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
    }
    for (int li = ra; li < rb; ++li) {
        const int slot = (li - ra) & 1;
        {
            const double* base = &ring[wave][slot][0];
            // The batch asked for a row ago has arrived once every vector memory operation of the wave has (they
            // complete in order).  The compiler does not connect the reads below with the loads that wrote LDS: the
            // wait is spelled out.  Where it stands: a functor in two parts (HasNodeVals) is waited for at the end of
            // the row before, between its arithmetic and its stores -- the batch has had that row's arithmetic to
            // arrive and the stores have this row's to complete; otherwise here, behind the stores of the row before.
            // (the builtin, not an asm string: the compiler's own bookkeeping of outstanding loads sees it.  gfx9
            // encoding: vmcnt 0, expcnt and lgkmcnt left alone)
            if constexpr (!(SPLIT && HasNodeVals<F>::value)) __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                e[k] = base[68 * k + 2 + lane];
                const double ev = base[68 * k + (lane == 0 ? 1 : 66)];  // (of the row that is the centre next)
                edge_e[k] = edge_lane ? ev : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NC; ++k) cv[k] = cp[k] ? base[68 * NS + 64 * k + lane] : 0.0;
            asm volatile("" ::: "memory");
        }
        const int l1 = min(li + 1, rmax);
        const double n_rh1 = g.rh[min(l1 + 1, rmax)], n_rhm = g.rhm[l1];
        asm volatile("" ::: "memory");
        if (li + 1 < rb) request(slot ^ 1, li + 1);
        St5 s[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            s[k].c = c[k];
            s[k].e = e[k];
            s[k].w = w[k];
            s[k].n = lane_above_or(c[k], edge[k]);
            s[k].s = lane_below_or(c[k], edge[k]);
        }
        const Geo ge{m_rh0, m_rh1, m_rhm, rks, rkn, rkm};  // = rh[lic], rh[min(lic + 1, rmax)], rhm[lic], lic = min(li, rmax)
        if constexpr (SPLIT && HasNodeVals<F>::value) {
            typename F::NodeOut o{};
            if (col_ok) o = f.node_vals(li, j, ge, s, cv);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
            if (col_ok) f.node_store(li, j, o);
        } else {
            if (col_ok) f.node(li, j, ge, s, cv);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            w[k] = c[k];
            c[k] = e[k];
            edge[k] = edge_e[k];
        }
        m_rh0 = m_rh1;
        m_rh1 = n_rh1;
        m_rhm = n_rhm;
    }
}

}  // namespace tedd
