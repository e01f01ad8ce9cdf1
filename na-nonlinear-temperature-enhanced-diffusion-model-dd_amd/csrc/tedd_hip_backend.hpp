// HIP execution backend for gfx950: memory, streams, kernel launchers over the node functors,
// wave64 shuffle reductions, and the cs-corrector kernel with the reference's global early
// exit.  One thread updates one node; a 256-thread workgroup covers 4 rows x 64 columns so
// that each wavefront reads 64 consecutive doubles (512 B) of a row.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cstddef>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "tedd_strip.hpp"
#include "tedd_march.hpp"
#include "tedd_codegen.hpp"
#include "tedd_ctx.hpp"

namespace tedd {

#define TEDD_HIP(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (void)hipGetLastError(); /* reported here: must not surface again at a later launch check */ \
            throw std::runtime_error(std::string(#call) + " failed: " + hipGetErrorString(e_));    \
        }                                                                                           \
    } while (0)

constexpr int kTileX = 64, kTileY = 4, kMaxReduceBlocks = 2048;

// Scratch of the entry point that allocates nothing (tedd_error_norms_noalloc): the per-workgroup partial
// sums of the error norms, kMaxReduceBlocks x 8 doubles = 128 KiB.  Every GPU the library runs on holds
// its own copy of this array at its own address until the process ends; host code never uses the array's
// name as an address: HipBackend::init asks the runtime for the copy's address on the current device
// (hipGetSymbolAddress) once per device and passes that pointer to the kernels.
constexpr int kScratchSums = 8;
constexpr size_t kScratchDoubles = (size_t)kMaxReduceBlocks * kScratchSums;
static __device__ __align__(256) double scratch[kScratchDoubles];
constexpr int kMaxDevices = 64;

// `flat` (a 1-D grid): the nodes of the rows are dealt to the threads one after the other instead of as
// 64 x 4 tiles.  On a small mesh whose width is just over a multiple of 64 (257 columns: 5 tiles across, the
// fifth with one live lane) the tiles make a quarter more waves than the nodes fill, each as long as a full
// one; with 66 049 nodes that is 1300 waves for 1024 SIMDs instead of 1033, and the launch is as long as the
// SIMDs that hold two.
__device__ __forceinline__ bool node_of_thread(bool flat, int r0, int r1, int ncols, int& li, int& j) {
    if (flat) {
        const int q = blockIdx.x * (kTileX * kTileY) + threadIdx.y * kTileX + threadIdx.x;
        const int row = q / ncols;
        li = r0 + row;
        j = q - row * ncols;
        return li < r1;
    }
    j = blockIdx.x * kTileX + threadIdx.x;
    li = r0 + blockIdx.y * kTileY + threadIdx.y;
    return j < ncols && li < r1;
}
template <class F>
__global__ void __launch_bounds__(256) k_nodes(const F f, int r0, int r1, int ncols, bool flat) {
    int li, j;
    if (node_of_thread(flat, r0, r1, ncols, li, j)) f(li, j);
}

// Node functors that interpret an expression program (tedd_prog.hpp): one wave per workgroup, the
// program's registers in LDS as [register][lane].
template <class F>
__global__ void __launch_bounds__(64) k_nodes_prog(const F f, int r0, int r1, int ncols) {
    extern __shared__ double prog_regs[];
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int li = r0 + blockIdx.y;
    if (j < ncols && li < r1) f(li, j, prog_regs + threadIdx.x, 64);
}

// dense (rows x width) <-> pitched copies on the device, so that a field crosses PCIe as one contiguous
// block instead of one small transfer per row
// Streaming probe of a freshly allocated arena of field buffers (HipBackend::arena_setup): every element of 13
// buffers read and of 7 written, the mix of the predictor launch.  The values are whatever the memory held.
__global__ void __launch_bounds__(256) k_arena_probe(double* base, size_t stride_elems, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 13; ++k) s += __builtin_nontemporal_load(base + k * stride_elems + i);
#pragma unroll
    for (int k = 0; k < 7; ++k) __builtin_nontemporal_store(s, base + (13 + k) * stride_elems + i);
}
__global__ void __launch_bounds__(256) k_copy_rows(double* dst, size_t dpitch, const double* src, size_t spitch, int width,
                                                  int rows) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    for (int i = blockIdx.y * 4 + threadIdx.y; i < rows; i += gridDim.y * 4)
        if (j < width) dst[(size_t)i * dpitch + j] = src[(size_t)i * spitch + j];
}

// max-reduction of non-negative values: grid-stride over 4x64 tiles, one atomic per block and slot
template <class F>
__global__ void __launch_bounds__(256) k_nodes_max(const F f, int r0, int r1, int ncols, int tiles_x, int ntiles,
                                                  unsigned long long* out) {
    double red[F::NR];
#pragma unroll
    for (int k = 0; k < F::NR; ++k) red[k] = 0.0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int j = (t % tiles_x) * kTileX + threadIdx.x;
        const int li = r0 + (t / tiles_x) * kTileY + threadIdx.y;
        if (j < ncols && li < r1) f(li, j, red);
    }
    __shared__ double sm[4][F::NR];
    const int wave = threadIdx.y, lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < F::NR; ++k) {
        const double v = wave_max_nan_inf(red[k]);
        if (lane == 0) sm[wave][k] = v;
    }
    __syncthreads();
    if (wave == 0 && lane < F::NR) {
        double v = sm[0][lane];
        for (int w = 1; w < 4; ++w) v = sm[w][lane] > v ? sm[w][lane] : v;
        atomicMax(out + lane, (unsigned long long)__double_as_longlong(v));
    }
}

// sum-reduction, deterministic: per-block partials, then a fixed-order final pass
template <class F>
__global__ void __launch_bounds__(256) k_nodes_sum(const F f, int r0, int r1, int ncols, int tiles_x, int ntiles,
                                                  double* partials) {
    double red[F::NR];
#pragma unroll
    for (int k = 0; k < F::NR; ++k) red[k] = 0.0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int j = (t % tiles_x) * kTileX + threadIdx.x;
        const int li = r0 + (t / tiles_x) * kTileY + threadIdx.y;
        if (j < ncols && li < r1) f(li, j, red);
    }
    __shared__ double sm[4][F::NR];
    const int wave = threadIdx.y, lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < F::NR; ++k) {
        const double v = wave_sum_fixed(red[k]);
        if (lane == 0) sm[wave][k] = v;
    }
    __syncthreads();
    if (wave == 0 && lane < F::NR)
        partials[(size_t)blockIdx.x * F::NR + lane] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}
// one wave per reduced quantity (blockDim = 64 x nr): lane l adds the partials l, l+64, ... in that
// order, then the lanes are combined by a fixed butterfly -- the same order on every run
// (a grid of several workgroups finishes several sets of partial sums, one set per workgroup: the sums of the
// steps of a run, whose final passes are all left to the end of the run)
__global__ void k_sum_final(const double* partials, int nblocks, int nr, double* out) {
    const int k = threadIdx.y;
    partials += (size_t)blockIdx.x * nblocks * nr;
    out += (size_t)blockIdx.x * nr;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partials[(size_t)b * nr + k];
    s = wave_sum_fixed(s);
    if (threadIdx.x == 0) out[k] = s;
}

// ---- cs corrector (RegHCsTriple): cp trapezoid + per-node scalar Newton for cs with the
// per-iteration global record  max|dx_k|, min|x_k|  that decides the reference's early exit
// (src/prob1base.py:3654-3663: stop after iteration k when max|dx| < rtol |x| at every node).
__global__ void k_cs_init(unsigned long long* gmax, unsigned long long* gmin, int iters) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < iters) {
        gmax[k] = 0ull;
        gmin[k] = 0x7FF0000000000000ull;  // +inf
    }
}
// Start of a predictor-corrector step (one workgroup).  Folds the convergence record of the step before
// into the sticky worst ratio  last_update / (tol * increment)  per system, publishes that record to a
// host-visible ring slot (the host reads it two steps later to choose sweep counts, without ever
// synchronising with the stream), then clears the record of the Newton solves and that of the cs corrector.
__device__ __forceinline__ void step_prologue_body(const StepRecordArgs& a, unsigned long long* gmax,
                                                   unsigned long long* gmin, int k, int nthreads, double* rec /*LDS, 16*/) {
    // (the record words are raised by atomics, which do not pass through this compute unit's L1: read them
    // past it, or a launch that runs several steps finds the zeros it stored itself a step earlier)
    if (k < 16)
        rec[k] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(a.red) + k,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    if (k < 3 && a.fold) fold_record(a, rec, k);
    if (k == 0 && a.ring_slot) {
        double* slot = a.ring_slot;
        for (int v = 0; v < 3; ++v) {
            slot[1 + v] = rec[2 * v];
            slot[4 + v] = rec[2 * v + 1];
            slot[7 + v] = rec[8 + 2 * v];
            slot[10 + v] = rec[9 + 2 * v];
        }
        __threadfence_system();
        *(volatile double*)slot = a.seq;
    }
    if (a.clear) {
        if (k < 16) a.red[k] = 0.0;
        for (int q = k; q < a.cs_iters; q += nthreads) {
            gmax[q] = 0ull;
            gmin[q] = 0x7FF0000000000000ull;
        }
    }
}
__global__ void k_step_prologue(const StepRecordArgs a, unsigned long long* gmax, unsigned long long* gmin) {
    __shared__ double rec[16];
    step_prologue_body(a, gmax, gmin, threadIdx.x, blockDim.x, rec);
}
// k_nodes whose first workgroup also carries out the prologue of the step (small meshes: a launch fewer).
// Nothing else in the launch, and no launch between this one and the solves, touches the records.
template <class F>
__global__ void __launch_bounds__(256) k_nodes_pro(const F f, int r0, int r1, int ncols, const StepRecordArgs a,
                                                  unsigned long long* gmax, unsigned long long* gmin, bool flat) {
    __shared__ double rec[16];
    if (blockIdx.x == 0 && blockIdx.y == 0)
        step_prologue_body(a, gmax, gmin, threadIdx.y * kTileX + threadIdx.x, 256, rec);
    int li, j;
    if (node_of_thread(flat, r0, r1, ncols, li, j)) f(li, j);
}
// Forcing arrays by threads that walk down columns: a workgroup covers 256 columns and `seg_rows` rows.  The
// column's table entries are loaded once into registers; the row's entries have the same address in every
// lane (scalar loads).  What a node costs is then arithmetic and five stores, and a wave's start-up
// (arguments, table loads) is paid once per segment instead of once per node.
template <class F, int QUAD>
__global__ void __launch_bounds__(256) k_forcing_columns(const F f, int r0, int r1, int ncols, int seg_rows) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int ra = r0 + blockIdx.y * seg_rows, rb = min(ra + seg_rows, r1);
    const auto ct = f.template column<QUAD>(min(j, ncols - 1));
    if (j >= ncols) return;
    for (int li = ra; li < rb; ++li) f.template node_in_column<QUAD>(ct, li, j);
}
// number of cs Newton iterations the reference keeps: k+1 for the first k with max|dx_k| < rtol min|x_k|
__device__ __forceinline__ bool cs_stop_after(const unsigned long long* gmax, const unsigned long long* gmin, int k,
                                              double rtol) {
    return __longlong_as_double((long long)gmax[k]) < rtol * __longlong_as_double((long long)gmin[k]);
}
// KREG: iterations whose record a thread keeps in registers (8 serves every count up to 8; the build for 5, the
// default count, needs 12 registers fewer, which is one more wave per SIMD)
template <int CASE, bool FLAT = false, int KREG = 8>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KREG <= 5 ? 5 : 4, 8))) k_cs_newton(const CorrectIn<CASE, REACT_REG> a, int r0, int r1, int ncols,
                                                  int tiles_x, int ntiles, double* cp_out, double* cs_out, int iters,
                                                  bool record, double rtol, int* out_keep, unsigned long long* gmax,
                                                  unsigned long long* gmin) {
    extern __shared__ unsigned long long sm[];  // [iters] max bits, [iters] min bits
    __shared__ int keep_sm;
    const int tid = threadIdx.y * kTileX + threadIdx.x;
    int keep = iters;
    TEDD_STAMP(0);
    if (!record) {
        // second pass: every workgroup reads the decision off the record of the first; the iteration is
        // redone only if the global test stopped it early
        if (tid == 0) keep_sm = iters;
        __syncthreads();
        for (int k = tid; k < iters; k += 256)
            if (cs_stop_after(gmax, gmin, k, rtol)) atomicMin(&keep_sm, k + 1);
        __syncthreads();
        keep = keep_sm;
        if (blockIdx.x == 0 && tid == 0) out_keep[0] = keep;
        if (keep >= iters) return;
    } else {
        for (int k = tid; k < iters; k += 256) {
            sm[k] = 0ull;
            sm[iters + k] = 0x7FF0000000000000ull;
        }
        __syncthreads();
    }
    // The record words thread `tid` may move at the end of the record pass, read now: a workgroup that does not
    // move them then knows so without a round trip at its very end (a value read too early only costs an atomic).
    unsigned long long seen_mx = 0ull, seen_mn = 0xFFFFFFFFFFFFFFFFull;
    if (record && tid < iters) {
        seen_mx = __atomic_load_n(gmax + tid, __ATOMIC_RELAXED);
        seen_mn = __atomic_load_n(gmin + tid, __ATOMIC_RELAXED);
    }
    // per-thread record of the first kCsReg iterations (the usual case: 5), reduced once at the end
    constexpr int kCsReg = KREG;
    const bool in_regs = (iters <= kCsReg);
    double rmx[kCsReg], rmn[kCsReg];
    int first_bad = 0x7fffffff;  // first iteration at which some node of this thread held a NaN
#pragma unroll
    for (int k = 0; k < kCsReg; ++k) {
        rmx[k] = 0.0;
        rmn[k] = INFINITY;
    }
    // a workgroup walks over its tiles; the six values a node reads are requested one tile ahead, so that
    // the iteration of the present tile runs while they travel
    // (FLAT: "tiles" are runs of 256 consecutive nodes of the rows, see k_nodes)
    auto place = [&](int t, int& j, int& li) {
        if (FLAT) {
            const int q = t * (kTileX * kTileY) + tid;
            const int row = q / ncols;
            li = r0 + row;
            j = q - row * ncols;
            return li < r1;
        }
        j = (t % tiles_x) * kTileX + threadIdx.x;
        li = r0 + (t / tiles_x) * kTileY + threadIdx.y;
        return j < ncols && li < r1;
    };
    CsNodeIn nxt{0, 0, 0, 0, 0, 0};
    {
        int j0, li0;
        if (blockIdx.x < ntiles && place(blockIdx.x, j0, li0)) nxt = cs_node_load(a, at(a.g, li0, j0));
    }
    TEDD_STAMP_WAIT();
    TEDD_STAMP(1);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int j, li;
        const bool active = place(t, j, li);
        const CsNodeIn v = nxt;
        {
            int jn, lin;
            const int tn = t + gridDim.x;
            if (tn < ntiles && place(tn, jn, lin)) nxt = cs_node_load(a, at(a.g, lin, jn));
        }
        double x = 0.0, y = 0.0, av = 0.0;
        bool in = false;
        size_t c = 0;
        if (active) {
            c = at(a.g, li, j);
            in = is_interior(a.g, li, j);
            if (record) put_once(cp_out + c, corrector_cp_vals(a.m, a.dt, v, in));
            cs_newton_inputs_vals(a.m, a.dt, v, y, av, x);
        }
        double ex = active ? cs_newton_exp(x, a.m.eta) : 1.0;
        if (in_regs) {
#pragma unroll
            for (int k = 0; k < kCsReg; ++k) {
                if (k < keep && active) {
                    const double dx = cs_newton_step(x, y, av, a.m.eta, ex);
                    x = x + dx;
                    if (k + 1 < keep) ex = cs_newton_exp_next(ex, x, dx, a.m.eta);
                    cs_record_update(dx, x, k, rmx[k], rmn[k], first_bad);
                }
            }
        } else {
            for (int k = 0; k < keep; ++k) {
                double adx = 0.0, ax = INFINITY;
                if (active) {
                    const double dx = cs_newton_step(x, y, av, a.m.eta, ex);
                    x = x + dx;
                    if (k + 1 < keep) ex = cs_newton_exp_next(ex, x, dx, a.m.eta);
                    adx = fabs(dx);
                    ax = fabs(x);
                    if (adx != adx) adx = INFINITY;
                    if (ax != ax) ax = 0.0;
                }
                if (record) {
                    const double wmx = wave_max_nn(adx), wmn = wave_min_nn(ax);  // (NaN replaced above)
                    if (threadIdx.x == 0) {
                        atomicMax(&sm[k], (unsigned long long)__double_as_longlong(wmx));
                        atomicMin(&sm[iters + k], (unsigned long long)__double_as_longlong(wmn));
                    }
                }
            }
        }
        if (active) put_once(cs_out + c, in ? x : 0.0 * x);
    }
    TEDD_STAMP(2);
    // The record of the usual case leaves the workgroup through plain LDS words, one set per wave (lane k keeps
    // the wave's values of iteration k), combined by thread k after the barrier: no LDS atomics.
    __shared__ unsigned long long wrec[kTileY][2 * kCsReg];
    if (record && in_regs) {
        double mine_mx = 0.0, mine_mn = INFINITY;
#pragma unroll
        for (int k = 0; k < kCsReg; ++k) {
            if (k < iters) {
                cs_record_finish(k, rmx[k], rmn[k], first_bad);
                const double wmx = wave_max_nn(rmx[k]), wmn = wave_min_nn(rmn[k]);  // (no NaN among them)
                if ((int)threadIdx.x == k) {
                    mine_mx = wmx;
                    mine_mn = wmn;
                }
            }
        }
        if ((int)threadIdx.x < iters) {
            wrec[threadIdx.y][threadIdx.x] = (unsigned long long)__double_as_longlong(mine_mx);
            wrec[threadIdx.y][kCsReg + threadIdx.x] = (unsigned long long)__double_as_longlong(mine_mn);
        }
    }
    TEDD_STAMP(3);
    if (record) {
        __syncthreads();
        TEDD_STAMP(4);
        if (in_regs) {
            if (tid < iters) {
                unsigned long long mx = wrec[0][tid], mn = wrec[0][kCsReg + tid];
#pragma unroll
                for (int w = 1; w < kTileY; ++w) {
                    mx = wrec[w][tid] > mx ? wrec[w][tid] : mx;
                    mn = wrec[w][kCsReg + tid] < mn ? wrec[w][kCsReg + tid] : mn;
                }
                if (mx > seen_mx) atomicMax(gmax + tid, mx);
                if (mn < seen_mn) atomicMin(gmin + tid, mn);
            }
        } else {
            for (int k = tid; k < iters; k += 256) {
                if (k != tid || sm[k] > seen_mx) atomicMax(gmax + k, sm[k]);
                if (k != tid || sm[iters + k] < seen_mn) atomicMin(gmin + k, sm[iters + k]);
            }
        }
    }
    TEDD_STAMP_WAIT();
    TEDD_STAMP(5);
}
__global__ void k_cs_decide(const unsigned long long* gmax, const unsigned long long* gmin, int iters, double rtol,
                            int* out_keep) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int keep = iters;
    for (int k = 0; k < iters; ++k) {
        if (cs_stop_after(gmax, gmin, k, rtol)) {
            keep = k + 1;
            break;
        }
    }
    out_keep[0] = keep;
}

// closed-form correctors (CsTriple: trapezoid; HCsTriple: sign cases): plain node functor
template <int CASE, int REACT>
struct CorrectClosedFn {
    CorrectIn<CASE, REACT> a;
    double *cp_out, *cs_out;
    int* bad_flag;  // set to 1 where the HCsTriple corrector's denominator is not positive
    __device__ void operator()(int li, int j) const {
        const size_t c = at(a.g, li, j);
        cp_out[c] = corrector_cp(a, li, j);
        if (REACT == REACT_HCS) {
            bool bad = false;
            cs_out[c] = corrector_cs_sign_case(a, li, j, &bad);
            if (bad) *bad_flag = 1;
        } else {
            cs_out[c] = corrector_cs_closed_form(a, li, j);
        }
    }
};

}  // namespace tedd
#include "tedd_persist.hpp"
namespace tedd {

// node functors that provide the marching interface (tedd_march.hpp)
template <class F, class = void>
struct Marches : std::false_type {};
template <class F>
struct Marches<F, std::enable_if_t<F::kMarches>> : std::true_type {};

struct HipBackend {
    static constexpr bool kHasStripSolve = true;
    // marching launcher for the stencil functors: -1 from the slab size each functor names (kMarchMinNodes; below, the plain
    // launcher's many small workgroups fill the GPU better), 0 never, 1 always
    int opt_march = -1;
    int opt_march_rows = 0;  // rows per segment (0: per functor, fewer on slabs with few rows)
    int opt_cs_lean = 1;          // cs corrector of at most 5 iterations: the build with 5 records per thread, 5 waves per SIMD
    int opt_march_lds = -1;       // marching kernels without look-ahead in registers prefetch through LDS (-1: per functor)
    int opt_march_ahead = -1;     // marching kernels request the next row a row ahead (-1: per functor)
    int opt_march_split = -1;     // look-ahead march: wait for the next row's loads between a node's arithmetic and its stores (0: the compiler's wait at the top of the row)
    int opt_strip_panel = 8;      // tile rows per panel of the strip launches' tile order (0: row-major)
    int opt_forcing_columns = 1;  // forcing arrays by column-walking threads (0: one thread per node)
    int opt_march_occ = 0;   // waves per SIMD the marching kernels' register budget is set for (0: per functor; 2, 3, 4)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int device = -1;
    int num_cus = 256;
    double* d_partials = nullptr;
    double* d_scratch = nullptr;              // this device's copy of the static scratch array
    unsigned long long* d_csstats = nullptr;  // [4096] max, [4096] min
    int cs_record_clean = 0;                  // leading entries of both known to hold their initial values
    bool profiling = false;
    // reduction of the cs corrector's record across the slabs of a decomposed mesh (tedd_set_reduce_hook)
    tedd_reduce_fn reduce_hook = nullptr;
    void* reduce_user = nullptr;
    struct Rec {
        int fam;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> event_pool;
    double fam_ms[FAM_COUNT] = {0};
    long long fam_n[FAM_COUNT] = {0};

    // ---- persistent step kernel (tedd_persist.hpp): while `rec_on`, launches the kernel can run are recorded
    // as descriptors instead of being launched; anything else this backend is asked to do first sends what was
    // recorded to the device as one launch (flush_record), so the order of the stream is kept.
    static constexpr bool kHasPersist = true;
    static constexpr size_t kRecCap = 1024;  // descriptors per launch
    bool rec_on = false;
    int rec_react = -1;
    int rec_groups = 1;  // workgroups of the launch
    std::vector<PDesc> rec;
    PDesc* d_descs = nullptr;
    PDesc* h_descs[2] = {nullptr, nullptr};  // pinned staging, used in turn
    hipEvent_t h_descs_ev[2] = {nullptr, nullptr};
    bool h_descs_busy[2] = {false, false};
    int h_descs_next = 0;
    unsigned* d_pwords = nullptr;  // [0] barrier counter, [1] abort word
    unsigned pbar_base = 0;
    int* h_pabort = nullptr;       // host-visible: a barrier of a persistent launch timed out
    int* d_pabort = nullptr;       // ... as the device addresses it
    long long persist_launches = 0, persist_descs = 0;

    static bool persist_react_ok(int react) { return react == REACT_REG; }
    void rec_begin(int react, int groups) {
        if (!persist_react_ok(react)) return;
        groups = std::max(1, std::min(groups, num_cus));
        if (rec_on && (react != rec_react || groups != rec_groups)) flush_record();
        if (!d_descs) {
            d_descs = (PDesc*)alloc(kRecCap * sizeof(PDesc));
            for (int k = 0; k < 2; ++k) {
                TEDD_HIP(hipHostMalloc((void**)&h_descs[k], kRecCap * sizeof(PDesc), hipHostMallocDefault));
                TEDD_HIP(hipEventCreateWithFlags(&h_descs_ev[k], hipEventDisableTiming));
            }
            d_pwords = (unsigned*)alloc(64);
            TEDD_HIP(hipMemsetAsync(d_pwords, 0, 64, stream));
            h_pabort = (int*)alloc_host_visible(64);
            d_pabort = (int*)device_view(h_pabort);
            rec.reserve(kRecCap);
        }
        rec_react = react;
        rec_groups = groups;
        rec_on = true;
    }
    // end of an entry point: what was recorded runs now
    void rec_end() {
        if (!rec_on) return;
        rec_on = false;
        flush_record();
    }
    void rec_discard() noexcept {
        rec_on = false;
        rec.clear();
    }
    PDesc& rec_push(int op, int ftype, int barrier, int r0, int r1, int ncols) {
        if (rec.size() >= kRecCap) flush_record();
        rec.emplace_back();
        PDesc& d = rec.back();
        std::memset(&d, 0, offsetof(PDesc, f));
        d.op = op;
        d.ftype = ftype;
        d.barrier = barrier;
        d.r0 = r0;
        d.r1 = r1;
        d.ncols = ncols;
        return d;
    }
    template <class F>
    static void rec_functor(PDesc& d, const F& f) {
        static_assert(sizeof(F) <= kPFunctorBytes, "functor larger than a descriptor holds");
        static_assert(std::is_trivially_copyable<F>::value, "functors are copied bytewise");
        std::memcpy(d.f, &f, sizeof(F));
    }
    void flush_record() {
        if (rec.empty()) return;
        const int slot = h_descs_next;
        h_descs_next ^= 1;
        if (h_descs_busy[slot]) TEDD_HIP(hipEventSynchronize(h_descs_ev[slot]));
        const size_t n = rec.size();
        std::memcpy(h_descs[slot], rec.data(), n * sizeof(PDesc));
        unsigned barriers = 0;
        for (const PDesc& d : rec) barriers += d.barrier ? 1u : 0u;
        rec.clear();
        TEDD_HIP(hipMemcpyAsync(d_descs, h_descs[slot], n * sizeof(PDesc), hipMemcpyHostToDevice, stream));
        TEDD_HIP(hipEventRecord(h_descs_ev[slot], stream));
        h_descs_busy[slot] = true;
        PersistCtl ctl{d_pwords, pbar_base, d_pabort, d_csstats, d_csstats + 4096};
        hipLaunchKernelGGL(k_persist<REACT_REG>, dim3(rec_groups), dim3(64, kPWaves), 0, stream, d_descs, (int)n, ctl);
        TEDD_HIP(hipGetLastError());
        if (rec_groups > 1) pbar_base += barriers * (unsigned)rec_groups;
        persist_launches += 1;
        persist_descs += (long long)n;
    }
    void check_persist_abort() {
        if (h_pabort && *(volatile int*)h_pabort != 0) {
            *(volatile int*)h_pabort = 0;
            (void)hipMemsetAsync(d_pwords, 0, 64, stream);
            (void)hipStreamSynchronize(stream);
            pbar_base = 0;
            throw std::runtime_error("persistent step kernel: a grid barrier timed out (workgroups not all resident); "
                                     "the state is undefined, set it again (option persistent = 0 uses separate launches)");
        }
    }

    void init(int dev) {
        int cur = 0;
        TEDD_HIP(hipGetDevice(&cur));
        device = (dev >= 0) ? dev : cur;
        TEDD_HIP(hipSetDevice(device));
        TEDD_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, device));
        TEDD_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        own_stream = true;
        d_scratch = scratch_on_device(device);
        d_partials = (double*)alloc((size_t)kMaxReduceBlocks * 8 * sizeof(double));
        d_csstats = (unsigned long long*)alloc(2 * 4096 * sizeof(unsigned long long));
    }
    void shutdown() {
        if (stream) (void)hipStreamSynchronize(stream);
        gen_unload();
        for (Rec& r : recs) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
        free(d_partials);
        free(d_csstats);
        free(d_descs);
        free(d_pwords);
        d_descs = nullptr;
        d_pwords = nullptr;
        for (int k = 0; k < 2; ++k) {
            if (h_descs[k]) (void)hipHostFree(h_descs[k]);
            if (h_descs_ev[k]) (void)hipEventDestroy(h_descs_ev[k]);
            h_descs[k] = nullptr;
            h_descs_ev[k] = nullptr;
        }
        if (h_pabort) free_host_visible(h_pabort);
        h_pabort = nullptr;
        rec_discard();
        free(d_stage);
        d_stage = nullptr;
        stage_elems = 0;
        if (h_pinned) (void)hipHostFree(h_pinned);
        h_pinned = nullptr;
        pinned_elems = 0;
        if (own_stream && stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
    // Work queued on the stream in use so far is ordered before anything the new stream is given.
    void set_stream(void* s) {
        hipStream_t ns = (hipStream_t)s;
        flush_record();
        if (ns == stream && !own_stream) return;
        hipEvent_t e = get_event();
        TEDD_HIP(hipEventRecord(e, stream));
        TEDD_HIP(hipStreamWaitEvent(ns, e, 0));
        event_pool.push_back(e);
        if (own_stream && stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        stream = ns;
        own_stream = false;
    }
    void use_device() { TEDD_HIP(hipSetDevice(device)); }
    // The caller's current device, put back when an entry point returns (a host program that drives
    // several devices must not find another one current after a call into this library).
    struct DeviceKeeper {
        int prev = -1;
        DeviceKeeper() {
            if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        }
        ~DeviceKeeper() {
            int now = -1;
            if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
        }
    };
    // Address of the static scratch array on device `dev` (which must be the current device); looked up
    // once per device and kept for the life of the process.
    static double* scratch_on_device(int dev) {
        static std::mutex mu;
        static double* ptr[kMaxDevices] = {nullptr};
        if (dev < 0 || dev >= kMaxDevices) throw std::runtime_error("device ordinal out of range");
        std::lock_guard<std::mutex> lock(mu);
        if (!ptr[dev]) {
            double* p = nullptr;
            TEDD_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
            if (!p) throw std::runtime_error("hipGetSymbolAddress returned a null scratch address");
            ptr[dev] = p;
        }
        return ptr[dev];
    }
    // Where the field buffers of a large slab lie in device memory changes the time of the launches that stream
    // many of them at once, by up to 8 % for the predictor (13 arrays read, 8 written) of a 16384^2 mesh on an
    // MI355X: 9.0-9.2 ms per launch with the buffers in the first memory a process is given by separate
    // hipMalloc calls (one after the other in physical memory, every buffer on a 2 GiB boundary), 8.2-8.9 ms
    // anywhere else, for the whole life of a context.  Counters (rocprofv3, identical launches in a slow and a
    // fast context of one process): requests, L2 hits and misses, address translations (UTCL1 hits, misses,
    // UTCL2 busy) all agree to 0.1 %; what moves is the time a read spends between the L2 and the memory
    // (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ: 1900 against 1710 cycles, +11 %).  A plain streaming kernel of the
    // same 13 + 8 arrays carved from ONE allocation runs at 5.2 to 6.5 TB/s depending on nothing but the
    // distance between consecutive arrays (reproducible to 0.2 % inside a process): the slowest distances are
    // the array's own size, i.e. buffers packed back to back; from about 1.8 times the size on the rate is
    // within 3 % of the best.  So the buffers of a large slab are carved from one allocation, two buffer sizes
    // apart, when that much memory is to spare (the gaps stay unused; at most half of the free device memory is
    // taken), packed back to back otherwise.  TEDD_FIELD_GAP_MB in the environment sets the gap between the end
    // of one buffer and the start of the next instead (0: packed; -1: separate allocations, no arena).  An
    // arena that cannot be allocated is not an error: the buffers are then allocated one by one, and
    // placement() says so.  With the arena in place two states are left, decided by which physical memory the
    // allocation was given: see the probe in arena_setup.
    static constexpr bool kHasArena = true;
    static constexpr double kArenaFastGBps = 6300.0;
    char* arena = nullptr;
    size_t arena_stride = 0;
    int arena_slots = 0, arena_used = 0;
    std::string placement_note = "separate allocations";
    const char* placement() const { return placement_note.c_str(); }
    void arena_setup(size_t field_bytes, int slots) {
        const size_t mb = (size_t)1 << 20;
        const size_t rounded = (field_bytes + 2 * mb - 1) / (2 * mb) * (2 * mb);
        long long gap_mb = -2;  // -2: automatic
        if (const char* e = std::getenv("TEDD_FIELD_GAP_MB")) gap_mb = std::atoll(e);
        if (gap_mb == -1) return;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            (void)hipGetLastError();
            placement_note = "separate allocations (hipMemGetInfo failed)";
            return;
        }
        size_t stride = rounded + (gap_mb >= 0 ? (size_t)gap_mb * mb : rounded);
        if (gap_mb == -2 && stride * slots > fr / 2) stride = rounded;  // no room for the gaps: packed
        if (stride * slots > fr - fr / 16) {
            placement_note = "separate allocations (one arena would not fit into the free device memory)";
            return;
        }
        void* p = nullptr;
        hipError_t err = hipMalloc(&p, stride * slots);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            placement_note = std::string("separate allocations (arena refused: ") + hipGetErrorString(err) + ")";
            return;
        }
        // Which physical memory the arena was given decides between two states of everything that streams from it
        // (13 reads + 7 writes: 6.2 or 5.3 TB/s; the predictor of a 16384^2 mesh 7.6 or 8.2 ms), and an arena that
        // is freed and allocated again -- at the same virtual address -- is given the two in turn.  So: stream
        // through the arena once, give it back, take it again, stream again, and if the first was the faster one
        // take turns once more.  Never more than one arena is held; every failure leaves the buffers to separate
        // allocations and says so.  TEDD_FIELD_PROBE=0 keeps the first allocation.  Taking 138 GB costs 2 s, so an
        // arena that streams at kArenaFastGBps or more is kept without a second look: on six MI355X the two states
        // of a 16384^2 arena streamed at 6.50-6.62 and at 5.92-5.99 TB/s.
        char probe_note[96] = "";
        // (arenas of a quarter of the device memory or more: smaller ones -- 34 GB at 8192^2, 18 GB at 6000^2 --
        // stream at one rate wherever they are taken, 2 % apart at most, and the turn would cost them seconds for nothing)
        bool probing = slots >= 20 && stride * slots >= tot / 4;
        if (const char* e = std::getenv("TEDD_FIELD_PROBE")) probing = probing && std::atoi(e) != 0;
        if (probing) {
            double rate[3] = {0.0, 0.0, 0.0};
            int kept = 0;
            for (int turn = 0; turn < 3; ++turn) {
                if (turn > 0) {
                    (void)hipFree(p);
                    p = nullptr;
                    err = hipMalloc(&p, stride * slots);
                    if (err != hipSuccess) {
                        (void)hipGetLastError();
                        placement_note = std::string("separate allocations (arena refused when taken again: ") + hipGetErrorString(err) + ")";
                        return;
                    }
                }
                rate[turn] = arena_probe_rate((double*)p, stride / sizeof(double), field_bytes / sizeof(double));
                kept = turn;
                if (!(rate[turn] > 0.0)) break;                         // (the probe could not be timed: keep what there is)
                if (rate[turn] >= kArenaFastGBps) break;                // the faster of the two states: keep it
                if (turn == 1 && rate[1] >= 0.98 * rate[0]) break;      // the second is no slower: keep it
                                                                        // (else the first was faster: one more turn)
            }
            std::snprintf(probe_note, sizeof probe_note, "; streamed at %.2f / %.2f / %.2f TB/s when taken, kept the %s", rate[0] / 1e3,
                          rate[1] / 1e3, rate[2] / 1e3, kept == 0 ? "first" : kept == 1 ? "second" : "third");
        }
        arena = (char*)p;
        arena_stride = stride;
        arena_slots = slots;
        arena_used = 0;
        char buf[256];
        std::snprintf(buf, sizeof buf, "one arena of %d buffers, %zu MiB apart (buffer %zu MiB)%s", slots, stride / mb,
                      rounded / mb, probe_note);
        placement_note = buf;
    }
    // GB/s of k_arena_probe over the first 20 buffers of an arena (second of two launches); 0 if it cannot be timed
    double arena_probe_rate(double* base, size_t stride_elems, size_t n) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            (void)hipGetLastError();
            if (a) (void)hipEventDestroy(a);
            return 0.0;
        }
        const unsigned blocks = (unsigned)((n + 255) / 256);
        float ms = 0.0f;
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; ++rep) {
            ok = ok && hipEventRecord(a, stream) == hipSuccess;
            hipLaunchKernelGGL(k_arena_probe, dim3(blocks), dim3(256), 0, stream, base, stride_elems, n);
            ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(b, stream) == hipSuccess &&
                 hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess;
        }
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        if (!ok || !(ms > 0.0f)) {
            (void)hipGetLastError();
            return 0.0;
        }
        return 20.0 * (double)n * 8.0 / ((double)ms * 1e-3) / 1e9;
    }
    // a field buffer: the next slot of the arena while it has one, else an allocation of its own
    void* field_alloc(size_t bytes) {
        if (arena && arena_used < arena_slots && bytes <= arena_stride) return arena + (size_t)(arena_used++) * arena_stride;
        return alloc(bytes);
    }
    void field_free(void* p) {
        if (arena && (char*)p >= arena && (char*)p < arena + arena_stride * arena_slots) return;
        free(p);
    }
    void arena_release() {
        if (arena) (void)hipFree(arena);
        arena = nullptr;
        arena_slots = arena_used = 0;
    }
    void* alloc(size_t bytes) {
        void* p = nullptr;
        TEDD_HIP(hipMalloc(&p, bytes));
        return p;
    }
    void free(void* p) {
        if (p) (void)hipFree(p);
    }
    void memset0(void* p, size_t bytes) {
        flush_record();
        TEDD_HIP(hipMemsetAsync(p, 0, bytes, stream));
    }
    // prologue of a step: fold / publish the record of the step before, clear the records (k_step_prologue)
    void step_prologue(StepRecordArgs a) {
        if (a.cs_iters > 4096) a.cs_iters = 0;  // (refused later, by correct)
        if (rec_on) {
            rec_functor(rec_push(PO_PROLOGUE, 0, 1, 0, 0, 0), a);
        } else {
            hipLaunchKernelGGL(k_step_prologue, dim3(1), dim3(256), 0, stream, a, d_csstats, d_csstats + 4096);
            TEDD_HIP(hipGetLastError());
        }
        if (a.clear) cs_record_clean = a.cs_iters;
    }
    // the predictor launch with the step's prologue folded into its first workgroup; where the launch is a
    // marching one (large slabs) the prologue keeps a launch of its own
    template <class F>
    void nodes_with_prologue(int fam, const F& f, int r0, int r1, int ncols, StepRecordArgs a) {
        check_range(r0, r1, ncols);
        bool marching = false;
        if constexpr (Marches<F>::value) {
            const long long n = (long long)(r1 - r0) * ncols;
            marching = opt_march > 0 || (opt_march < 0 && n >= F::kMarchMinNodes);
        }
        if (marching || r1 == r0) {
            step_prologue(a);
            nodes(fam, f, r0, r1, ncols);
            return;
        }
        if (a.cs_iters > 4096) a.cs_iters = 0;  // (refused later, by correct)
        if (rec_on) {
            constexpr int kind = IndexOf<F, PNodeList<REACT_REG>>::value;
            if constexpr (kind >= 0) {
                // (the prologue and the predictor touch different data: no barrier between them, as in the one launch)
                rec_functor(rec_push(PO_PROLOGUE, 0, 1, 0, 0, 0), a);
                rec_functor(rec_push(PO_NODES, kind, 0, r0, r1, ncols), f);
                if (a.clear) cs_record_clean = a.cs_iters;
                return;
            }
            flush_record();
        }
        Scope sc(this, fam);
        const bool flat = flat_nodes(r0, r1, ncols);
        dim3 block(kTileX, kTileY), grid = node_grid(flat, r0, r1, ncols);
        hipLaunchKernelGGL(k_nodes_pro<F>, grid, block, 0, stream, f, r0, r1, ncols, a, d_csstats, d_csstats + 4096, flat);
        TEDD_HIP(hipGetLastError());
        if (a.clear) cs_record_clean = a.cs_iters;
    }
    // host-visible memory the device writes and the host polls (the ring of step records)
    void* alloc_host_visible(size_t bytes) {
        void* p = nullptr;
        TEDD_HIP(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(p, 0, bytes);
        return p;
    }
    void free_host_visible(void* p) {
        if (p) (void)hipHostFree(p);
    }
    void* device_view(void* host_visible) {
        void* d = nullptr;
        TEDD_HIP(hipHostGetDevicePointer(&d, host_visible, 0));
        return d;
    }
    void h2d(void* d, const void* s, size_t bytes) {
        flush_record();
        TEDD_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyHostToDevice, stream));
        TEDD_HIP(hipStreamSynchronize(stream));  // the source may be a temporary
    }
    void d2h(void* d, const void* s, size_t bytes) {
        flush_record();
        TEDD_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, stream));
    }
    // Host arrays are dense, device arrays pitched.  A 2-D copy between the two is executed row by row
    // (1.7 ms for a 257 x 257 field); instead the rows are (un)packed on the device through a staging
    // array and cross PCIe as one block.  All of it is ordered on the context's stream.
    double* d_stage = nullptr;
    size_t stage_elems = 0;
    double* stage(size_t elems) {
        if (elems > stage_elems) {
            if (d_stage) {
                TEDD_HIP(hipStreamSynchronize(stream));
                free(d_stage);
            }
            d_stage = (double*)alloc(elems * sizeof(double));
            stage_elems = elems;
        }
        return d_stage;
    }
    void copy_rows(double* d, size_t dpitch, const double* s, size_t spitch, int width, int rows) {
        flush_record();
        dim3 block(64, 4), grid((width + 63) / 64, std::min((rows + 3) / 4, 65535));
        hipLaunchKernelGGL(k_copy_rows, grid, block, 0, stream, d, dpitch, s, spitch, width, rows);
        TEDD_HIP(hipGetLastError());
    }
    void h2d_2d(double* d, size_t dpitch, const double* s, size_t spitch, int width, int rows) {
        if (rows <= 0 || width <= 0) return;
        flush_record();
        const size_t n = (size_t)rows * width;
        if (spitch == (size_t)width && dpitch == (size_t)width) {
            TEDD_HIP(hipMemcpyAsync(d, s, n * sizeof(double), hipMemcpyHostToDevice, stream));
        } else if (spitch == (size_t)width) {
            double* st = stage(n);
            TEDD_HIP(hipMemcpyAsync(st, s, n * sizeof(double), hipMemcpyHostToDevice, stream));
            copy_rows(d, dpitch, st, width, width, rows);
        } else {
            TEDD_HIP(hipMemcpy2DAsync(d, dpitch * sizeof(double), s, spitch * sizeof(double),
                                      (size_t)width * sizeof(double), rows, hipMemcpyHostToDevice, stream));
        }
        TEDD_HIP(hipStreamSynchronize(stream));  // the source may be a temporary
    }
    void d2h_2d(double* d, size_t dpitch, const double* s, size_t spitch, int width, int rows) {
        if (rows <= 0 || width <= 0) return;
        flush_record();
        const size_t n = (size_t)rows * width;
        if (spitch == (size_t)width && dpitch == (size_t)width) {
            TEDD_HIP(hipMemcpyAsync(d, s, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        } else if (dpitch == (size_t)width) {
            double* st = stage(n);
            copy_rows(st, width, s, spitch, width, rows);
            TEDD_HIP(hipMemcpyAsync(d, st, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        } else {
            TEDD_HIP(hipMemcpy2DAsync(d, dpitch * sizeof(double), s, spitch * sizeof(double),
                                      (size_t)width * sizeof(double), rows, hipMemcpyDeviceToHost, stream));
        }
    }
    void sync() {
        flush_record();
        TEDD_HIP(hipStreamSynchronize(stream));
        check_persist_abort();
    }
    void sync_quiet() noexcept {
        if (stream) (void)hipStreamSynchronize(stream);
    }
    // Several pitched device fields to dense host arrays, synchronously.  Small fields (a state of a
    // convergence-study mesh) go through one pinned buffer: one transfer and one wait instead of one per field.
    double* h_pinned = nullptr;
    size_t pinned_elems = 0;
    void d2h_fields(double* const* dst, const double* const* src, int count, size_t spitch, int width, int rows) {
        const size_t n = (size_t)rows * width;
        int wanted = 0;
        for (int v = 0; v < count; ++v) wanted += dst[v] != nullptr;
        if (wanted == 0) return;
        flush_record();
        if (n * wanted * sizeof(double) > ((size_t)32 << 20)) {
            for (int v = 0; v < count; ++v)
                if (dst[v]) d2h_2d(dst[v], width, src[v], spitch, width, rows);
            sync();
            return;
        }
        if (n * wanted > pinned_elems) {
            if (h_pinned) (void)hipHostFree(h_pinned);
            h_pinned = nullptr;
            TEDD_HIP(hipHostMalloc((void**)&h_pinned, n * wanted * sizeof(double), hipHostMallocDefault));
            pinned_elems = n * wanted;
        }
        double* st = stage(n * wanted);
        int k = 0;
        for (int v = 0; v < count; ++v)
            if (dst[v]) copy_rows(st + n * k++, width, src[v], spitch, width, rows);
        TEDD_HIP(hipMemcpyAsync(h_pinned, st, n * wanted * sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
        k = 0;
        for (int v = 0; v < count; ++v)
            if (dst[v]) std::memcpy(dst[v], h_pinned + n * k++, n * sizeof(double));
    }

    // ---- timing record
    hipEvent_t get_event() {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        TEDD_HIP(hipEventCreate(&e));
        return e;
    }
    struct Scope {
        HipBackend* b;
        int fam;
        hipEvent_t a{}, e{};
        Scope(HipBackend* b_, int fam_) : b(b_), fam(fam_) {
            if (b->profiling) {
                a = b->get_event();
                e = b->get_event();
                (void)hipEventRecord(a, b->stream);
            }
        }
        ~Scope() {
            if (b->profiling) {
                (void)hipEventRecord(e, b->stream);
                b->recs.push_back(Rec{fam, a, e});
            }
        }
    };
    void profile_enable(bool on) {
        profile_collect();
        for (int f = 0; f < FAM_COUNT; ++f) {
            fam_ms[f] = 0;
            fam_n[f] = 0;
        }
        profiling = on;
    }
    void profile_collect() {
        flush_record();
        if (recs.empty()) return;
        TEDD_HIP(hipStreamSynchronize(stream));
        for (Rec& r : recs) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
                fam_ms[r.fam] += ms;
                fam_n[r.fam] += 1;
            }
            event_pool.push_back(r.a);
            event_pool.push_back(r.b);
        }
        recs.clear();
    }

    // ---- launchers
    // small slabs whose tiles would be more than a tenth empty lanes: nodes dealt to the threads in a row (k_nodes)
    static bool flat_nodes(int r0, int r1, int ncols) {
        const long long n = (long long)(r1 - r0) * ncols;
        const long long tiled = (long long)((ncols + kTileX - 1) / kTileX) * kTileX * (((r1 - r0) + kTileY - 1) / kTileY) * kTileY;
        return n <= (1LL << 22) && 10 * tiled > 11 * n;
    }
    static dim3 node_grid(bool flat, int r0, int r1, int ncols) {
        if (flat) return dim3((unsigned)(((long long)(r1 - r0) * ncols + kTileX * kTileY - 1) / (kTileX * kTileY)));
        return dim3((ncols + kTileX - 1) / kTileX, (r1 - r0 + kTileY - 1) / kTileY);
    }
    static void check_range(int r0, int r1, int ncols) {
        if (r1 < r0 || ncols <= 0) throw std::runtime_error("bad launch range");
    }
    template <class F>
    void nodes(int fam, const F& f, int r0, int r1, int ncols) {
        check_range(r0, r1, ncols);
        if (r1 == r0) return;
        if (rec_on) {
            constexpr int kind = IndexOf<F, PNodeList<REACT_REG>>::value;
            if constexpr (kind >= 0) {
                rec_functor(rec_push(PO_NODES, kind, 1, r0, r1, ncols), f);
                return;
            }
            flush_record();
        }
        Scope sc(this, fam);
        if constexpr (Marches<F>::value) {
            const long long n = (long long)(r1 - r0) * ncols;
            if (opt_march > 0 || (opt_march < 0 && n >= F::kMarchMinNodes)) {
                const int tiles_x = (ncols + 64 * kMarchWaves - 1) / (64 * kMarchWaves);
                int seg = opt_march_rows > 0 ? opt_march_rows : F::kMarchRows;
                // enough workgroups for every compute unit to see many: shorter segments on slabs with few rows
                while (opt_march_rows <= 0 && seg > 8 && (long long)tiles_x * ((r1 - r0 + seg - 1) / seg) < 16LL * num_cus)
                    seg /= 2;
                const long long nt = (long long)tiles_x * ((r1 - r0 + seg - 1) / seg);
                if (nt > 0x7fffffffLL) throw std::runtime_error("mesh too large for the tile index");
                if (f.g.nrows < 1 || r0 < 0 || r1 > f.g.nrows || ncols > f.g.ld) throw std::runtime_error("bad marching range");
                // register budget: the functor's own choice unless an option overrides it.  Three builds of
                // the kernel: 2 waves per SIMD with the next row requested a row ahead, 3 and 4 without.
                const int occ = opt_march_occ > 0 ? opt_march_occ : F::kMarchOcc;
                const bool ahead = opt_march_ahead >= 0 ? opt_march_ahead != 0 : (occ <= 2 || F::kMarchAhead);
                if constexpr (F::kMarchLds) {
                    if (opt_march_lds != 0 && !ahead) {
                        // (two builds per functor: its own number of waves per SIMD and one fewer)
                        constexpr int K = F::kMarchOcc;
                        if (opt_march_split == 0)
                            hipLaunchKernelGGL((k_march_lds<F, K, false>), dim3((unsigned)nt), dim3(64, kMarchWaves), 0, stream, f, r0, r1,
                                               ncols, seg, tiles_x);
                        else if (occ < K)
                            hipLaunchKernelGGL((k_march_lds<F, K - 1>), dim3((unsigned)nt), dim3(64, kMarchWaves), 0, stream, f, r0,
                                               r1, ncols, seg, tiles_x);
                        else
                            hipLaunchKernelGGL((k_march_lds<F, K>), dim3((unsigned)nt), dim3(64, kMarchWaves), 0, stream, f, r0, r1,
                                               ncols, seg, tiles_x);
                        TEDD_HIP(hipGetLastError());
                        return;
                    }
                }
                auto go = [&](auto occ_c, auto ahead_c) {
                    if (opt_march_split == 0 && decltype(ahead_c)::value && HasNodeVals<F>::value)
                        hipLaunchKernelGGL((k_march<F, decltype(occ_c)::value, decltype(ahead_c)::value, false>), dim3((unsigned)nt),
                                           dim3(64, kMarchWaves), 0, stream, f, r0, r1, ncols, seg, tiles_x);
                    else
                        hipLaunchKernelGGL((k_march<F, decltype(occ_c)::value, decltype(ahead_c)::value>), dim3((unsigned)nt),
                                           dim3(64, kMarchWaves), 0, stream, f, r0, r1, ncols, seg, tiles_x);
                };
                auto by_ahead = [&](auto occ_c) {
                    if (ahead)
                        go(occ_c, std::true_type{});
                    else
                        go(occ_c, std::false_type{});
                };
                if (occ <= 2)
                    by_ahead(std::integral_constant<int, 2>{});
                else if (occ == 3)
                    by_ahead(std::integral_constant<int, 3>{});
                else
                    by_ahead(std::integral_constant<int, 4>{});
                TEDD_HIP(hipGetLastError());
                return;
            }
        }
        const bool flat = flat_nodes(r0, r1, ncols);
        dim3 block(kTileX, kTileY), grid = node_grid(flat, r0, r1, ncols);
        hipLaunchKernelGGL(k_nodes<F>, grid, block, 0, stream, f, r0, r1, ncols, flat);
        TEDD_HIP(hipGetLastError());
    }
    // the five forcing arrays of a closed-form case (F: ForcingFn)
    template <class F>
    void forcing(int fam, const F& f, int r0, int r1, int ncols) {
        check_range(r0, r1, ncols);
        if (r1 == r0) return;
        if (rec_on) {  // (recorded or not, as any node functor)
            nodes(fam, f, r0, r1, ncols);
            return;
        }
        const int tiles_x = (ncols + 255) / 256;
        int seg = 32;
        while (seg > 1 && (long long)tiles_x * ((r1 - r0 + seg - 1) / seg) < 16LL * num_cus) seg /= 2;
        const int tiles_y = (r1 - r0 + seg - 1) / seg;
        if (opt_forcing_columns == 0 || seg < 4 || tiles_y > 65535) {  // small slab: one thread per node fills the chip better
            nodes(fam, f, r0, r1, ncols);
            return;
        }
        Scope sc(this, fam);
        if (f.quad_mode() == 1)
            hipLaunchKernelGGL((k_forcing_columns<F, 1>), dim3(tiles_x, tiles_y), dim3(256), 0, stream, f, r0, r1, ncols, seg);
        else
            hipLaunchKernelGGL((k_forcing_columns<F, 2>), dim3(tiles_x, tiles_y), dim3(256), 0, stream, f, r0, r1, ncols, seg);
        TEDD_HIP(hipGetLastError());
    }
    // ---- expression programs compiled at run time (tedd_codegen.hpp)
    static constexpr bool kHasCodegen = true;
    hipModule_t gen_mod = nullptr;
    hipFunction_t gen_forcing_fn = nullptr, gen_exact_fn = nullptr;
    void gen_unload() {
        if (gen_mod) (void)hipModuleUnload(gen_mod);
        gen_mod = nullptr;
        gen_forcing_fn = gen_exact_fn = nullptr;
    }
    bool gen_ready() const { return gen_mod != nullptr; }
    void gen_load(const HostProg& node, const HostProg& cell, int react) {
        gen_unload();
        hipDeviceProp_t pr;
        TEDD_HIP(hipGetDeviceProperties(&pr, device));
        const std::vector<char>& code = gen_compile(gen_source(node, cell, react), pr.gcnArchName);
        TEDD_HIP(hipModuleLoadData(&gen_mod, code.data()));
        hipError_t e = hipModuleGetFunction(&gen_forcing_fn, gen_mod, "tedd_gen_forcing");
        if (e == hipSuccess) e = hipModuleGetFunction(&gen_exact_fn, gen_mod, "tedd_gen_exact");
        if (e != hipSuccess) {
            gen_unload();
            TEDD_HIP(e);
        }
    }
    void gen_launch(int fam, hipFunction_t fn, void** args, int r0, int r1, int ncols) {
        check_range(r0, r1, ncols);
        if (r1 == r0) return;
        if (!fn) throw std::runtime_error("generated MMS kernels are not loaded");
        if ((r1 - r0 + 3) / 4 > 65535) throw std::runtime_error("slab too tall for the program launch");
        flush_record();
        Scope sc(this, fam);
        TEDD_HIP(hipModuleLaunchKernel(fn, (unsigned)((ncols + 63) / 64), (unsigned)((r1 - r0 + 3) / 4), 1, 64, 4, 1, 0, stream,
                                       args, nullptr));
    }
    void gen_forcing(int fam, GridDev g, ModelC m, ProgAxes ax, double t, Out5 out, int r0, int r1, int ncols) {
        void* args[] = {&g, &m, &ax, &t, &out, &r0, &r1};
        gen_launch(fam, gen_forcing_fn, args, r0, r1, ncols);
    }
    void gen_exact(int fam, GridDev g, ProgAxes ax, double t, Out5 out, int what, int r0, int r1, int ncols) {
        void* args[] = {&g, &ax, &t, &out, &what, &r0, &r1};
        gen_launch(fam, gen_exact_fn, args, r0, r1, ncols);
    }
    template <class F>
    void nodes_prog(int fam, const F& f, int r0, int r1, int ncols, int n_regs) {
        check_range(r0, r1, ncols);
        if (r1 == r0) return;
        if (n_regs < 3 || n_regs > kProgMaxRegs) throw std::runtime_error("bad program register count");
        if (r1 - r0 > 65535) throw std::runtime_error("slab too tall for the program launch");
        flush_record();
        Scope sc(this, fam);
        dim3 block(64, 1), grid((ncols + 63) / 64, r1 - r0);
        hipLaunchKernelGGL(k_nodes_prog<F>, grid, block, (size_t)n_regs * 64 * sizeof(double), stream, f, r0, r1, ncols);
        TEDD_HIP(hipGetLastError());
    }
    static void tiling(int r0, int r1, int ncols, int& tiles_x, int& ntiles, int& nblocks) {
        tiles_x = (ncols + kTileX - 1) / kTileX;
        const long long nt = (long long)tiles_x * ((r1 - r0 + kTileY - 1) / kTileY);
        if (nt > 0x7fffffffLL) throw std::runtime_error("mesh too large for the tile index");
        ntiles = (int)nt;
        nblocks = ntiles < kMaxReduceBlocks ? ntiles : kMaxReduceBlocks;
    }
    template <class F>
    void nodes_max(int fam, const F& f, int r0, int r1, int ncols, double* dev_out) {
        check_range(r0, r1, ncols);
        if (r1 == r0) return;
        if (rec_on) {
            constexpr int kind = IndexOf<F, PMaxList<REACT_REG>>::value;
            if constexpr (kind >= 0) {
                PDesc& d = rec_push(PO_MAX, kind, 1, r0, r1, ncols);
                d.out = dev_out;
                rec_functor(d, f);
                return;
            }
            flush_record();
        }
        Scope sc(this, fam);
        int tx, nt, nb;
        tiling(r0, r1, ncols, tx, nt, nb);
        hipLaunchKernelGGL(k_nodes_max<F>, dim3(nb), dim3(kTileX, kTileY), 0, stream, f, r0, r1, ncols, tx, nt,
                           (unsigned long long*)dev_out);
        TEDD_HIP(hipGetLastError());
    }
    // workgroups (= partial sums per quantity) of a sum launch over these rows
    static int sum_blocks(int r0, int r1, int ncols) {
        if (r1 <= r0) return 0;
        int tx, nt, nb;
        tiling(r0, r1, ncols, tx, nt, nb);
        return nb;
    }
    // the final passes of `count` sum launches whose partial sums lie one set after the other
    void sums_final_many(const double* partials, int nb, int nr, int count, double* dev_out) {
        if (count <= 0 || nb <= 0) return;
        flush_record();
        hipLaunchKernelGGL(k_sum_final, dim3(count), dim3(64, nr), 0, stream, partials, nb, nr, dev_out);
        TEDD_HIP(hipGetLastError());
    }
    // partials_out (optional): leave the partial sums there and skip the final pass (sums_final_many does it later)
    template <class F>
    void nodes_sum(int fam, const F& f, int r0, int r1, int ncols, double* dev_out, double* partials_out = nullptr) {
        check_range(r0, r1, ncols);
        if (r1 == r0) {
            if (partials_out) throw std::runtime_error("deferred sums over an empty range");
            memset0(dev_out, F::NR * sizeof(double));
            return;
        }
        int tx, nt, nb;
        tiling(r0, r1, ncols, tx, nt, nb);
        if (partials_out) {
            flush_record();
            Scope sc(this, fam);
            hipLaunchKernelGGL(k_nodes_sum<F>, dim3(nb), dim3(kTileX, kTileY), 0, stream, f, r0, r1, ncols, tx, nt, partials_out);
            TEDD_HIP(hipGetLastError());
            return;
        }
        if (rec_on) {
            constexpr int kind = IndexOf<F, PSumList>::value;
            if constexpr (kind >= 0) {
                PDesc& d = rec_push(PO_SUM, kind, 1, r0, r1, ncols);
                d.tiles_x = tx;
                d.ntiles = nt;
                d.nb = nb;
                d.nr = F::NR;
                d.out = d_partials;
                rec_functor(d, f);
                PDesc& e = rec_push(PO_SUM_FINAL, 0, 1, 0, 0, 0);
                e.nb = nb;
                e.nr = F::NR;
                e.out = dev_out;
                e.out2 = d_partials;
                return;
            }
            flush_record();
        }
        Scope sc(this, fam);
        hipLaunchKernelGGL(k_nodes_sum<F>, dim3(nb), dim3(kTileX, kTileY), 0, stream, f, r0, r1, ncols, tx, nt, d_partials);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64, F::NR), 0, stream, d_partials, nb, (int)F::NR, dev_out);
        TEDD_HIP(hipGetLastError());
    }
    // nodes_sum without any allocation: the partial sums go to this device's static scratch array, no
    // timing events are taken, and dev_out is the caller's device memory (F::NR doubles).  The partials
    // of min(tiles, kMaxReduceBlocks) workgroups are overwritten before the final pass reads exactly that
    // many, so nothing of an earlier (larger) call is ever read.  Calls that overlap on one GPU (two
    // streams, two contexts) race on the array.
    template <class F>
    void nodes_sum_noalloc(const F& f, int r0, int r1, int ncols, double* dev_out) {
        static_assert(F::NR <= kScratchSums, "scratch holds at most kScratchSums sums per workgroup");
        check_range(r0, r1, ncols);
        if (!d_scratch) throw std::runtime_error("static scratch address not initialised");
        flush_record();
        if (r1 == r0) {
            TEDD_HIP(hipMemsetAsync(dev_out, 0, F::NR * sizeof(double), stream));
            return;
        }
        int tx, nt, nb;
        tiling(r0, r1, ncols, tx, nt, nb);
        if ((size_t)nb * F::NR > kScratchDoubles) throw std::runtime_error("reduction exceeds the static scratch array");
        hipLaunchKernelGGL(k_nodes_sum<F>, dim3(nb), dim3(kTileX, kTileY), 0, stream, f, r0, r1, ncols, tx, nt, d_scratch);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64, F::NR), 0, stream, d_scratch, nb, (int)F::NR, dev_out);
        TEDD_HIP(hipGetLastError());
    }
    // one launch of the register-resident Newton solve chain (tedd_strip.hpp)
    // shape: one of the SHAPE_* of tedd_strip.hpp, or -1 = by mesh size and system: small tiles when
    // the large ones would leave most compute units without work (fewer than half a tile per unit); otherwise
    // 64-row tiles of 8 waves x 8 rows for T and 48-row tiles for cl and cd (tedd_strip.hpp, StripShape).  A shape
    // whose tile the depth H does not fit falls back to the large one.
    // tile shape of a strip launch of depth H over rows [r0, r1): the caller's choice, or by mesh size and system
    int strip_shape_for(int var, int H, int r0, int r1, int ncols, int shape) const {
        const int TIC = kSTile - 2 * strip_col_halo(H);
        const long long tiles_x = (ncols + TIC - 1) / TIC;
        auto n_tiles = [&](int sh) {
            const int TIR = strip_tile_rows(sh) - 2 * H;
            return tiles_x * ((r1 - r0 + TIR - 1) / TIR);
        };
        auto fits = [&](int sh) { return strip_tile_rows(sh) - 2 * H >= (sh == SHAPE_TINY ? 6 : 8); };
        if (shape == SHAPE_OCT && var != VAR_T) shape = -1;  // (built for the T system only)
        if (shape < 0) {
            // (half a tile per compute unit is where the small tiles stop paying: alternated inside one context,
            // per step with small tiles / with the shapes of large meshes: 512^2 0.11 / 0.11 ms, 640^2 0.16 / 0.14,
            // 768^2 0.20 / 0.17, 1024^2 0.31 / 0.24 -- the T chain alone 0.11 / 0.04 there; 256^2 0.05 / 0.07)
            if (fits(SHAPE_SMALL) && n_tiles(SHAPE_LARGE) < num_cus / 2) {
                // too few tiles to fill the chip: the latency of one tile is the run time
                shape = SHAPE_SMALL;
                // ... and compute units to spare: two rows per thread (cl, cd)
                if (var != VAR_T && fits(SHAPE_TINY) && n_tiles(SHAPE_TINY) <= num_cus) shape = SHAPE_TINY;
            } else {
                shape = (var == VAR_T) ? SHAPE_OCT : (H <= 8) ? SHAPE_MID : SHAPE_LARGE;
            }
        }
        if (shape == SHAPE_TINY && (var == VAR_T || !fits(SHAPE_TINY))) shape = SHAPE_SMALL;  // (built for cl and cd only)
        if (shape == SHAPE_SMALL && !fits(SHAPE_SMALL)) shape = SHAPE_LARGE;
        if (shape == SHAPE_MID && !fits(SHAPE_MID)) shape = SHAPE_LARGE;
        return shape;
    }
    // a single-launch chain of the cl or cd system on 4-row strips can evaluate its right-hand side itself
    bool strip_can_fuse(int var, int H, int r0, int r1, int ncols, int shape) const {
        if (var == VAR_T || r1 <= r0) return false;
        return strip_shape_for(var, H, r0, r1, ncols, shape) != SHAPE_LARGE;
    }
    // one launch of the register-resident Newton solve chain (tedd_strip.hpp)
    // shape: one of the SHAPE_* of tedd_strip.hpp, or -1 = by mesh size and system: small tiles when
    // the large ones would leave most compute units without work (fewer than half a tile per unit); otherwise
    // 64-row tiles of 8 waves x 8 rows for T and 48-row tiles for cl and cd (tedd_strip.hpp, StripShape).  A shape
    // whose tile the depth H does not fit falls back to the large one.
    template <int VAR, int REACT, bool LAST>
    void solve_strip(int fam, StripArgs A, int ncols, int shape = -1) {
        if (A.r1 <= A.r0) return;
        flush_record();
        if (A.H < 0 || A.H > 16 || A.nsweeps < 0 || A.nsweeps > A.H) throw std::runtime_error("bad strip-solve depth");
        const bool fused = (A.Y != nullptr);
        if (!fused && (A.b_in == nullptr || (A.d_in == nullptr && A.u_prev == nullptr)))
            throw std::runtime_error("strip solve needs rhs/diag and a starting iterate");
        if (fused && (A.fv == nullptr || A.a.Tn == nullptr || (VAR == VAR_CD && A.a.cln == nullptr)))
            throw std::runtime_error("fused strip solve needs Y, the forcing array and the new T (and cl)");
        Scope sc(this, fam);
        const int TIC = kSTile - 2 * strip_col_halo(A.H + (fused ? 1 : 0));  // (as the kernel computes it)
        A.tiles_x = (ncols + TIC - 1) / TIC;
        shape = strip_shape_for(VAR, A.H, A.r0, A.r1, ncols, shape);
        const int TIR = strip_tile_rows(shape) - 2 * A.H;
        A.tiles_y = (A.r1 - A.r0 + TIR - 1) / TIR;
        A.panel = opt_strip_panel;
        const long long nt = (long long)A.tiles_x * A.tiles_y;
        if (nt > 0x7fffffffLL) throw std::runtime_error("mesh too large for the tile index");
        const bool warm = (A.u_prev != nullptr);
        if (warm && A.stats_first == nullptr) throw std::runtime_error("warm strip solve needs a record slot");
        auto launch = [&](auto shape_c, auto warm_c, auto fused_c, auto xface_c) {
            constexpr int SH = decltype(shape_c)::value;
            constexpr bool WM = decltype(warm_c)::value;
            constexpr bool FU = decltype(fused_c)::value;
            constexpr bool XF = decltype(xface_c)::value;
            // (a warm single-launch chain of long T strips parks the lagged unknown of the rows it writes in LDS)
            size_t lds = 0;
            if (LAST && WM && VAR == VAR_T && StripShape<VAR, SH>::R > 4 && !FU)
                lds = (size_t)(std::max(StripShape<VAR, SH>::TR - 2 * A.H, 0) + 1) * kSTile * sizeof(double);
            hipLaunchKernelGGL((k_solve_strip<VAR, REACT, LAST, SH, WM, FU, XF>), dim3((unsigned)nt),
                               dim3(kSTile, StripShape<VAR, SH>::W), lds, stream, A);
        };
        auto by_warm = [&](auto shape_c, auto fused_c, auto xface_c) {
            if (warm)
                launch(shape_c, std::true_type{}, fused_c, xface_c);
            else
                launch(shape_c, std::false_type{}, fused_c, xface_c);
        };
        const bool corr = (A.cp_out != nullptr);
        if (corr && !(fused && !warm && VAR == VAR_CD && REACT == REACT_REG && LAST && strip_shape_carries_corrector(shape)))
            throw std::runtime_error("this strip launch cannot carry the corrector");
        if (corr && (A.cs_iters < 0 || A.cs_iters > kStripCsRegs || !A.ncp || !A.ycs || !A.cs0 || !A.cs_out || !A.gmax || !A.gmin))
            throw std::runtime_error("fused corrector: bad arguments");
        auto by_fused = [&](auto shape_c) {
            constexpr int SH = decltype(shape_c)::value;
            if constexpr (LAST && VAR == VAR_CD && REACT == REACT_REG && (SH == SHAPE_SMALL || SH == SHAPE_TINY)) {
                if (corr) {
                    hipLaunchKernelGGL((k_solve_strip<VAR, REACT, LAST, SH, false, true, false, true>), dim3((unsigned)nt),
                                       dim3(kSTile, StripShape<VAR, SH>::W), 0, stream, A);
                    return;
                }
            }
            if constexpr (LAST && VAR != VAR_T && SH != SHAPE_LARGE) {
                if (fused) {
                    by_warm(shape_c, std::true_type{}, std::false_type{});
                    return;
                }
                if (A.xface) {  // (test hook: the launch a fused one must equal bit for bit)
                    by_warm(shape_c, std::false_type{}, std::true_type{});
                    return;
                }
            }
            if (fused) throw std::runtime_error("this strip launch cannot evaluate its right-hand side");
            by_warm(shape_c, std::false_type{}, std::false_type{});
        };
        if (shape == SHAPE_SMALL)
            by_fused(std::integral_constant<int, SHAPE_SMALL>{});
        else if (shape == SHAPE_TINY)
            by_fused(std::integral_constant<int, (VAR == VAR_T) ? SHAPE_SMALL : SHAPE_TINY>{});
        else if (shape == SHAPE_MID)
            by_fused(std::integral_constant<int, SHAPE_MID>{});
        else if (shape == SHAPE_OCT)
            by_fused(std::integral_constant<int, (VAR == VAR_T) ? SHAPE_OCT : SHAPE_LARGE>{});
        else
            by_fused(std::integral_constant<int, SHAPE_LARGE>{});
        TEDD_HIP(hipGetLastError());
    }
    // the tile shapes of small meshes (the corrector needs ~50 registers more than the 48-row tiles' budget has)
    static bool strip_shape_carries_corrector(int shape) { return shape == SHAPE_SMALL || shape == SHAPE_TINY; }
    // Can the last launch of a cd chain carry the record pass of the cs corrector (k_solve_strip, CORR)?
    bool can_fuse_corrector(int iters) const {
        return !rec_on && !reduce_hook && iters >= 0 && iters <= kStripCsRegs && cs_record_clean >= iters;
    }
    unsigned long long* cs_record_max() { return d_csstats; }
    unsigned long long* cs_record_min() { return d_csstats + 4096; }
    // skip_record: the record pass was carried by the cd solve launch; only the deciding pass is left
    template <int CASE, int REACT>
    void correct(const CorrectIn<CASE, REACT>& a, int r0, int r1, int ncols, double* cp_out, double* cs_out, int iters,
                 double rtol, int* d_keep, bool skip_record = false) {
        check_range(r0, r1, ncols);
        if (r1 == r0) return;
        if constexpr (REACT != REACT_REG) {
            CorrectClosedFn<CASE, REACT> f{a, cp_out, cs_out, d_keep + 1};
            nodes(FAM_CORRECT, f, r0, r1, ncols);
        } else {
            if (rec_on) {
                // (the record must hold its initial values: the prologue of the step saw to it)
                if (std::is_same<CorrectIn<CASE, REACT>, CorrectIn<CASE_STREAM, REACT_REG>>::value && iters <= kPCsRegs &&
                    cs_record_clean >= iters && !reduce_hook) {
                    PDesc& d = rec_push(PO_CS_RECORD, 0, 1, r0, r1, ncols);
                    d.iters = iters;
                    d.rtol = rtol;
                    d.out = cp_out;
                    d.out2 = cs_out;
                    rec_functor(d, a);
                    PDesc& e = rec_push(PO_CS_DECIDE, 0, 1, r0, r1, ncols);
                    e.iters = iters;
                    e.rtol = rtol;
                    e.out = cp_out;
                    e.out2 = cs_out;
                    e.keep = d_keep;
                    rec_functor(e, a);
                    cs_record_clean = 0;
                    return;
                }
                flush_record();
            }
            Scope sc(this, FAM_CS_NEWTON);
            int tx, nt, nb;
            tiling(r0, r1, ncols, tx, nt, nb);
            const bool flat = flat_nodes(r0, r1, ncols);
            if (flat) {  // (runs of 256 nodes instead of 64 x 4 tiles: see k_nodes)
                tx = 0;
                nt = (int)(((long long)(r1 - r0) * ncols + kTileX * kTileY - 1) / (kTileX * kTileY));
                nb = nt < kMaxReduceBlocks ? nt : kMaxReduceBlocks;
            }
            unsigned long long* gmax = d_csstats;
            unsigned long long* gmin = d_csstats + 4096;
            const size_t lds = (size_t)(iters > 0 ? 2 * iters : 2) * sizeof(unsigned long long);
            if (iters > 4096) throw std::runtime_error("more than 4096 cs Newton iterations");
            // as many workgroups as the chip holds at once: each walks over an equal share of the tiles (a
            // grid larger than that runs its surplus as a second, thinner round)
            // (tiled launches of at most 5 iterations -- the default count -- use the build that keeps 5 records per
            // thread and runs 5 waves per SIMD)
            const bool lean = !flat && iters <= 5 && opt_cs_lean != 0;
            static int per_cu[2] = {0, 0};
            if (per_cu[lean] == 0) {
                int n = 0;
                const hipError_t e = lean ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_cs_newton<CASE, false, 5>, kTileX * kTileY, lds)
                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_cs_newton<CASE, false>, kTileX * kTileY, lds);
                if (e != hipSuccess || n < 1) n = 4;
                per_cu[lean] = n;
            }
            nb = std::min(nb, per_cu[lean] * num_cus);
            if (!skip_record) {
                if (iters > 0 && cs_record_clean < iters)
                    hipLaunchKernelGGL(k_cs_init, dim3((iters + 255) / 256), dim3(256), 0, stream, gmax, gmin, iters);
                if (flat)
                    hipLaunchKernelGGL((k_cs_newton<CASE, true>), dim3(nb), dim3(kTileX, kTileY), lds, stream, a, r0, r1, ncols,
                                       tx, nt, cp_out, cs_out, iters, true, rtol, d_keep, gmax, gmin);
                else if (lean)
                    hipLaunchKernelGGL((k_cs_newton<CASE, false, 5>), dim3(nb), dim3(kTileX, kTileY), lds, stream, a, r0, r1, ncols,
                                       tx, nt, cp_out, cs_out, iters, true, rtol, d_keep, gmax, gmin);
                else
                    hipLaunchKernelGGL((k_cs_newton<CASE, false>), dim3(nb), dim3(kTileX, kTileY), lds, stream, a, r0, r1, ncols,
                                       tx, nt, cp_out, cs_out, iters, true, rtol, d_keep, gmax, gmin);
            }
            cs_record_clean = 0;
            // the record holds bit patterns of non-negative doubles (+inf where nothing was seen): their
            // order as doubles is their order as integers, so the hook may reduce them as doubles
            if (reduce_hook && iters > 0) {
                TEDD_HIP(hipGetLastError());
                if (reduce_hook(reduce_user, reinterpret_cast<double*>(gmax), reinterpret_cast<double*>(gmin), iters) != 0)
                    throw std::runtime_error("the reduce hook failed (cs corrector record)");
            }
            // The second pass decides by itself whether anything is left to do, which is almost never: on small
            // meshes it is launched with few workgroups (what it costs is their number: 3.5 us for the 325 of a
            // 256^2 mesh), each of which walks over 16 tiles in the rare case that it has to iterate again.
            const int nb2 = (nt <= 4096) ? std::min(nb, std::max(32, nt / 16)) : nb;
            if (iters > 1 && flat)
                hipLaunchKernelGGL((k_cs_newton<CASE, true>), dim3(nb2), dim3(kTileX, kTileY), lds, stream, a, r0, r1, ncols, tx,
                                   nt, cp_out, cs_out, iters, false, rtol, d_keep, gmax, gmin);
            else if (iters > 1 && lean)
                hipLaunchKernelGGL((k_cs_newton<CASE, false, 5>), dim3(nb2), dim3(kTileX, kTileY), lds, stream, a, r0, r1, ncols, tx,
                                   nt, cp_out, cs_out, iters, false, rtol, d_keep, gmax, gmin);
            else if (iters > 1)
                hipLaunchKernelGGL((k_cs_newton<CASE, false>), dim3(nb2), dim3(kTileX, kTileY), lds, stream, a, r0, r1, ncols, tx,
                                   nt, cp_out, cs_out, iters, false, rtol, d_keep, gmax, gmin);
            else
                hipLaunchKernelGGL(k_cs_decide, dim3(1), dim3(1), 0, stream, gmax, gmin, iters, rtol, d_keep);
            TEDD_HIP(hipGetLastError());
        }
    }
};

}  // namespace tedd
