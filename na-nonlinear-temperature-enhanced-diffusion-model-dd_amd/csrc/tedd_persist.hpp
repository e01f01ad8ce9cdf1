// Persistent step kernel for small meshes: the launches of one or many predictor-corrector steps
// (src/prob1base.py:3117-3149, with the harness norms of src/mms_trial_utils.py:81-110 between them)
// executed by ONE launch.  The host sequences the steps exactly as it does for separate launches, but
// instead of launching each node functor it records a descriptor (the functor's bytes, its row range, what
// kind of launch it would have been); this kernel walks over the descriptors, every workgroup taking its
// share of the nodes of each, with a barrier over the whole grid where the stream order of the separate
// launches stood.  With one workgroup the barrier is the workgroup barrier; with several it is a counter in
// device memory behind an agent-scope release / acquire pair.  The node functors are the very functions the
// separate launches run (the per-sweep solver, JacobiFn), so the results are bit-identical to those launches.
//
// Every workgroup walks the same descriptor list and meets every barrier, whether it has nodes in a phase
// or not; a barrier that does not fill within kPSpinLimit polls (another program holding compute units, so
// that not all workgroups are resident) raises the abort word and every workgroup leaves.
#pragma once
#include <hip/hip_runtime.h>

#include "tedd_nodes.hpp"

namespace tedd {

enum : int { PO_NODES = 0, PO_MAX, PO_SUM, PO_SUM_FINAL, PO_PROLOGUE, PO_CS_RECORD, PO_CS_DECIDE };
constexpr int kPFunctorBytes = 896;
constexpr int kPWaves = 16;            // waves per workgroup (64 x 16 threads)
constexpr int kPThreads = 64 * kPWaves;
constexpr long long kPSpinLimit = 400000;  // polls of one barrier before the launch gives up (~0.2 s)
constexpr int kPCsRegs = 8;            // cs Newton iterations kept in registers (more are not recorded)

struct alignas(16) PDesc {
    int op, ftype, barrier, r0, r1, ncols;
    int tiles_x, ntiles, nb, nr;  // PO_SUM / PO_SUM_FINAL: the tiling of the stand-alone launch (same sums, same order)
    int iters;                    // cs corrector
    int pad;
    double rtol;
    double* out;   // PO_MAX: record words; PO_SUM: partial sums; PO_SUM_FINAL: the nr sums; cs: cp_out
    double* out2;  // PO_SUM_FINAL: partial sums; cs: cs_out
    int* keep;     // cs decide: iterations kept
    unsigned char f[kPFunctorBytes];
};
struct PersistCtl {
    unsigned* words;   // [0] barrier counter, [1] abort word (device memory)
    unsigned base;     // value of the counter when the launch starts
    int* host_abort;   // host-visible: set when a barrier timed out
    unsigned long long *gmax, *gmin;  // record of the cs corrector
};

template <class... Ts>
struct TypeList {
    static constexpr int size = sizeof...(Ts);
};
template <class T, class L>
struct IndexOf {
    static constexpr int value = -1;
};
template <class T, class... Ts>
struct IndexOf<T, TypeList<T, Ts...>> {
    static constexpr int value = 0;
};
template <class T, class U, class... Ts>
struct IndexOf<T, TypeList<U, Ts...>> {
    static constexpr int rest = IndexOf<T, TypeList<Ts...>>::value;
    static constexpr int value = rest < 0 ? -1 : rest + 1;
};

// the functors the kernel can run, by kind of launch.  (The predictor that evaluates the forcing of the new time
// level itself is not among them: it needs more registers than a wave of a 1024-thread workgroup has; recorded
// steps evaluate the forcing as a phase of its own.)
template <int R>
using PNodeList = TypeList<PredictFn<CASE_STREAM, R, CASE_STREAM>, ForcingFn<CASE_POL, R>, ForcingFn<CASE_EXPSIN, R>,
                           ForcingFn<CASE_PEAKS, R>, ForcingFn<CASE_NFS, R>, NewtonRhsFn<VAR_T, CASE_STREAM, R>,
                           NewtonRhsFn<VAR_CL, CASE_STREAM, R>, NewtonRhsFn<VAR_CD, CASE_STREAM, R>,
                           JacobiFn<VAR_T, R, JAC_PLAIN>, JacobiFn<VAR_CL, R, JAC_PLAIN>, JacobiFn<VAR_CD, R, JAC_PLAIN>>;
template <int R>
using PMaxList = TypeList<JacobiFn<VAR_T, R, JAC_FINAL>, JacobiFn<VAR_CL, R, JAC_FINAL>, JacobiFn<VAR_CD, R, JAC_FINAL>>;
using PSumList = TypeList<ErrNormFn<CASE_POL>, ErrNormFn<CASE_EXPSIN>, ErrNormFn<CASE_PEAKS>, ErrNormFn<CASE_NFS>>;

__device__ __forceinline__ unsigned long long p_load_bits(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// raise a word holding the bit pattern of a non-negative double (most callers do not raise it: they skip the atomic)
__device__ __forceinline__ void p_raise(unsigned long long* p, double v) {
    if (v != v) v = INFINITY;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits > p_load_bits(p)) atomicMax(p, bits);
}
__device__ __forceinline__ void p_lower(unsigned long long* p, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits < p_load_bits(p)) atomicMin(p, bits);
}

// Barrier over the grid.  One workgroup: the workgroup barrier (its waves share a compute unit and its L1).
// Several: every wave waits for its own stores, the workgroup meets, lane 0 writes the L2 back (release),
// adds to the counter and polls it, then invalidates (acquire); the others wait for it at the workgroup barrier.
__device__ __forceinline__ bool p_grid_barrier(const PersistCtl& c, unsigned& passed, int* sh_abort) {
    const unsigned G = gridDim.x;
    if (G == 1) {
        __syncthreads();
        return true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    passed += 1;
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(c.words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = c.base + passed * G;
        int ok = 1;
        for (long long spin = 0;; ++spin) {
            const unsigned v = __hip_atomic_load(c.words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - target) >= 0) break;
            if ((spin & 63) == 63 && __hip_atomic_load(c.words + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                ok = 0;
                break;
            }
            if (spin > kPSpinLimit) {
                __hip_atomic_store(c.words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c.host_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *sh_abort = ok ? 0 : 1;
    }
    __syncthreads();
    return *sh_abort == 0;
}

// ---- phases
// plain node functor: the nodes of the row range dealt to the threads of the grid one by one (a row of a small
// mesh is shorter than a wave, so lanes are not tied to columns)
template <class F>
__device__ __forceinline__ void p_nodes(const PDesc& d) {
    const F& f = *reinterpret_cast<const F*>(d.f);
    const int ncols = d.ncols, n = (d.r1 - d.r0) * ncols;
    for (int q = blockIdx.x * kPThreads + threadIdx.y * 64 + threadIdx.x; q < n; q += gridDim.x * kPThreads) {
        const int row = q / ncols;
        f(d.r0 + row, q - row * ncols);
    }
}
// maximum-reducing functor (the last sweep of a solve): the order of a maximum does not matter
template <class F>
__device__ __forceinline__ void p_max(const PDesc& d) {
    const F& f = *reinterpret_cast<const F*>(d.f);
    double red[F::NR];
#pragma unroll
    for (int k = 0; k < F::NR; ++k) red[k] = 0.0;
    const int ncols = d.ncols, n = (d.r1 - d.r0) * ncols;
    for (int q = blockIdx.x * kPThreads + threadIdx.y * 64 + threadIdx.x; q < n; q += gridDim.x * kPThreads) {
        const int row = q / ncols;
        f(d.r0 + row, q - row * ncols, red);
    }
#pragma unroll
    for (int k = 0; k < F::NR; ++k) {
        const double v = wave_max_nan_inf(red[k]);
        if (threadIdx.x == 0) p_raise(reinterpret_cast<unsigned long long*>(d.out) + k, v);
    }
}
// sum-reducing functor (error norms): the sums of the stand-alone launch in its order.  There a workgroup of
// 4 waves walks over 64 x 4 tiles b, b + nb, ... and leaves one partial sum; here each group of four waves of
// a workgroup stands for one such workgroup at a time.
template <class F>
__device__ __forceinline__ void p_sum(const PDesc& d, double (*sm)[8]) {
    static_assert(F::NR <= 8, "at most 8 sums");
    const F& f = *reinterpret_cast<const F*>(d.f);
    const int lane = threadIdx.x, wave = threadIdx.y, quad = wave >> 2, ty = wave & 3;
    const int per_round = gridDim.x * (kPWaves / 4);
    const int rounds = (d.nb + per_round - 1) / per_round;
    for (int k = 0; k < rounds; ++k) {
        const int vb = (k * gridDim.x + blockIdx.x) * (kPWaves / 4) + quad;
        double red[F::NR];
#pragma unroll
        for (int q = 0; q < F::NR; ++q) red[q] = 0.0;
        if (vb < d.nb) {
            for (int t = vb; t < d.ntiles; t += d.nb) {
                const int j = (t % d.tiles_x) * 64 + lane;
                const int li = d.r0 + (t / d.tiles_x) * 4 + ty;
                if (j < d.ncols && li < d.r1) f(li, j, red);
            }
        }
#pragma unroll
        for (int q = 0; q < F::NR; ++q) {
            const double v = wave_sum_fixed(red[q]);
            if (lane == 0) sm[wave][q] = v;
        }
        __syncthreads();
        if (ty == 0 && lane < F::NR && vb < d.nb)
            d.out[(size_t)vb * F::NR + lane] =
                (sm[4 * quad][lane] + sm[4 * quad + 1][lane]) + (sm[4 * quad + 2][lane] + sm[4 * quad + 3][lane]);
        __syncthreads();
    }
}
// final pass of the sums: one wave per quantity, lane l adds the partials l, l + 64, ... then a fixed butterfly
__device__ __forceinline__ void p_sum_final(const PDesc& d) {
    if (blockIdx.x != 0 || (int)threadIdx.y >= d.nr) return;
    const int k = threadIdx.y;
    double s = 0.0;
    for (int b = threadIdx.x; b < d.nb; b += 64) s += d.out2[(size_t)b * d.nr + k];
    s = wave_sum_fixed(s);
    if (threadIdx.x == 0) d.out[k] = s;
}

// cs corrector of RegHCsTriple (src/prob1base.py:3647-3702): record pass (cp trapezoid, all iterations of
// the scalar Newton iteration, per-iteration max|dx|, min|x| into the global record) and decide pass (reads
// the decision off the record; iterates again, fewer times, only where the reference's global test stops early)
template <int CASE>
__device__ __forceinline__ void p_cs(const PDesc& d, const PersistCtl& c, bool record) {
    const CorrectIn<CASE, REACT_REG>& a = *reinterpret_cast<const CorrectIn<CASE, REACT_REG>*>(d.f);
    const int iters = d.iters;
    int keep = iters;
    if (!record) {
        for (int k = 0; k < iters; ++k) {
            const double mx = __longlong_as_double((long long)p_load_bits(c.gmax + k));
            const double mn = __longlong_as_double((long long)p_load_bits(c.gmin + k));
            if (mx < d.rtol * mn) {
                keep = k + 1;
                break;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) d.keep[0] = keep;
        if (keep >= iters) return;
    }
    double rmx[kPCsRegs], rmn[kPCsRegs];
#pragma unroll
    for (int k = 0; k < kPCsRegs; ++k) {
        rmx[k] = 0.0;
        rmn[k] = INFINITY;
    }
    const int ncols = d.ncols, n = (d.r1 - d.r0) * ncols;
    for (int q = blockIdx.x * kPThreads + threadIdx.y * 64 + threadIdx.x; q < n; q += gridDim.x * kPThreads) {
        const int row = q / ncols;
        const int li = d.r0 + row, j = q - row * ncols;
        const size_t at_c = at(a.g, li, j);
        const bool in = is_interior(a.g, li, j);
        const CsNodeIn v = cs_node_load(a, at_c);
        if (record) put_once(d.out + at_c, corrector_cp_vals(a.m, a.dt, v, in));
        double x, y, av;
        cs_newton_inputs_vals(a.m, a.dt, v, y, av, x);
        double ex = cs_newton_exp(x, a.m.eta);
#pragma unroll
        for (int k = 0; k < kPCsRegs; ++k) {
            if (k < keep) {
                const double dx = cs_newton_step(x, y, av, a.m.eta, ex);
                x = x + dx;
                if (k + 1 < keep) ex = cs_newton_exp_next(ex, x, dx, a.m.eta);
                double adx = fabs(dx), ax = fabs(x);
                if (adx != adx) adx = INFINITY;
                if (ax != ax) ax = 0.0;
                rmx[k] = adx > rmx[k] ? adx : rmx[k];
                rmn[k] = ax < rmn[k] ? ax : rmn[k];
            }
        }
        put_once(d.out2 + at_c, in ? x : 0.0 * x);
    }
    if (record) {
#pragma unroll
        for (int k = 0; k < kPCsRegs; ++k) {
            if (k < iters) {
                const double wmx = wave_max_nn(rmx[k]), wmn = wave_min_nn(rmn[k]);
                if (threadIdx.x == 0) {
                    p_raise(c.gmax + k, wmx);
                    p_lower(c.gmin + k, wmn);
                }
            }
        }
    }
}

template <class L>
struct PRun;
template <class... Fs>
struct PRun<TypeList<Fs...>> {
    static __device__ __forceinline__ void nodes(const PDesc& d) {
        int k = 0;
        ((d.ftype == k++ ? p_nodes<Fs>(d) : (void)0), ...);
    }
    static __device__ __forceinline__ void maxes(const PDesc& d) {
        int k = 0;
        ((d.ftype == k++ ? p_max<Fs>(d) : (void)0), ...);
    }
    static __device__ __forceinline__ void sums(const PDesc& d, double (*sm)[8]) {
        int k = 0;
        ((d.ftype == k++ ? p_sum<Fs>(d, sm) : (void)0), ...);
    }
};

template <int R>
__global__ void __launch_bounds__(kPThreads) k_persist(const PDesc* __restrict__ descs, int ndesc, const PersistCtl ctl) {
    __shared__ double sm[kPWaves][8];
    __shared__ double rec[16];
    __shared__ int sh_abort;
    unsigned passed = 0;
    if (threadIdx.x == 0 && threadIdx.y == 0) sh_abort = 0;
    __syncthreads();
    for (int i = 0; i < ndesc; ++i) {
        const PDesc& d = descs[i];
        if (d.barrier && !p_grid_barrier(ctl, passed, &sh_abort)) return;
        switch (d.op) {
            case PO_NODES: PRun<PNodeList<R>>::nodes(d); break;
            case PO_MAX: PRun<PMaxList<R>>::maxes(d); break;
            case PO_SUM: PRun<PSumList>::sums(d, sm); break;
            case PO_SUM_FINAL: p_sum_final(d); break;
            case PO_PROLOGUE:
                if (blockIdx.x == 0)
                    step_prologue_body(*reinterpret_cast<const StepRecordArgs*>(d.f), ctl.gmax, ctl.gmin,
                                       threadIdx.y * 64 + threadIdx.x, kPThreads, rec);
                break;
            case PO_CS_RECORD:
                if (R == REACT_REG) p_cs<CASE_STREAM>(d, ctl, true);
                break;
            case PO_CS_DECIDE:
                if (R == REACT_REG) p_cs<CASE_STREAM>(d, ctl, false);
                break;
            default: break;
        }
    }
}

}  // namespace tedd
