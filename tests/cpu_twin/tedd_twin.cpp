// TEST INFRASTRUCTURE: host-compiled twin of libtedd_hip.so.
// It compiles the very same node functors and step sequencing (csrc/tedd_nodes.hpp,
// csrc/tedd_ctx.hpp) with a plain-loop backend, exporting the C ABI of include/tedd.h, so that
// the arithmetic and the buffer/ghost-row bookkeeping can be debugged in a container without
// a GPU.  The product package never loads this library; only tests do, by explicit path.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "tedd_ctx.hpp"

namespace tedd {

struct CpuBackend {
    static constexpr bool kHasStripSolve = false;
    static constexpr bool kHasPersist = false;
    static constexpr bool kHasArena = false;
    static constexpr bool kHasCodegen = false;
    void rec_begin(int, int) {}
    void rec_end() {}
    void rec_discard() noexcept {}
    double fam_ms[FAM_COUNT] = {0};
    long long fam_n[FAM_COUNT] = {0};
    tedd_reduce_fn reduce_hook = nullptr;
    void* reduce_user = nullptr;
    void init(int) {}
    void shutdown() {}
    void use_device() {}
    void set_stream(void*) {}
    void* alloc(size_t bytes) { return std::calloc(bytes ? bytes : 1, 1); }
    void free(void* p) { std::free(p); }
    void* field_alloc(size_t bytes) { return alloc(bytes); }
    void field_free(void* p) { free(p); }
    void memset0(void* p, size_t bytes) { std::memset(p, 0, bytes); }
    // The row loops run under OpenMP on meshes large enough to pay for it (the CPU-baseline leg of the
    // benchmark times this library on one core and on all cores); the small meshes of the tests stay
    // serial.  Sums are taken per row and combined in row order, so a result does not depend on the
    // number of threads.
    static bool wide(int r0, int r1, int ncols) { return (long long)(r1 - r0) * ncols >= 32768; }
    template <class F>
    void nodes_prog(int, const F& f, int r0, int r1, int ncols, int n_regs) {
#pragma omp parallel if (wide(r0, r1, ncols))
        {
            std::vector<double> regs((size_t)n_regs);
#pragma omp for schedule(static)
            for (int li = r0; li < r1; ++li)
                for (int j = 0; j < ncols; ++j) f(li, j, regs.data(), 1);
        }
    }
    void step_prologue(const StepRecordArgs& a) {
        double rec[16];
        std::memcpy(rec, a.red, sizeof(rec));
        if (a.fold)
            for (int v = 0; v < 3; ++v) fold_record(a, rec, v);
        if (a.ring_slot) {
            for (int v = 0; v < 3; ++v) {
                a.ring_slot[1 + v] = rec[2 * v];
                a.ring_slot[4 + v] = rec[2 * v + 1];
                a.ring_slot[7 + v] = rec[8 + 2 * v];
                a.ring_slot[10 + v] = rec[9 + 2 * v];
            }
            a.ring_slot[0] = a.seq;
        }
        if (a.clear) std::memset(a.red, 0, 16 * sizeof(double));
    }
    template <class F>
    void nodes_with_prologue(int fam, const F& f, int r0, int r1, int ncols, const StepRecordArgs& a) {
        step_prologue(a);
        nodes(fam, f, r0, r1, ncols);
    }
    void* alloc_host_visible(size_t bytes) { return std::calloc(bytes ? bytes : 1, 1); }
    void free_host_visible(void* p) { std::free(p); }
    void* device_view(void* p) { return p; }
    void h2d(void* d, const void* s, size_t b) { std::memcpy(d, s, b); }
    void d2h(void* d, const void* s, size_t b) { std::memcpy(d, s, b); }
    void h2d_2d(double* d, size_t dp, const double* s, size_t sp, int w, int rows) {
        for (int r = 0; r < rows; ++r) std::memcpy(d + r * dp, s + r * sp, (size_t)w * sizeof(double));
    }
    void d2h_2d(double* d, size_t dp, const double* s, size_t sp, int w, int rows) { h2d_2d(d, dp, s, sp, w, rows); }
    void d2h_fields(double* const* dst, const double* const* src, int count, size_t sp, int w, int rows) {
        for (int v = 0; v < count; ++v)
            if (dst[v]) d2h_2d(dst[v], w, src[v], sp, w, rows);
    }
    void sync() {}
    void sync_quiet() noexcept {}
    struct DeviceKeeper {};
    void profile_enable(bool) {}
    void profile_collect() {}

    template <class F>
    void nodes(int, const F& f, int r0, int r1, int ncols) {
#pragma omp parallel for schedule(static) if (wide(r0, r1, ncols))
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j) f(li, j);
    }
    template <class F>
    void forcing(int fam, const F& f, int r0, int r1, int ncols) {
        nodes(fam, f, r0, r1, ncols);
    }
    template <class F>
    void nodes_max(int, const F& f, int r0, int r1, int ncols, double* out) {
        std::vector<double> rows((size_t)std::max(r1 - r0, 0) * F::NR, 0.0);
#pragma omp parallel for schedule(static) if (wide(r0, r1, ncols))
        for (int li = r0; li < r1; ++li) {
            double* red = rows.data() + (size_t)(li - r0) * F::NR;
            for (int j = 0; j < ncols; ++j) f(li, j, red);
        }
        for (int li = r0; li < r1; ++li)
            for (int k = 0; k < F::NR; ++k) {
                const double v = rows[(size_t)(li - r0) * F::NR + k];
                out[k] = (v > out[k] || v != v) ? v : out[k];
            }
    }
    template <class F>
    void nodes_sum(int, const F& f, int r0, int r1, int ncols, double* out) {
        std::vector<double> rows((size_t)std::max(r1 - r0, 0) * F::NR, 0.0);
#pragma omp parallel for schedule(static) if (wide(r0, r1, ncols))
        for (int li = r0; li < r1; ++li) {
            double* red = rows.data() + (size_t)(li - r0) * F::NR;
            for (int j = 0; j < ncols; ++j) f(li, j, red);
        }
        for (int k = 0; k < F::NR; ++k) out[k] = 0.0;
        for (int li = r0; li < r1; ++li)
            for (int k = 0; k < F::NR; ++k) out[k] += rows[(size_t)(li - r0) * F::NR + k];
    }
    template <class F>
    void nodes_sum_noalloc(const F& f, int r0, int r1, int ncols, double* out) {
        nodes_sum(0, f, r0, r1, ncols, out);
    }
    template <int CASE, int REACT>
    void correct(const CorrectIn<CASE, REACT>& a, int r0, int r1, int ncols, double* cp_out, double* cs_out, int iters,
                 double rtol, int* keep_out) {
#pragma omp parallel for schedule(static) if (wide(r0, r1, ncols))
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j) cp_out[at(a.g, li, j)] = corrector_cp(a, li, j);
        if (REACT == REACT_CS3) {
            for (int li = r0; li < r1; ++li)
                for (int j = 0; j < ncols; ++j) cs_out[at(a.g, li, j)] = corrector_cs_closed_form(a, li, j);
            return;
        }
        if (REACT == REACT_HCS) {
            bool bad = false;
            for (int li = r0; li < r1; ++li)
                for (int j = 0; j < ncols; ++j) cs_out[at(a.g, li, j)] = corrector_cs_sign_case(a, li, j, &bad);
            if (bad) keep_out[1] = 1;
            return;
        }
        // record pass: all iterations, keeping max|dx_k| and min|x_k| per iteration; then (after the hook
        // has combined the record with that of the other slabs) the decision, and the iteration again if
        // the reference's global test stops it early (src/prob1base.py:3654-3663) -- as the device does
        const size_t n = (size_t)(r1 - r0) * ncols;
        std::vector<double> x0(n), x(n), y(n), av(n), ex(n);
        size_t q = 0;
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j, ++q) cs_newton_inputs(a, li, j, y[q], av[q], x0[q]);
        auto iterate = [&](int count, double* mx, double* mn) {
            x = x0;
            for (size_t qq = 0; qq < n; ++qq) ex[qq] = cs_newton_exp(x[qq], a.m.eta);
            for (int k = 0; k < count; ++k) {
                double vmx = 0.0, vmn = INFINITY;
#pragma omp parallel for schedule(static) reduction(max : vmx) reduction(min : vmn) if (n >= 32768)
                for (long long qq = 0; qq < (long long)n; ++qq) {
                    const double dx = cs_newton_step(x[qq], y[qq], av[qq], a.m.eta, ex[qq]);
                    x[qq] += dx;
                    if (k + 1 < count) ex[qq] = cs_newton_exp_next(ex[qq], x[qq], dx, a.m.eta);
                    double adx = std::fabs(dx), ax = std::fabs(x[qq]);
                    if (adx != adx) adx = INFINITY;
                    if (ax != ax) ax = 0.0;
                    vmx = adx > vmx ? adx : vmx;
                    vmn = ax < vmn ? ax : vmn;
                }
                if (mx) {
                    mx[k] = vmx;
                    mn[k] = vmn;
                }
            }
        };
        std::vector<double> mx((size_t)std::max(iters, 1), 0.0), mn((size_t)std::max(iters, 1), INFINITY);
        iterate(iters, mx.data(), mn.data());
        if (reduce_hook && iters > 0 && reduce_hook(reduce_user, mx.data(), mn.data(), iters) != 0)
            throw std::runtime_error("the reduce hook failed (cs corrector record)");
        int keep = iters;
        for (int k = 0; k < iters; ++k) {
            if (mx[k] < rtol * mn[k]) {
                keep = k + 1;
                break;
            }
        }
        if (keep < iters) iterate(keep, nullptr, nullptr);
        keep_out[0] = keep;
        q = 0;
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j, ++q) cs_out[at(a.g, li, j)] = is_interior(a.g, li, j) ? x[q] : 0.0 * x[q];
    }
};

}  // namespace tedd

#define TEDD_BACKEND tedd::CpuBackend
#include "tedd_api_impl.hpp"
