// TEST INFRASTRUCTURE: host-compiled twin of libtedd_hip.so.
// It compiles the very same node functors and step sequencing (csrc/tedd_nodes.hpp,
// csrc/tedd_ctx.hpp) with a plain-loop backend, exporting the C ABI of include/tedd.h, so that
// the arithmetic and the buffer/ghost-row bookkeeping can be debugged in a container without
// a GPU.  The product package never loads this library; only tests do, by explicit path.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tedd_ctx.hpp"

namespace tedd {

struct CpuBackend {
    static constexpr bool kHasStripSolve = false;
    double fam_ms[FAM_COUNT] = {0};
    long long fam_n[FAM_COUNT] = {0};
    void init(int) {}
    void shutdown() {}
    void use_device() {}
    void set_stream(void*) {}
    void* alloc(size_t bytes) { return std::calloc(bytes ? bytes : 1, 1); }
    void free(void* p) { std::free(p); }
    void memset0(void* p, size_t bytes) { std::memset(p, 0, bytes); }
    template <class F>
    void nodes_prog(int, const F& f, int r0, int r1, int ncols, int n_regs) {
        std::vector<double> regs((size_t)n_regs);
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j) f(li, j, regs.data(), 1);
    }
    void step_reset(double* red, int nred, int) { std::memset(red, 0, nred * sizeof(double)); }
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
    void profile_enable(bool) {}
    void profile_collect() {}

    template <class F>
    void nodes(int, const F& f, int r0, int r1, int ncols) {
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j) f(li, j);
    }
    template <class F>
    void nodes_max(int, const F& f, int r0, int r1, int ncols, double* out) {
        double red[F::NR];
        for (int k = 0; k < F::NR; ++k) red[k] = 0.0;
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j) f(li, j, red);
        for (int k = 0; k < F::NR; ++k) out[k] = (red[k] > out[k] || red[k] != red[k]) ? red[k] : out[k];
    }
    template <class F>
    void nodes_sum(int, const F& f, int r0, int r1, int ncols, double* out) {
        double red[F::NR];
        for (int k = 0; k < F::NR; ++k) red[k] = 0.0;
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j) f(li, j, red);
        for (int k = 0; k < F::NR; ++k) out[k] = red[k];
    }
    template <class F>
    void nodes_sum_noalloc(const F& f, int r0, int r1, int ncols, double* out) {
        nodes_sum(0, f, r0, r1, ncols, out);
    }
    template <int CASE, int REACT>
    void correct(const CorrectIn<CASE, REACT>& a, int r0, int r1, int ncols, double* cp_out, double* cs_out, int iters,
                 double rtol, int* keep_out) {
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
        const size_t n = (size_t)(r1 - r0) * ncols;
        std::vector<double> x(n), y(n), av(n);
        size_t q = 0;
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j, ++q) cs_newton_inputs(a, li, j, y[q], av[q], x[q]);
        int keep = iters;
        for (int k = 0; k < iters; ++k) {
            double mx = 0.0, mn = INFINITY;
            for (q = 0; q < n; ++q) {
                const double dx = cs_newton_step(x[q], y[q], av[q], a.m.eta);
                x[q] += dx;
                double adx = std::fabs(dx), ax = std::fabs(x[q]);
                if (adx != adx) adx = INFINITY;
                if (ax != ax) ax = 0.0;
                mx = adx > mx ? adx : mx;
                mn = ax < mn ? ax : mn;
            }
            if (mx < rtol * mn) {
                keep = k + 1;
                break;
            }
        }
        keep_out[0] = keep;
        q = 0;
        for (int li = r0; li < r1; ++li)
            for (int j = 0; j < ncols; ++j, ++q) cs_out[at(a.g, li, j)] = is_interior(a.g, li, j) ? x[q] : 0.0 * x[q];
    }
};

}  // namespace tedd

#define TEDD_BACKEND tedd::CpuBackend
#include "tedd_api_impl.hpp"
