// Where the time of the launches of a step goes on a small mesh: the kernels built with time stamps (a 100 MHz
// counter all compute units share) at the start of a wave, when its loads have arrived, and after each section
// of its work.  Prints, for the solve launches (T; cl and cd with their right-hand sides fused in; 4 and 2 rows per
// thread), the predictor launch and the record pass of the cs corrector on an N x N mesh (default 256), the span
// of the launch on the device and the median length of each section over the waves.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I na-*/csrc tools/kernel_stamps.hip -o tools/bin/kernel_stamps
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kStampWaves = 16, kStampSlots = 8, kStampTiles = 4096;
static __device__ unsigned long long g_stamps[(size_t)kStampTiles * kStampWaves * kStampSlots];
#if defined(__HIP_DEVICE_COMPILE__)
#define TEDD_STAMP(i)                                                                                              \
    do {                                                                                                           \
        const unsigned bid_ = blockIdx.y * gridDim.x + blockIdx.x;                                                 \
        if (threadIdx.x == 0 && bid_ < kStampTiles)                                                                \
            g_stamps[((size_t)bid_ * kStampWaves + threadIdx.y) * kStampSlots + (i)] = __builtin_amdgcn_s_memrealtime(); \
        if (threadIdx.x == 0 && bid_ < kStampTiles && ((i) == 0 || (i) == 5))                                      \
            g_stamps[((size_t)bid_ * kStampWaves + threadIdx.y) * kStampSlots + 6 + (i) / 5] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define TEDD_STAMP(i)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define TEDD_STAMP_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define TEDD_STAMP_WAIT()
#endif
#include "tedd_hip_backend.hpp"

using namespace tedd;

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

static double* dev(const std::vector<double>& h) {
    double* d = nullptr;
    CK(hipMalloc(&d, h.size() * sizeof(double)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    return d;
}

static const char* kStripSections[6] = {"start after the first wave", "arguments + loads", "matrix rows (+ right-hand side)",
                                        "sweeps", "write-back", "record + drain"};
template <class Launch>
static void report(const char* name, int tiles, int waves, Launch&& launch, const char* const* nm = kStripSections) {
    for (int k = 0; k < 30; ++k) launch();
    CK(hipDeviceSynchronize());
    {
        void* sym = nullptr;
        CK(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_stamps)));
        CK(hipMemset(sym, 0, sizeof(unsigned long long) * (size_t)kStampTiles * kStampWaves * kStampSlots));
    }
    launch();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)kStampTiles * kStampWaves * kStampSlots);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * sizeof(unsigned long long)));
    unsigned long long first = ~0ull, last = 0;
    std::vector<double> sec[6];
    for (int t = 0; t < std::min(tiles, kStampTiles); ++t)
        for (int w = 0; w < waves; ++w) {
            const unsigned long long* s = &st[((size_t)t * kStampWaves + w) * kStampSlots];
            if (s[0] == 0 || s[5] == 0) continue;  // (a wave without a node)
            first = std::min(first, s[0]);
            last = std::max(last, s[5]);
        }
    for (int t = 0; t < std::min(tiles, kStampTiles); ++t)
        for (int w = 0; w < waves; ++w) {
            const unsigned long long* s = &st[((size_t)t * kStampWaves + w) * kStampSlots];
            if (s[0] == 0 || s[5] == 0) continue;
            sec[0].push_back(10.0 * (double)(s[0] - first));
            for (int k = 1; k <= 5; ++k) sec[k].push_back(10.0 * (double)(s[k] - s[k - 1]));
        }
    auto med = [](std::vector<double>& v) {
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    };
    auto mx = [](std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
    const unsigned long long* s0 = &st[0];
    std::printf("%s: %d tiles x %d waves; first wave start to last wave end %.2f us; shader clock %.0f MHz\n", name, tiles,
                waves, 10.0 * (double)(last - first) / 1e3, 100.0 * (double)(s0[7] - s0[6]) / (double)(s0[5] - s0[0]));
    for (int k = 0; k < 6; ++k) std::printf("   %-34s median %7.0f ns   max %7.0f ns\n", nm[k], med(sec[k]), mx(sec[k]));
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 256;
    const int nrows = N + 1, ncols = N + 1, ld = ((ncols + 15) / 16) * 16;
    const double h = 1.0 / N, dt = std::pow(h, 1.5);
    std::vector<double> rows(4 * (size_t)nrows, 0.0), cols(4 * (size_t)ld, 0.0);
    for (int i = 1; i <= N; ++i) {
        rows[i] = 1.0 / h;
        rows[3 * (size_t)nrows + i] = h;
        if (i < N) {
            rows[(size_t)nrows + i] = 1.0 / h;
            rows[2 * (size_t)nrows + i] = h;
        }
    }
    for (int j = 1; j <= N; ++j) {
        cols[j] = 1.0 / h;
        cols[3 * (size_t)ld + j] = h;
        if (j < N) {
            cols[(size_t)ld + j] = 1.0 / h;
            cols[2 * (size_t)ld + j] = h;
        }
    }
    double* d_rows = dev(rows);
    double* d_cols = dev(cols);
    GridDev g{nrows, ncols, ld, 0, N, N, d_rows, d_rows + nrows, d_cols, d_cols + ld, d_rows + 2 * (size_t)nrows,
              d_cols + 2 * (size_t)ld, d_rows + 3 * (size_t)nrows, d_cols + 3 * (size_t)ld};
    ModelC m{1e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-5, 1e-5, 1e-9, 1e-2, 1.0, 1e-6, 1e-5, 1.927e-20, 5e-2, 300.0, 50.0};
    auto field = [&](double a, double b) {
        std::vector<double> f((size_t)nrows * ld, 0.0);
        for (int i = 0; i < nrows; ++i)
            for (int j = 0; j < ncols; ++j) f[(size_t)i * ld + j] = a + b * std::sin(3.0 * i * h) * std::cos(2.0 * j * h);
        return dev(f);
    };
    double *cp1 = field(0.3, 0.2), *cs1 = field(0.04, 0.01), *To = field(0.5, 0.3), *clo = field(-0.2, 0.1),
           *cdo = field(0.4, 0.1), *Tn = field(0.5, 0.31), *cln = field(-0.2, 0.11), *Y = field(0.8, 0.2),
           *fv = field(0.0, 1e-3), *b = field(0.0, 1e-4), *out = field(0.0, 0.0);
    unsigned long long* stats = nullptr;
    CK(hipMalloc(&stats, 64 * sizeof(unsigned long long)));
    CK(hipMemset(stats, 0, 64 * sizeof(unsigned long long)));

    StripArgs A;
    A.g = g;
    A.m = m;
    A.a = NewtonIn{cp1, cs1, To, clo, cdo, Tn, cln, Y, Y, Y};
    A.dt = dt;
    A.r0 = 0;
    A.r1 = nrows;
    A.stats = stats;
    A.panel = 8;
    int shape_rows = SHAPE_SMALL;
    auto tiling = [&](int H, bool fused, int& nt) {
        const int TIC = kSTile - 2 * strip_col_halo(H + (fused ? 1 : 0));
        A.tiles_x = (ncols + TIC - 1) / TIC;
        const int TIR = strip_tile_rows(shape_rows) - 2 * H;
        A.tiles_y = (nrows + TIR - 1) / TIR;
        nt = A.tiles_x * A.tiles_y;
    };
    {
        int nt;
        A.H = A.nsweeps = 9;  // a chain of 10 sweeps from a zero iterate
        A.b_in = A.d_in = b;
        A.d_out = nullptr;
        A.u_out = out;
        tiling(A.H, false, nt);
        report("T chain, 9 sweeps", nt, StripShape<VAR_T, SHAPE_SMALL>::W, [&]() {
            hipLaunchKernelGGL((k_solve_strip<VAR_T, REACT_REG, true, SHAPE_SMALL>), dim3(nt),
                               dim3(kSTile, StripShape<VAR_T, SHAPE_SMALL>::W), 0, 0, A);
        });
    }
    {
        int nt;
        A.H = A.nsweeps = 3;  // 4 sweeps
        A.b_in = A.d_in = nullptr;
        A.Y = Y;
        A.fv = fv;
        tiling(A.H, true, nt);
        report("cd chain with fused right-hand side, 3 sweeps", nt, StripShape<VAR_CD, SHAPE_SMALL>::W, [&]() {
            hipLaunchKernelGGL((k_solve_strip<VAR_CD, REACT_REG, true, SHAPE_SMALL, false, true>), dim3(nt),
                               dim3(kSTile, StripShape<VAR_CD, SHAPE_SMALL>::W), 0, 0, A);
        });
        A.H = A.nsweeps = 4;
        tiling(A.H, true, nt);
        report("cl chain with fused right-hand side, 4 sweeps", nt, StripShape<VAR_CL, SHAPE_SMALL>::W, [&]() {
            hipLaunchKernelGGL((k_solve_strip<VAR_CL, REACT_REG, true, SHAPE_SMALL, false, true>), dim3(nt),
                               dim3(kSTile, StripShape<VAR_CL, SHAPE_SMALL>::W), 0, 0, A);
        });
    }
    // two rows per thread
    A.Y = nullptr;
    A.fv = nullptr;
    {
        int nt;
        shape_rows = SHAPE_TINY;
        A.H = A.nsweeps = 3;
        A.b_in = A.d_in = nullptr;
        A.Y = Y;
        A.fv = fv;
        tiling(A.H, true, nt);
        report("cd chain fused, 3 sweeps, 8 waves x 2 rows", nt, StripShape<VAR_CD, SHAPE_TINY>::W, [&]() {
            hipLaunchKernelGGL((k_solve_strip<VAR_CD, REACT_REG, true, SHAPE_TINY, false, true>), dim3(nt),
                               dim3(kSTile, StripShape<VAR_CD, SHAPE_TINY>::W), 0, 0, A);
        });
        A.H = A.nsweeps = 4;
        tiling(A.H, true, nt);
        report("cl chain fused, 4 sweeps, 8 waves x 2 rows", nt, StripShape<VAR_CL, SHAPE_TINY>::W, [&]() {
            hipLaunchKernelGGL((k_solve_strip<VAR_CL, REACT_REG, true, SHAPE_TINY, false, true>), dim3(nt),
                               dim3(kSTile, StripShape<VAR_CL, SHAPE_TINY>::W), 0, 0, A);
        });
    }
    // ---- predictor launch (array-fed forcing at t0, closed-form ExpSin forcing of t1 evaluated in the launch)
    {
        std::vector<double> xg(N + 1);
        for (int i = 0; i <= N; ++i) xg[i] = i * h;
        MmsParams mp{};
        std::vector<double> nx, qx, ny, qy;
        build_axis_tables(CASE_EXPSIN, mp, xg, 0, nrows, nrows, nx, qx);
        build_axis_tables(CASE_EXPSIN, mp, xg, 0, ncols, ld, ny, qy);
        MmsTabs tb{dev(nx), dev(ny), dev(qx), dev(qy), nrows, ld};
        ForcingSrc f0{}, f1{};
        f0.tm = mms_time_fill(CASE_EXPSIN, m, mp, 0.1);
        f1.tm = mms_time_fill(CASE_EXPSIN, m, mp, 0.1 + dt);
        Out5 fo;
        for (int v = 0; v < 5; ++v) {
            f0.f[v] = field(0.0, 1e-3);
            fo.v[v] = field(0.0, 0.0);
            f1.f[v] = fo.v[v];
        }
        Fields5 S{{cp1, To, clo, cdo, cs1}};
        PredictFn<CASE_STREAM, REACT_REG, CASE_EXPSIN> f{g, m, tb, f0, f1, S, field(0, 0), field(0, 0), field(0, 0),
                                                         field(0, 0), field(0, 0), nullptr, nullptr, nullptr, dt,
                                                         field(0, 0), field(0, 0), field(0, 0), fo};
        double* red = nullptr;
        CK(hipMalloc(&red, 64 * sizeof(double)));
        CK(hipMemset(red, 0, 64 * sizeof(double)));
        StepRecordArgs ra{red, nullptr, 0.0, {1e-13, 1e-13, 1e-13}, 1, 1, 5};
        dim3 block(kTileX, kTileY), grid((ncols + kTileX - 1) / kTileX, (nrows + kTileY - 1) / kTileY);
        static const char* nm[6] = {"start after the first wave", "arguments + loads", "forcing of t1 (tables, 5 stores)",
                                    "field F(S0, t0)", "predictors, b_T, corrector parts", "stores drained"};
        report("predictor launch with the prologue", (int)(grid.x * grid.y), 4, [&]() {
            hipLaunchKernelGGL((k_nodes_pro<decltype(f)>), grid, block, 0, 0, f, 0, nrows, ncols, ra, stats, stats + 32, false);
        }, nm);
        // ---- record pass of the cs corrector
        CorrectIn<CASE_STREAM, REACT_REG> ci{g, m, field(0.3, 0.1), field(0.08, 0.02), cs1, Tn, cln, cdo, dt};
        unsigned long long* cst = nullptr;
        CK(hipMalloc(&cst, 2 * 4096 * sizeof(unsigned long long)));
        std::vector<unsigned long long> init(2 * 4096, 0ull);
        for (int k = 0; k < 4096; ++k) init[4096 + k] = 0x7FF0000000000000ull;
        CK(hipMemcpy(cst, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        int* keep = nullptr;
        CK(hipMalloc(&keep, 64));
        const int tx = (ncols + kTileX - 1) / kTileX, nt = tx * ((nrows + kTileY - 1) / kTileY);
        double *cpo = field(0, 0), *cso = field(0, 0);
        static const char* nc[6] = {"start after the first wave", "arguments + loads", "cp trapezoid + 5 Newton iterations",
                                    "wave reductions + LDS record", "workgroup barrier", "global record + drain"};
        report("cs corrector, record pass", nt, 4, [&]() {
            hipLaunchKernelGGL((k_cs_newton<CASE_STREAM, false>), dim3(nt), dim3(kTileX, kTileY), 10 * sizeof(unsigned long long), 0, ci,
                               0, nrows, ncols, tx, nt, cpo, cso, 5, true, 1e-6, keep, cst, cst + 4096);
        }, nc);
    }
    return 0;
}
