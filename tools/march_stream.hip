// What does the marching geometry of the library's streaming launches cost by itself?  A wave owns 64 columns and
// walks down a segment of rows (as k_march does: four waves side by side in a workgroup, tiles dealt to the XCDs in
// contiguous runs); per row it reads NR arrays and writes NW, asks for row li + D at the top of row li, and spends
// `work` dependent-chain vector instructions (four chains) between the request and the use.  Arrays lie two sizes
// apart in one allocation, as the library's arena puts them.  Occupancy is set through LDS (workgroups per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/march_stream.hip -o tools/bin/march_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NR, int NW, int D, bool SPLIT = false>
__global__ void __launch_bounds__(256) k_march_stream(const double* base, size_t array_step, size_t ld, int nrows, int ncols,
                                                      int seg_rows, int tiles_x, int work) {
    extern __shared__ double pad[];
    const int lane = threadIdx.x, wave = threadIdx.y;
    const int nb = gridDim.x, bid = blockIdx.x;
    const int q = nb / 8, rem = nb % 8, xcd = bid % 8;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + bid / 8;
    const int ty = t / tiles_x, tx = t % tiles_x;
    const int col0 = (tx * 4 + wave) * 64;
    if (col0 >= ncols) return;
    const int j = min(col0 + lane, ncols - 1);
    const bool col_ok = col0 + lane < ncols;
    const int ra = ty * seg_rows, rb = min(ra + seg_rows, nrows);
    if (work < 0) pad[lane] = 0.0;  // (keeps the LDS allocation alive)
    double ring[D][NR];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int k = 0; k < NR; ++k)
            ring[d][k] = __builtin_nontemporal_load(base + k * array_step + (size_t)min(ra + d, nrows - 1) * ld + j);
    if (SPLIT) {  // (the library's placement of the waits: once before the loop, then between arithmetic and stores)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
    }
    for (int li = ra; li < rb; ++li) {
        double cur[NR], nxt[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) cur[k] = ring[0][k];
        const size_t on = (size_t)min(li + D, nrows - 1) * ld + j;
#pragma unroll
        for (int k = 0; k < NR; ++k) nxt[k] = __builtin_nontemporal_load(base + k * array_step + on);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (k % 4 == 0) s0 += cur[k];
            if (k % 4 == 1) s1 += cur[k];
            if (k % 4 == 2) s2 += cur[k];
            if (k % 4 == 3) s3 += cur[k];
        }
        for (int w = 0; w < work; w += 4) {
            s0 = __builtin_fma(s0, 1.0000001, 0.5);
            s1 = __builtin_fma(s1, 0.9999999, 0.25);
            s2 = __builtin_fma(s2, 1.0000002, 0.125);
            s3 = __builtin_fma(s3, 0.9999998, 0.0625);
        }
        const double s = (s0 + s1) + (s2 + s3);
        if (SPLIT) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
        }
        if (col_ok) {
            const size_t oc = (size_t)li * ld + j;
#pragma unroll
            for (int k = 0; k < NW; ++k) __builtin_nontemporal_store(s + k, const_cast<double*>(base) + (NR + k) * array_step + oc);
        }
#pragma unroll
        for (int d = 0; d + 1 < D; ++d)
#pragma unroll
            for (int k = 0; k < NR; ++k) ring[d][k] = ring[d + 1][k];
#pragma unroll
        for (int k = 0; k < NR; ++k) ring[D - 1][k] = nxt[k];
    }
}

template <int NR, int NW, int D, bool SPLIT = false>
double run(const double* base, size_t array_step, size_t ld, int n, int seg_rows, int wg_per_cu, int work, int reps) {
    const int tiles_x = (n + 255) / 256, tiles_y = (n - 1 + seg_rows - 1) / seg_rows;
    const int lds = wg_per_cu >= 8 ? 0 : (160 * 1024 / wg_per_cu) / 1024 * 1024 - 1024;
    auto kern = k_march_stream<NR, NW, D, SPLIT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 block(64, 4), grid(tiles_x * tiles_y);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, grid, block, lds, 0, base, array_step, ld, n - 1, n, seg_rows, tiles_x, work);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, block, lds, 0, base, array_step, ld, n - 1, n, seg_rows, tiles_x, work);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return (double)(NR + NW) * (n - 1) * (double)n * 8.0 * reps / (ms * 1e-3) / 1e9;
}

template <int NR, int NW>
void table(const char* name, const double* base, size_t step, size_t ld, int n) {
    printf("%s: %d reads + %d writes per node (GB/s); rows: workgroups per CU (x4 waves), columns: rows ahead 1 2 3\n", name, NR, NW);
    for (int work : {0, 256, 564}) {
        for (int seg : {32, 128}) {
            for (int wg : {2, 3, 4, 6, 8}) {
                printf("  work %3d seg %3d wg/CU %d: %5.0f %5.0f %5.0f\n", work, seg, wg, run<NR, NW, 1>(base, step, ld, n, seg, wg, work, 4),
                       run<NR, NW, 2>(base, step, ld, n, seg, wg, work, 4), run<NR, NW, 3>(base, step, ld, n, seg, wg, work, 4));
                fflush(stdout);
            }
        }
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16385;
    const size_t ld = ((size_t)n + 15) / 16 * 16, bytes = ld * n * 8;
    const size_t MB = 1 << 20;
    const size_t S2M = (bytes + 2 * MB - 1) / (2 * MB) * (2 * MB);
    const size_t arena_bytes = 32 * 2 * S2M;
    char* arena;
    if (argc > 2 && argv[2][0] == 'r') {
        // the same arena allocated, streamed and freed again and again: do two physical regions take turns, and does a
        // plain stream tell them apart?  (the library's contexts alternate between two states when made one after the other)
        for (int rep = 0; rep < 8; ++rep) {
            CK(hipMalloc(&arena, arena_bytes));
            CK(hipMemset(arena, 0, arena_bytes));
            const double* b = (const double*)arena;
            const size_t st = 2 * S2M / 8;
            printf("allocation %d at %p: 13 R + 7 W %5.0f GB/s, with 564 instructions a row %5.0f, 6 R + 1 W %5.0f, 4 R + 1 W %5.0f\n", rep,
                   (void*)arena, run<13, 7, 1, true>(b, st, ld, n, 32, 2, 0, 4), run<13, 7, 1, true>(b, st, ld, n, 32, 2, 564, 4),
                   run<6, 1, 1, true>(b, st, ld, n, 32, 4, 0, 4), run<4, 1, 1, true>(b, st, ld, n, 32, 4, 0, 4));
            fflush(stdout);
            CK(hipFree(arena));
        }
        return 0;
    }
    CK(hipMalloc(&arena, arena_bytes));
    CK(hipMemset(arena, 0, arena_bytes));
    const double* base = (const double*)arena;
    if (argc > 2) {  // the ceiling of every read/write mix of the step's launches: no arithmetic, best of the occupancies
        auto best = [&](auto fn) { double b = 0; for (int wg : {2, 3, 4, 6, 8}) { const double v = fn(wg); b = v > b ? v : b; } return b; };
        const size_t st = 2 * S2M / 8;
        printf("marching streams without arithmetic, 32-row segments, one row ahead, best of 2..8 workgroups per CU (GB/s):\n");
        printf("the compiler's waits | the wait placed between the row's sum and its stores\n");
#define MIX(NR_, NW_, what) printf("  %2d R + %d W (%s): %5.0f | %5.0f\n", NR_, NW_, what, \
        best([&](int wg) { return run<NR_, NW_, 1, false>(base, st, ld, n, 32, wg, 0, 4); }), \
        best([&](int wg) { return run<NR_, NW_, 1, true>(base, st, ld, n, 32, wg, 0, 4); }))
        MIX(13, 7, "predictor");
        MIX(6, 1, "rhs_cl");
        MIX(9, 1, "rhs_cd");
        MIX(6, 2, "cs corrector");
        MIX(4, 1, "T, cl solve");
        MIX(6, 1, "cd solve");
        MIX(1, 5, "forcing + 1");
        MIX(1, 1, "copy");
        printf("with 564 instructions of dependent-chain work a row, 13 R + 7 W, by workgroups per CU:\n");
        for (int wg : {2, 3, 4}) printf("  wg/CU %d: %5.0f | %5.0f\n", wg, run<13, 7, 1, false>(base, st, ld, n, 32, wg, 564, 4), run<13, 7, 1, true>(base, st, ld, n, 32, wg, 564, 4));
        return 0;
    }
    table<13, 7>("predictor", base, 2 * S2M / 8, ld, n);
    table<6, 1>("rhs_cl", base, 2 * S2M / 8, ld, n);
    table<9, 1>("rhs_cd", base, 2 * S2M / 8, ld, n);
    return 0;
}
