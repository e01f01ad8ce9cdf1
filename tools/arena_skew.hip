// 21 arrays of 16385 x 16400 doubles carved from ONE allocation at base + k * (S + skew): how does the rate of the
// 13-read + 8-write streaming kernel depend on the distance between the arrays?  (Separate hipMallocs in a fresh
// process put every array on a 2 GiB-aligned physical block: all 21 streams then share their low address bits.)
//   hipcc -O3 --offload-arch=gfx950 tools/arena_skew.hip -o tools/bin/arena_skew
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NR, int NW> struct Ptrs { const double* r[NR]; double* w[NW]; };

template <int NR, int NW>
__global__ void __launch_bounds__(256) k_stream(Ptrs<NR, NW> p, int nrows, int ncols, size_t ld) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= ncols || i >= nrows) return;
    const size_t o = (size_t)i * ld + j;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += __builtin_nontemporal_load(p.r[k] + o);
#pragma unroll
    for (int k = 0; k < NW; ++k) __builtin_nontemporal_store(s + k, p.w[k] + o);
}

double run(double* const* buf, int n, int reps) {
    Ptrs<13, 8> p;
    for (int k = 0; k < 13; ++k) p.r[k] = buf[k];
    for (int k = 0; k < 8; ++k) p.w[k] = buf[13 + k];
    dim3 block(64, 4), grid((n + 63) / 64, (n + 3) / 4);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t ld = ((size_t)n + 15) / 16 * 16;
    k_stream<13, 8><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) k_stream<13, 8><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return 21.0 * (n - 1) * (double)n * 8.0 * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16385;
    const size_t ld = ((size_t)n + 15) / 16 * 16, bytes = ld * n * 8;
    const size_t MB = 1 << 20, GB = 1 << 30;
    const size_t S2M = (bytes + 2 * MB - 1) / (2 * MB) * (2 * MB);
    // first: 21 separate allocations, as the library makes them (the slow placement of a fresh process)
    std::vector<double*> sep(21);
    for (auto& b : sep) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
    printf("21 separate allocations: %.0f %.0f GB/s\n", run(sep.data(), n, 5), run(sep.data(), n, 5));
    const size_t arena_bytes = (argc > 2 && atoi(argv[2]) < 0) ? 200 * GB : 21 * (9 * GB / 2);
    char* arena;
    CK(hipMalloc(&arena, arena_bytes));
    CK(hipMemset(arena, 0, arena_bytes));
    printf("arena of %.1f GB at %p; array bytes %zu, rounded to 2 MB %zu\n", arena_bytes / 1e9, (void*)arena, bytes, S2M);
    if (argc > 2 && atoi(argv[2]) < 0) {
        // (a) strides from 3.5 GiB in steps of 0.5 GiB; (b) arrays in pairs: the two of a pair D apart, the pairs
        // P apart -- at which distance do two streams stop disturbing each other?
        for (size_t h = 7; h <= 19; ++h) {
            const size_t stride = h * GB / 2;
            if ((size_t)20 * stride + bytes > arena_bytes) break;
            std::vector<double*> a(21);
            for (int k = 0; k < 21; ++k) a[k] = (double*)(arena + k * stride);
            printf("stride %.1f GiB: %.0f %.0f GB/s\n", h / 2.0, run(a.data(), n, 5), run(a.data(), n, 5));
        }
        const size_t P = 18 * GB;
        for (size_t q = 0; q <= 14; ++q) {
            const size_t D = S2M + q * GB / 2;
            if ((size_t)10 * P + D + bytes > arena_bytes) break;
            std::vector<double*> a(21);
            // reads and writes mixed over the pairs: array k -> pair k / 2
            for (int k = 0; k < 21; ++k) a[k] = (double*)(arena + (k / 2) * P + (k % 2) * D);
            printf("pairs 18 GiB apart, the two of a pair %.2f GiB apart: %.0f %.0f GB/s\n", D / (double)GB, run(a.data(), n, 5),
                   run(a.data(), n, 5));
        }
        // triples and quadruples packed tightly, the groups far apart
        for (int g = 1; g <= 4; ++g) {
            const size_t Pg = (size_t)g * 5 * GB;
            std::vector<double*> a(21);
            for (int k = 0; k < 21; ++k) a[k] = (double*)(arena + (k / g) * Pg + (k % g) * S2M);
            if ((size_t)(20 / g) * Pg + g * S2M > arena_bytes) continue;
            printf("groups of %d tightly packed arrays, groups %d GiB apart: %.0f %.0f GB/s\n", g, 5 * g, run(a.data(), n, 5),
                   run(a.data(), n, 5));
        }
        return 0;
    }
    if (argc > 2) {   // scan: stride = 2 MB-rounded size + m * step for m = 0 .. count-1   (argv: n step_MB count [extra_bytes])
        const size_t step = (size_t)atoi(argv[2]) * MB;
        const int count = argc > 3 ? atoi(argv[3]) : 32;
        const size_t extra = argc > 4 ? (size_t)atoll(argv[4]) : 0;
        for (int m = 0; m < count; ++m) {
            const size_t stride = S2M + m * step + extra;
            if ((size_t)20 * stride + bytes > arena_bytes) break;
            std::vector<double*> a(21);
            for (int k = 0; k < 21; ++k) a[k] = (double*)(arena + k * stride);
            printf("stride 2MB-rounded + %5zu MB + %zu B = %12zu: %.0f %.0f GB/s\n", m * step / MB, extra, stride, run(a.data(), n, 5),
                   run(a.data(), n, 5));
        }
        return 0;
    }
    struct Case { const char* name; size_t stride; };
    const Case cases[] = {
        {"2 MB-rounded size", S2M}, {"2.5 GiB", 5 * GB / 2}, {"3 GiB", 3 * GB},
        {"size + 128 B", bytes + 128 - bytes % 128}, {"2MB-rounded + 256 B", S2M + 256}, {"2MB-rounded + 1 KB", S2M + 1024},
        {"2MB-rounded + 4 KB", S2M + 4096}, {"2MB-rounded + 4 KB + 256", S2M + 4096 + 256},
        {"2MB-rounded + 16 KB", S2M + 16384}, {"2MB-rounded + 64 KB", S2M + 65536}, {"2MB-rounded + 68 KB + 256", S2M + 65536 + 4096 + 256},
        {"2MB-rounded + 1 MB", S2M + MB}, {"2MB-rounded + 1 MB + 68 KB + 256", S2M + MB + 65536 + 4096 + 256},
        {"2MB-rounded + 34 MB + 68 KB + 256", S2M + 34 * MB + 65536 + 4096 + 256},
        {"2MB-rounded + 512 MB", S2M + 512 * MB}, {"2 MB-rounded size (again)", S2M},
    };
    for (const Case& c : cases) {
        std::vector<double*> a(21);
        for (int k = 0; k < 21; ++k) a[k] = (double*)(arena + k * c.stride);
        if ((size_t)20 * c.stride + bytes > arena_bytes) { printf("%s: does not fit\n", c.name); continue; }
        printf("stride %-36s %12zu: %.0f %.0f GB/s\n", c.name, c.stride, run(a.data(), n, 5), run(a.data(), n, 5));
    }
    // a pseudo-random skew per array (multiples of 256 B below 1 MB)
    {
        std::vector<double*> a(21);
        unsigned x = 12345;
        for (int k = 0; k < 21; ++k) { x = x * 1664525u + 1013904223u; a[k] = (double*)(arena + k * (S2M + 2 * MB) + ((x >> 8) % 4096) * 256); }
        printf("random skew below 1 MB per array: %.0f %.0f GB/s\n", run(a.data(), n, 5), run(a.data(), n, 5));
    }
    printf("21 separate allocations again: %.0f GB/s\n", run(sep.data(), n, 5));
    return 0;
}
