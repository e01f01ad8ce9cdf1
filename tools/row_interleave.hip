// Does the 13-read + 8-write streaming kernel run faster when the 21 arrays are interleaved row block by row block
// inside ONE allocation instead of lying apart as whole arrays?  Array k, row i, column j lives at
//     base + ((i / B) * G * B + k * B + i % B) * ld + j          (G arrays per group, B rows per block)
// so B = 1 interleaves single rows, B >= nrows with G = 1 is the planar layout (arrays `stride` bytes apart).
//   hipcc -O3 --offload-arch=gfx950 tools/row_interleave.hip -o tools/bin/row_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Layout {
    size_t array_step;  // doubles between array k and k+1 at the same (i, j)
    size_t block_step;  // doubles between row block b and b+1 of one array
    size_t ld;          // doubles between rows inside a block
    int B;              // rows per block
};

template <int NR, int NW>
__global__ void __launch_bounds__(256) k_stream(const double* base, Layout L, int nrows, int ncols) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= ncols || i >= nrows) return;
    const size_t o = (size_t)(i / L.B) * L.block_step + (size_t)(i % L.B) * L.ld + j;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += __builtin_nontemporal_load(base + k * L.array_step + o);
#pragma unroll
    for (int k = 0; k < NW; ++k) __builtin_nontemporal_store(s + k, const_cast<double*>(base) + (NR + k) * L.array_step + o);
}

double run(const double* base, Layout L, int n, int reps) {
    dim3 block(64, 4), grid((n + 63) / 64, (n + 3) / 4);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_stream<13, 8><<<grid, block>>>(base, L, n - 1, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) k_stream<13, 8><<<grid, block>>>(base, L, n - 1, n);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return 21.0 * (n - 1) * (double)n * 8.0 * reps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16385;
    const size_t ld = ((size_t)n + 15) / 16 * 16, bytes = ld * n * 8;
    const size_t MB = 1 << 20;
    const size_t S2M = (bytes + 2 * MB - 1) / (2 * MB) * (2 * MB);
    const size_t arena_bytes = 32 * 2 * S2M;   // what the library takes for 32 buffers two sizes apart
    char* arena;
    CK(hipMalloc(&arena, arena_bytes));
    CK(hipMemset(arena, 0, arena_bytes));
    const double* base = (const double*)arena;
    printf("arena of %.1f GB; array %zu bytes, ld %zu\n", arena_bytes / 1e9, bytes, ld);
    const int huge = 1 << 30;
    for (int rep = 0; rep < 2; ++rep) {
        printf("planar, packed (one size apart):        %.0f %.0f GB/s\n", run(base, Layout{S2M / 8, 0, ld, huge}, n, 5),
               run(base, Layout{S2M / 8, 0, ld, huge}, n, 5));
        printf("planar, two sizes apart:                %.0f %.0f GB/s\n", run(base, Layout{2 * S2M / 8, 0, ld, huge}, n, 5),
               run(base, Layout{2 * S2M / 8, 0, ld, huge}, n, 5));
        for (int G : {21, 32}) {
            for (int B : {1, 2, 4, 8, 16, 32, 64, 128, 256, 1024}) {
                const Layout L{(size_t)B * ld, (size_t)G * B * ld, ld, B};
                if (((size_t)((n + B - 1) / B) * G * B) * ld * 8 > arena_bytes) { printf("G %d B %d does not fit\n", G, B); continue; }
                printf("interleaved, %2d arrays a group, %4d rows a block: %.0f %.0f GB/s\n", G, B, run(base, L, n, 5), run(base, L, n, 5));
            }
        }
        // row pitch padded so that consecutive arrays' rows do not start on the same channel
        for (size_t pad : {(size_t)0, (size_t)32, (size_t)512, (size_t)544}) {
            const size_t ldp = ld + pad;
            const Layout L{ldp, 21 * ldp, ldp, 1};
            printf("interleaved rows of 21, pitch %zu doubles: %.0f %.0f GB/s\n", ldp, run(base, L, n, 5), run(base, L, n, 5));
        }
    }
    return 0;
}
