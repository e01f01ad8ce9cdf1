// Does the streaming rate of a kernel depend on WHICH device memory its arrays lie in?  Sets of 21 arrays of
// 16385^2 doubles (the predictor's 13 reads + 8 writes at the bench mesh) are allocated one after the other
// in one process -- set 0 is the first memory the process is given -- and the same streaming kernel is timed
// on every set, on sets mixed from two of them, on pairs of arrays (copy) and again after set 0 was freed and
// allocated anew.
//   hipcc -O3 --offload-arch=gfx950 tools/region_bandwidth.hip -o tools/bin/region_bandwidth
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

template <int NR, int NW>
double run(double* const* buf, int n, int reps) {
    Ptrs<NR, NW> p;
    for (int k = 0; k < NR; ++k) p.r[k] = buf[k];
    for (int k = 0; k < NW; ++k) p.w[k] = buf[NR + k];
    dim3 block(64, 4), grid((n + 63) / 64, (n + 3) / 4);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t ld = ((size_t)n + 15) / 16 * 16;
    k_stream<NR, NW><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) k_stream<NR, NW><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    const double bytes = (double)(NR + NW) * (n - 1) * (double)n * 8.0 * reps;
    return bytes / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16385;
    const int nsets = argc > 2 ? atoi(argv[2]) : 5;
    const int reps = 5;
    const size_t ld = ((size_t)n + 15) / 16 * 16, bytes = ld * n * 8;
    std::vector<std::vector<double*>> set(nsets, std::vector<double*>(21));
    for (int s = 0; s < nsets; ++s)
        for (auto& b : set[s]) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
    CK(hipDeviceSynchronize());
    printf("n=%d, %d sets of 21 arrays of %.3f GB (GB/s, 13 reads + 8 writes per node)\n", n, nsets, bytes / 1e9);
    for (int s = 0; s < nsets; ++s) printf("set %d: first array at %p, last at %p\n", s, (void*)set[s][0], (void*)set[s][20]);
    for (int round = 0; round < 3; ++round) {
        printf("round %d:", round);
        for (int s = 0; s < nsets; ++s) printf("  set%d %.0f", s, run<13, 8>(set[s].data(), n, reps));
        printf("\n");
    }
    printf("copies inside a set (array 2k -> 2k+1):\n");
    for (int s = 0; s < nsets; ++s) {
        printf("  set%d:", s);
        for (int k = 0; k + 1 < 21; k += 2) printf(" %.0f", run<1, 1>(set[s].data() + k, n, reps));
        printf("\n");
    }
    if (nsets >= 2) {
        std::vector<double*> mix(21), mix2(21);
        for (int k = 0; k < 21; ++k) { mix[k] = set[k & 1][k]; mix2[k] = set[k < 13 ? 0 : 1][k]; }
        printf("arrays alternating between set 0 and set 1: %.0f   reads from set 0, writes to set 1: %.0f\n",
               run<13, 8>(mix.data(), n, reps), run<13, 8>(mix2.data(), n, reps));
        for (int k = 0; k < 21; ++k) mix2[k] = set[k < 13 ? 1 : 0][k];
        printf("reads from set 1, writes to set 0: %.0f\n", run<13, 8>(mix2.data(), n, reps));
    }
    // set 0 freed and allocated again (the other sets stay): same memory back?
    for (auto& b : set[0]) CK(hipFree(b));
    for (auto& b : set[0]) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
    printf("set 0 freed and allocated again: first array at %p: %.0f %.0f\n", (void*)set[0][0],
           run<13, 8>(set[0].data(), n, reps), run<13, 8>(set[0].data(), n, reps));
    // set 0 freed array by array in reverse and allocated again
    for (int k = 20; k >= 0; --k) CK(hipFree(set[0][k]));
    for (auto& b : set[0]) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
    printf("... and once more (freed in reverse order): first array at %p: %.0f %.0f\n", (void*)set[0][0],
           run<13, 8>(set[0].data(), n, reps), run<13, 8>(set[0].data(), n, reps));
    return 0;
}
