// Streaming ceiling on one GPU: NR arrays read + NW arrays written per node, double precision,
// same launch geometry as the node kernels (64x4 threads, one node per thread) or a grid-stride
// variant, with plain or non-temporal accesses.  Prints achieved GB/s per variant.
//   hipcc -O3 --offload-arch=gfx950 tools/stream_ceiling.hip -o tools/stream_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NR, int NW> struct Ptrs { const double* r[NR]; double* w[NW]; };

template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_stream(Ptrs<NR, NW> p, int nrows, int ncols, size_t ld) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j >= ncols || i >= nrows) return;
    const size_t o = (size_t)i * ld + j;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += NT ? __builtin_nontemporal_load(p.r[k] + o) : p.r[k][o];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        if (NT) __builtin_nontemporal_store(s + k, p.w[k] + o);
        else p.w[k][o] = s + k;
    }
}

// two adjacent nodes per thread: 16-byte accesses
template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) k_stream2(Ptrs<NR, NW> p, int nrows, int ncols, size_t ld) {
    const int j = 2 * (blockIdx.x * 64 + threadIdx.x);
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (j + 1 >= ncols || i >= nrows) return;
    const size_t o = (size_t)i * ld + j;
    double2 s = make_double2(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d v = NT ? __builtin_nontemporal_load((const v2d*)(p.r[k] + o)) : *(const v2d*)(p.r[k] + o);
        s.x += v.x;
        s.y += v.y;
    }
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d v;
        v.x = s.x + k;
        v.y = s.y + k;
        if (NT) __builtin_nontemporal_store(v, (v2d*)(p.w[k] + o));
        else *(v2d*)(p.w[k] + o) = v;
    }
}
template <int NR, int NW, bool NT>
double run2(std::vector<double*>& buf, int n, int reps) {
    Ptrs<NR, NW> p;
    for (int k = 0; k < NR; ++k) p.r[k] = buf[k];
    for (int k = 0; k < NW; ++k) p.w[k] = buf[NR + k];
    const size_t ld = ((size_t)n + 15) / 16 * 16;
    dim3 block(64, 4), grid((n / 2 + 63) / 64, (n + 3) / 4);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_stream2<NR, NW, NT><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) k_stream2<NR, NW, NT><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)(NR + NW) * (n - 1) * (double)(n / 2 * 2) * 8.0 * reps;
    return bytes / (ms * 1e-3) / 1e9;
}

template <int NR, int NW, bool NT>
double run(std::vector<double*>& buf, int n, int reps) {
    Ptrs<NR, NW> p;
    for (int k = 0; k < NR; ++k) p.r[k] = buf[k];
    for (int k = 0; k < NW; ++k) p.w[k] = buf[NR + k];
    dim3 block(64, 4), grid((n + 63) / 64, (n + 3) / 4);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t ld = ((size_t)n + 15) / 16 * 16;  // rows start on 128-byte lines, as in the library
    k_stream<NR, NW, NT><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) k_stream<NR, NW, NT><<<grid, block>>>(p, n - 1, n, ld);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)(NR + NW) * (n - 1) * (double)n * 8.0 * reps;
    return bytes / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16385;
    const int reps = 5;
    std::vector<double*> buf(21);
    for (auto& b : buf) { CK(hipMalloc(&b, (size_t)n * n * 8)); CK(hipMemset(b, 0, (size_t)n * n * 8)); }
    printf("n=%d  (GB/s, algorithmic bytes / time)\n", n);
    printf("copy 1r+1w   plain %.0f  nt %.0f\n", run<1, 1, false>(buf, n, reps), run<1, 1, true>(buf, n, reps));
    printf("read 4r+1w   plain %.0f  nt %.0f\n", run<4, 1, false>(buf, n, reps), run<4, 1, true>(buf, n, reps));
    printf("mix 11r+10w  plain %.0f  nt %.0f\n", run<11, 10, false>(buf, n, reps), run<11, 10, true>(buf, n, reps));
    printf("mix 16r+5w   plain %.0f  nt %.0f\n", run<16, 5, false>(buf, n, reps), run<16, 5, true>(buf, n, reps));
    printf("mix 6r+1w    plain %.0f  nt %.0f\n", run<6, 1, false>(buf, n, reps), run<6, 1, true>(buf, n, reps));
    printf("16-byte accesses (2 nodes per thread):\n");
    printf("copy 1r+1w   plain %.0f  nt %.0f\n", run2<1, 1, false>(buf, n, reps), run2<1, 1, true>(buf, n, reps));
    printf("mix 13r+8w   plain %.0f  nt %.0f\n", run2<13, 8, false>(buf, n, reps), run2<13, 8, true>(buf, n, reps));
    printf("mix 16r+5w   plain %.0f  nt %.0f\n", run2<16, 5, false>(buf, n, reps), run2<16, 5, true>(buf, n, reps));
    printf("8-byte accesses again: mix 13r+8w plain %.0f  nt %.0f\n", run<13, 8, false>(buf, n, reps), run<13, 8, true>(buf, n, reps));
    {   // hipMemcpy device to device
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipMemcpy(buf[1], buf[0], (size_t)n * n * 8, hipMemcpyDeviceToDevice));
        CK(hipEventRecord(a));
        for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(buf[1], buf[0], (size_t)n * n * 8, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("hipMemcpy D2D %.0f\n", 2.0 * n * (double)n * 8.0 * reps / (ms * 1e-3) / 1e9);
    }
    return 0;
}
