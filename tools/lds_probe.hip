// Probe of the load-to-LDS instruction of gfx950 (global_load_lds_dwordx4): lane l of a wave writes its 16 bytes at
// byte 16 l of the destination, masked lanes write nothing.  hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/bin/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
__global__ void __launch_bounds__(256) k(const double* src, double* out, int ld, int nrows) {
    __shared__ double buf[4][2][68];
    const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int col0 = (blockIdx.x * 4 + wave) * 64;
    // granule g of 16 bytes covers doubles col0-2+2g, col0-1+2g
    const int g = lane < 34 ? lane : 33;
    int c = col0 - 2 + 2 * g;
    c = c < 0 ? 0 : (c > ld - 2 ? ld - 2 : c);
    for (int r = 0; r < nrows; ++r) {
        const double* p = src + (size_t)r * ld + c;
        if (lane < 34)
            __builtin_amdgcn_global_load_lds((glb_ptr)p, (lds_ptr)&buf[wave][r & 1][0], 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);  // everything
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const double v = buf[wave][r & 1][2 + lane];
        const double lo = buf[wave][r & 1][1 + lane], hi = buf[wave][r & 1][3 + lane];
        out[((size_t)r * ld + col0 + lane) * 3 + 0] = v;
        out[((size_t)r * ld + col0 + lane) * 3 + 1] = lo;
        out[((size_t)r * ld + col0 + lane) * 3 + 2] = hi;
    }
}
int main() {
    const int ld = 512, nrows = 5;
    std::vector<double> h(ld * nrows);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i;
    double *d, *o;
    hipMalloc(&d, h.size() * 8);
    hipMalloc(&o, h.size() * 8 * 3);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(ld / 256), dim3(64, 4), 0, 0, d, o, ld, nrows);
    std::vector<double> ho(h.size() * 3);
    hipError_t e = hipMemcpy(ho.data(), o, ho.size() * 8, hipMemcpyDeviceToHost);
    printf("copy rc %d\n", (int)e);
    int bad = 0;
    for (int r = 0; r < nrows; ++r)
        for (int j = 0; j < ld; ++j) {
            const double v = ho[((size_t)r * ld + j) * 3], lo = ho[((size_t)r * ld + j) * 3 + 1], hi = ho[((size_t)r * ld + j) * 3 + 2];
            const double ev = h[(size_t)r * ld + j];
            const bool ok = v == ev && (j == 0 || lo == ev - 1) && (j == ld - 1 || hi == ev + 1);
            if (!ok && bad++ < 10) printf("r %d j %d: %g %g %g (want %g)\n", r, j, v, lo, hi, ev);
        }
    printf("bad %d\n", bad);
    return bad != 0;
}
