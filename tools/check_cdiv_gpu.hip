// GPU check of bz_cdiv (Markstein division with a tabulated reciprocal) against the IEEE FP64 division hipcc emits.
//   hipcc --offload-arch=gfx950 -O3 tools/check_cdiv_gpu.hip -o /tmp/check_cdiv && /tmp/check_cdiv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ double bz_cdiv(double a, double b, double rb) { const double q0 = a * rb; return fma(fma(-q0, b, a), rb, q0); }
__global__ void k(unsigned long long *bad, double *ex, int mode)
{
    unsigned long long s = 88172645463325252ULL + 977ULL * (blockIdx.x * blockDim.x + threadIdx.x);
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto u01 = [&]() { return (rnd() >> 11) * (1.0 / 9007199254740992.0); };
    unsigned long long nb = 0;
    for (int it = 0; it < 4000; ++it) {
        const double b = 0.05 + 1.25 * u01();
        const double rb = 1.0 / b;
        double a;
        if (mode == 0) a = (u01() - 0.5) * 200.0;
        else if (mode == 1) a = (u01() - 0.5) * 1e-3;
        else if (mode == 2) a = (u01() - 0.5) * exp2(-40.0 * u01());
        else a = 250.0 + 100.0 * u01();
        const double q = bz_cdiv(a, b, rb), r = a / b;
        if (q != r) { ++nb; ex[0] = a; ex[1] = b; ex[2] = q; ex[3] = r; }
    }
    if (nb) atomicAdd(bad, nb);
}
int main()
{
    unsigned long long *bad; double *ex;
    hipMalloc(&bad, 8); hipMalloc(&ex, 32);
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, bad, ex, mode);
        unsigned long long h; double e[4];
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(e, ex, 32, hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches of %llu", mode, h, 1024ULL * 256 * 4000);
        if (h) printf("  e.g. a=%.17g b=%.17g cdiv=%.17g ieee=%.17g", e[0], e[1], e[2], e[3]);
        printf("\n");
    }
    return 0;
}
