// tools/dpp_check.hip — do v_mov_b32_dpp wave_shr:1 / wave_shl:1 move a value one lane up / down the whole 64-lane wavefront on gfx950,
// as __shfl_up / __shfl_down (ds_bpermute_b32, 10 ns of the LDS pipe each) do?   hipcc --offload-arch=gfx950 -O3 tools/dpp_check.hip -o tools/dpp_check
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double shr1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shl1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(const double *a, double *up_d, double *dn_d, double *up_s, double *dn_s, int nact)
{
    const int i = threadIdx.x;
    if ((i & 63) >= nact) return;      // a ragged row: the lanes beyond it have left
    const double v = a[i];
    up_d[i] = shr1(v); dn_d[i] = shl1(v);
    up_s[i] = __shfl_up(v, 1); dn_s[i] = __shfl_down(v, 1);
}
int main()
{
    const int n = 256;
    double h[n], r[4][n], *d[5];
    for (int i = 0; i < n; ++i) h[i] = 1000.0 + i + 1.0 / (i + 3);
    for (auto &p : d) hipMalloc(&p, n * sizeof(double));
    int bad = 0;
    for (int nact : {64, 40}) {
        hipMemcpy(d[0], h, sizeof(h), hipMemcpyHostToDevice);
        for (int q = 1; q < 5; ++q) hipMemset(d[q], 0, n * sizeof(double));
        hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, d[0], d[1], d[2], d[3], d[4], nact);
        for (int q = 0; q < 4; ++q) hipMemcpy(r[q], d[q + 1], sizeof(h), hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) {
            const int l = i & 63;
            if (l >= nact) continue;
            if (l > 0 && r[0][i] != r[2][i]) { ++bad; printf("up   lane %d: dpp %g shfl %g\n", i, r[0][i], r[2][i]); }
            if (l < nact - 1 && r[1][i] != r[3][i]) { ++bad; printf("down lane %d: dpp %g shfl %g\n", i, r[1][i], r[3][i]); }
            if (l > 0 && r[0][i] != h[i - 1]) { ++bad; printf("up   lane %d: dpp %g want %g\n", i, r[0][i], h[i - 1]); }
            if (l < nact - 1 && r[1][i] != h[i + 1]) { ++bad; printf("down lane %d: dpp %g want %g\n", i, r[1][i], h[i + 1]); }
        }
    }
    printf(bad ? "dpp_check: %d MISMATCHES\n" : "dpp_check: wave_shr:1 / wave_shl:1 == __shfl_up / __shfl_down (%d)\n", bad);
    return bad != 0;
}
