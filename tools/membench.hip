// membench.hip — streaming-pattern microbenchmarks for MI355X (standalone; not part of the library).
// hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o gpurun_out/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void copy1(double* __restrict__ y, const double* __restrict__ x, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, s = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += s) y[i] = x[i];
}
__global__ void copy2(double2* __restrict__ y, const double2* __restrict__ x, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, s = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += s) y[i] = x[i];
}
struct RK { double* u[5]; const double* u0[5]; const double* G[5]; };
template <int NF> __global__ void rk1(RK F, long long n, double dt, double a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, s = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += s) {
#pragma unroll
        for (int f = 0; f < NF; ++f) F.u[f][i] = (1 - a) * F.u0[f][i] + a * (F.u[f][i] + dt * F.G[f][i]);
    }
}
template <int NF> __global__ void rk2(RK F, long long n2, double dt, double a) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, s = (long long)gridDim.x * blockDim.x;
    for (; i < n2; i += s) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            double2 u = ((double2*)F.u[f])[i], u0 = ((const double2*)F.u0[f])[i], g = ((const double2*)F.G[f])[i];
            u.x = (1 - a) * u0.x + a * (u.x + dt * g.x); u.y = (1 - a) * u0.y + a * (u.y + dt * g.y);
            ((double2*)F.u[f])[i] = u;
        }
    }
}
// 3-D padded indexing, one thread per cell, block (64,4), z = blockIdx.z
template <int NF> __global__ void rk3d(RK F, int Nx, int Ny, int H, double dt, double a) {
    int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
    if (i >= Nx || j >= Ny) return;
    long long Sx = Nx + 2 * H, Sy = Ny + 2 * H;
    long long n = (i + H) + Sx * ((j + H) + Sy * (long long)(k + H));
#pragma unroll
    for (int f = 0; f < NF; ++f) F.u[f][n] = (1 - a) * F.u0[f][n] + a * (F.u[f][n] + dt * F.G[f][n]);
}
// 3-D padded, block covers a full row segment of 256 x 1, each thread 1 cell
template <int NF> __global__ void rk3d_row(RK F, int Nx, int Ny, int H, double dt, double a) {
    int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= Nx) return;
    long long Sx = Nx + 2 * H, Sy = Ny + 2 * H;
    long long n = (i + H) + Sx * ((j + H) + Sy * (long long)(k + H));
#pragma unroll
    for (int f = 0; f < NF; ++f) F.u[f][n] = (1 - a) * F.u0[f][n] + a * (F.u[f][n] + dt * F.G[f][n]);
}
template <class K> float timeit(K launch, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const int N = 512, H = 3; const long long S = N + 2 * H, n = S * S * S;
    std::vector<double*> bufs(15);
    for (auto& p : bufs) { CK(hipMalloc(&p, n * sizeof(double))); CK(hipMemset(p, 0, n * sizeof(double))); }
    RK F; for (int f = 0; f < 5; ++f) { F.u[f] = bufs[3 * f]; F.u0[f] = bufs[3 * f + 1]; F.G[f] = bufs[3 * f + 2]; }
    double gb = n * 8 / 1e9;
    auto rep = [&](const char* name, float ms, double words) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, words * gb / (ms * 1e-3)); };
    rep("copy 8B/lane grid-stride(8192x256)", timeit([&] { copy1<<<8192, 256>>>(bufs[0], bufs[1], n); }), 2);
    rep("copy 16B/lane grid-stride(8192x256)", timeit([&] { copy2<<<8192, 256>>>((double2*)bufs[0], (const double2*)bufs[1], n / 2); }), 2);
    rep("copy 8B/lane one elem/thread", timeit([&] { copy1<<<(unsigned)((n + 255) / 256), 256>>>(bufs[0], bufs[1], n); }), 2);
    rep("rk 5 fields 8B 1D grid-stride", timeit([&] { rk1<5><<<8192, 256>>>(F, n, 1.0, 0.25); }), 20);
    rep("rk 5 fields 16B 1D grid-stride", timeit([&] { rk2<5><<<8192, 256>>>(F, n / 2, 1.0, 0.25); }), 20);
    rep("rk 5 fields 8B 1D one elem/thread", timeit([&] { rk1<5><<<(unsigned)((n + 255) / 256), 256>>>(F, n, 1.0, 0.25); }), 20);
    rep("rk 1 field x5 launches 8B 1D", timeit([&] { for (int f = 0; f < 5; ++f) { RK G1 = F; G1.u[0] = F.u[f]; G1.u0[0] = F.u0[f]; G1.G[0] = F.G[f]; rk1<1><<<8192, 256>>>(G1, n, 1.0, 0.25); } }), 20);
    rep("rk 1 field x5 launches 16B 1D", timeit([&] { for (int f = 0; f < 5; ++f) { RK G1 = F; G1.u[0] = F.u[f]; G1.u0[0] = F.u0[f]; G1.G[0] = F.G[f]; rk2<1><<<8192, 256>>>(G1, n / 2, 1.0, 0.25); } }), 20);
    double gbi = (double)N * N * N * 8 / 1e9;
    auto repi = [&](const char* name, float ms, double words) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, words * gbi / (ms * 1e-3)); };
    repi("rk 5 fields 3D padded (64x4 blocks per level)", timeit([&] { rk3d<5><<<dim3(N / 64, N / 4, N), dim3(64, 4)>>>(F, N, N, H, 1.0, 0.25); }), 20);
    repi("rk 5 fields 3D padded (256x1 row blocks)", timeit([&] { rk3d_row<5><<<dim3(N / 256, N, N), dim3(256)>>>(F, N, N, H, 1.0, 0.25); }), 20);
    repi("rk 1 field x5 3D padded (256x1 row blocks)", timeit([&] { for (int f = 0; f < 5; ++f) { RK G1 = F; G1.u[0] = F.u[f]; G1.u0[0] = F.u0[f]; G1.G[0] = F.G[f]; rk3d_row<1><<<dim3(N / 256, N, N), dim3(256)>>>(G1, N, N, H, 1.0, 0.25); } }), 20);
    return 0;
}
