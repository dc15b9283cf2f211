// kbench_mix.hip — traffic-only stand-in for the sparse-bitmask COMPRESS byte mix at 8192^2 bf16, 50 % dense: 134.2 MB read, 67.1 MB of
// "values" + 8.4 MB of "bitmask" written, NO dependency between workgroups (what a compress kernel could reach if the prefix over the tensor
// were free).  Shapes: (a) lane = 4 consecutive units in (64 B), 2 x 16 B nt out + 4 B out; (b) lane = 1 unit per step, 4 steps one block
// apart (16 B in each), 8 B nt out each + 1 B... folded into one 4 B store.  HBM-cold rotation, 5 blocks x 60 launches, median.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 kbench_mix.hip -o kbench_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void mix_a(const u32x4* __restrict__ in, u32x4* __restrict__ vals, uint32_t* __restrict__ mask, int64_t lanes) {
    const int64_t l = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (l >= lanes) return;
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = in[l * 4 + i];
    const u32x4 a = {r[0].x ^ r[1].y, r[0].z ^ r[1].w, r[1].x ^ r[0].y, r[1].z ^ r[0].w};
    const u32x4 b = {r[2].x ^ r[3].y, r[2].z ^ r[3].w, r[3].x ^ r[2].y, r[3].z ^ r[2].w};
    __builtin_nontemporal_store(a, vals + l * 2);
    __builtin_nontemporal_store(b, vals + l * 2 + 1);
    __builtin_nontemporal_store(a.x + b.y, mask + l);
}
// lane = one unit per step, U steps one block apart (1 KiB contiguous per wave load instruction); 8 B of values per unit
template <int U>
__global__ __launch_bounds__(kBlock) void mix_b(const u32x4* __restrict__ in, u32x2* __restrict__ vals, uint8_t* __restrict__ mask, int64_t units) {
    const int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x;
    u32x4 r[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { const int64_t u = base + (int64_t)i * kBlock; if (u < units) r[i] = in[u]; }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u >= units) continue;
        __builtin_nontemporal_store(u32x2{r[i].x ^ r[i].z, r[i].y ^ r[i].w}, vals + u);
        mask[u] = (uint8_t)(r[i].x + r[i].w);
    }
}
// ---- W4-compress-shaped mix: 134.2 MB in, 33.5 MB out
__global__ __launch_bounds__(kBlock) void cmp_a(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t lanes) {  // the shipped shape: 64 B per lane in, 16 B out
    const int64_t l = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (l >= lanes) return;
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = in[l * 4 + i];
    __builtin_nontemporal_store(u32x4{r[0].x ^ r[0].y ^ r[0].z ^ r[0].w, r[1].x ^ r[1].y ^ r[1].z ^ r[1].w, r[2].x ^ r[2].y ^ r[2].z ^ r[2].w, r[3].x ^ r[3].y ^ r[3].z ^ r[3].w}, out + l);
}
template <int U>
__global__ __launch_bounds__(kBlock) void cmp_b(const u32x4* __restrict__ in, uint32_t* __restrict__ out, int64_t units) {  // 16 B per lane and step in (1 KiB per wave instruction), 4 B out
    const int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x;
    u32x4 r[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { const int64_t u = base + (int64_t)i * kBlock; if (u < units) r[i] = in[u]; }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u < units) __builtin_nontemporal_store(r[i].x ^ r[i].y ^ r[i].z ^ r[i].w, out + u);
    }
}
// sparse-compress mix with the resident kernel's store shape: 16 B per lane, wave-contiguous, at a 2-byte-aligned (SHIFT = 2) or 16-byte-aligned address
typedef u32x4 u32x4_a2 __attribute__((aligned(2)));
__global__ __launch_bounds__(kBlock) void mix_d(const u32x4* __restrict__ in, uint8_t* __restrict__ vals, uint8_t* __restrict__ mask, int64_t units, int shift) {
    const int64_t base = (int64_t)blockIdx.x * kBlock * 2 + threadIdx.x;
    if (base + kBlock >= units) return;
    const u32x4 a = in[base], b = in[base + kBlock];
    const int64_t o = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 16 + shift;
    __builtin_nontemporal_store(u32x4{a.x ^ a.z, a.y ^ a.w, b.x ^ b.z, b.y ^ b.w}, reinterpret_cast<u32x4_a2*>(vals + o));
    mask[base] = (uint8_t)(a.x + a.w);
    mask[base + kBlock] = (uint8_t)(b.x + b.w);
}
__global__ void fill(uint32_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = ((uint32_t)i + seed) * 0x9E3779B1u;
}
static double timed(const std::function<void(int)>& fn, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 300; ++i) fn(i);
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    std::vector<double> per;
    for (int blk = 0; blk < 5; ++blk) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) fn(i);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); per.push_back(ms * 1000.0 / iters);
    }
    std::sort(per.begin(), per.end());
    return per[2];
}
int main() {
    const int64_t n = 8192, e = n * n, units = e / 8, lanes = units / 4;
    std::vector<u32x4*> in;
    for (int i = 0; i < 6; ++i) { u32x4* p; CK(hipMalloc(&p, e * 2)); hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)p, e / 2, 17u * i); in.push_back(p); }
    void* vals[2]; void* mask[2];
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&vals[i], e + 64)); CK(hipMalloc(&mask[i], units)); }
    CK(hipDeviceSynchronize());
    const double bytes = 2.0 * e + 1.0 * e + e / 8.0;
    for (int rep = 0; rep < 2; ++rep) {
        double us = timed([&](int i) { hipLaunchKernelGGL(mix_a, dim3((unsigned)((lanes + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, in[i % 6], (u32x4*)vals[i & 1], (uint32_t*)mask[i & 1], lanes); }, 60);
        printf("mix (a) lane = 64 B in, 2 x 16 B + 4 B out          : %7.2f us  %7.1f GB/s  %5.2f %% of 8 TB/s\n", us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
        us = timed([&](int i) { hipLaunchKernelGGL((mix_b<2>), dim3((unsigned)((units + kBlock * 2 - 1) / (kBlock * 2))), dim3(kBlock), 0, 0, in[i % 6], (u32x2*)vals[i & 1], (uint8_t*)mask[i & 1], units); }, 60);
        printf("mix (b) lane = 2 x (16 B in, 8 B + 1 B out)          : %7.2f us  %7.1f GB/s  %5.2f %%\n", us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
        us = timed([&](int i) { hipLaunchKernelGGL((mix_b<4>), dim3((unsigned)((units + kBlock * 4 - 1) / (kBlock * 4))), dim3(kBlock), 0, 0, in[i % 6], (u32x2*)vals[i & 1], (uint8_t*)mask[i & 1], units); }, 60);
        printf("mix (b) lane = 4 x (16 B in, 8 B + 1 B out)          : %7.2f us  %7.1f GB/s  %5.2f %%\n", us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
        for (int shift : {0, 2, 4, 6, 8, 12}) {
            us = timed([&](int i) { hipLaunchKernelGGL(mix_d, dim3((unsigned)((units + kBlock * 2 - 1) / (kBlock * 2))), dim3(kBlock), 0, 0, in[i % 6], (uint8_t*)vals[i & 1], (uint8_t*)mask[i & 1], units, shift); }, 60);
            printf("mix (d) 2 x 16 B in, ONE 16 B out at offset %% 16 == %d  : %7.2f us  %7.1f GB/s  %5.2f %%\n", shift, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
        }
        const double cb = 2.0 * e + 0.5 * e;
        us = timed([&](int i) { hipLaunchKernelGGL(cmp_a, dim3((unsigned)((lanes + kBlock - 1) / kBlock)), dim3(kBlock), 0, 0, in[i % 6], (u32x4*)vals[i & 1], lanes); }, 60);
        printf("compress-shaped (a) lane = 64 B in, 16 B out         : %7.2f us  %7.1f GB/s  %5.2f %%\n", us, cb / us / 1e3, cb / us / 1e3 / 80.0);
        us = timed([&](int i) { hipLaunchKernelGGL((cmp_b<2>), dim3((unsigned)((units + kBlock * 2 - 1) / (kBlock * 2))), dim3(kBlock), 0, 0, in[i % 6], (uint32_t*)vals[i & 1], units); }, 60);
        printf("compress-shaped (b) lane = 2 x (16 B in, 4 B out)    : %7.2f us  %7.1f GB/s  %5.2f %%\n", us, cb / us / 1e3, cb / us / 1e3 / 80.0);
        us = timed([&](int i) { hipLaunchKernelGGL((cmp_b<4>), dim3((unsigned)((units + kBlock * 4 - 1) / (kBlock * 4))), dim3(kBlock), 0, 0, in[i % 6], (uint32_t*)vals[i & 1], units); }, 60);
        printf("compress-shaped (b) lane = 4 x (16 B in, 4 B out)    : %7.2f us  %7.1f GB/s  %5.2f %%\n", us, cb / us / 1e3, cb / us / 1e3 / 80.0);
        us = timed([&](int i) { hipLaunchKernelGGL((cmp_b<8>), dim3((unsigned)((units + kBlock * 8 - 1) / (kBlock * 8))), dim3(kBlock), 0, 0, in[i % 6], (uint32_t*)vals[i & 1], units); }, 60);
        printf("compress-shaped (b) lane = 8 x (16 B in, 4 B out)    : %7.2f us  %7.1f GB/s  %5.2f %%\n", us, cb / us / 1e3, cb / us / 1e3 / 80.0);
        fflush(stdout);
    }
    return 0;
}
