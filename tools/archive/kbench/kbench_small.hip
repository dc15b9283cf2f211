// kbench_small.hip — developer micro-benchmark (not part of the product): which launch SHAPE is fastest for the small
// single-tensor launches north_star names (4096x4096: 42-50 MB per launch, ONE residency round of the chip), where the
// per-launch fixed cost (boundary + ramp + tail, ~2 us) is a quarter of the time.  Traffic-shaped stand-ins of the four
// kernels (same bytes per lane, same access shapes, a little arithmetic), HBM-cold rotation, HIP events, median of 5 blocks.
//   compress-shaped   (W4 quant+pack):   64 B in per lane (4 x 16 B contiguous) -> 16 B out
//   decompress-shaped (W4 unpack+dequant): 4 B in -> 16 B out, U units per lane one block apart
//   q8 quant-shaped:   32 B in -> 16 B out;   q8 dequant-shaped: 8 B in -> 16 B out, U units per lane
// Variants: F chunks per workgroup with all loads up front ("fat"), block size, units per lane, and the same launches
// alternating between two streams (what hides the boundary when the tensors are independent).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 kbench_small.hip -o kbench_small ; run: ./kbench_small [n=4096]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <functional>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void st16(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ uint32_t mix(uint32_t a, uint32_t b) { return (a * 0x9E3779B1u) ^ (b + 0x7F4A7C15u); }

// ---- compress-shaped: lane = 4 consecutive 16-byte vectors in, one 16-byte vector out; F chunks of BLOCK lanes per workgroup
template <int BLOCK, int F>
__global__ __launch_bounds__(BLOCK) void comp_kernel(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale, u32x4* __restrict__ out, int64_t lanes) {
    u32x4 r[F][4];
    uint32_t s[F];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int64_t g = ((int64_t)blockIdx.x * F + f) * BLOCK + threadIdx.x;
        if (g < lanes) {
            s[f] = scale[g >> 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[f][i] = in[g * 4 + i];
        }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int64_t g = ((int64_t)blockIdx.x * F + f) * BLOCK + threadIdx.x;
        if (g >= lanes) continue;
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = mix(mix(r[f][i].x, r[f][i].y), mix(r[f][i].z, r[f][i].w)) + s[f];
        st16(out + g, u32x4{w[0], w[1], w[2], w[3]});
    }
}

// ---- decompress-shaped: lane = U words one block apart in (BYTES_IN each: 4 or 8), 16 bytes out each
template <int BLOCK, int U, int BYTES_IN>
__global__ __launch_bounds__(BLOCK) void decomp_kernel(const void* __restrict__ in, const uint16_t* __restrict__ scale, u32x4* __restrict__ out, int64_t units) {
    uint32_t lo[U], hi[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = (int64_t)blockIdx.x * BLOCK * U + (int64_t)i * BLOCK + threadIdx.x;
        if (u < units) {
            if constexpr (BYTES_IN == 4) { lo[i] = static_cast<const uint32_t*>(in)[u]; hi[i] = 0; }
            else { const u32x2 v = static_cast<const u32x2*>(in)[u]; lo[i] = v.x; hi[i] = v.y; }
        }
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = (int64_t)blockIdx.x * BLOCK * U + (int64_t)i * BLOCK + threadIdx.x;
        if (u >= units) continue;
        const uint32_t s = scale[u >> 4];
        st16(out + u, u32x4{mix(lo[i], s), mix(hi[i], s), lo[i] ^ s, hi[i] + s});
    }
}

// ---- q8 quant-shaped: lane = 2 consecutive 16-byte vectors in, 16 bytes out; F chunks per workgroup
template <int BLOCK, int F>
__global__ __launch_bounds__(BLOCK) void q8q_kernel(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale, u32x4* __restrict__ out, int64_t lanes) {
    u32x4 r[F][2];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int64_t g = ((int64_t)blockIdx.x * F + f) * BLOCK + threadIdx.x;
        if (g < lanes) { r[f][0] = in[g * 2]; r[f][1] = in[g * 2 + 1]; }
    }
    const uint32_t s = scale[0];
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int64_t g = ((int64_t)blockIdx.x * F + f) * BLOCK + threadIdx.x;
        if (g >= lanes) continue;
        st16(out + g, u32x4{mix(r[f][0].x, r[f][0].y) + s, mix(r[f][0].z, r[f][0].w), mix(r[f][1].x, r[f][1].y), mix(r[f][1].z, r[f][1].w)});
    }
}

__global__ void null_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    // median over 5 blocks of `iters` launches, after >= 30 ms of the same launches
    double run(const std::function<void(int)>& fn, int iters, hipStream_t s = 0) {
        for (int i = 0; i < 2000; ++i) fn(i);
        CK(hipDeviceSynchronize());
        std::vector<double> per;
        for (int blk = 0; blk < 5; ++blk) {
            CK(hipEventRecord(a, s));
            for (int i = 0; i < iters; ++i) fn(i);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            per.push_back(ms * 1000.0 / iters);
        }
        std::sort(per.begin(), per.end());
        return per[2];
    }
};

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const int64_t elems = n * n, units = elems / 8, lanes4 = units / 4, lanes2 = units / 2;
    const int nsets = (int)std::max<int64_t>(6, (int64_t)(2 * 256 * 1048576LL) / (elems / 2) + 1);  // packed stream >= 2x the Infinity Cache
    printf("# n=%lld sets=%d\n", (long long)n, nsets);
    std::vector<void*> w(nsets), pk(nsets), q8(nsets), o16(nsets);
    void* scale;
    CK(hipMalloc(&scale, elems / 64));
    CK(hipMemset(scale, 0x3c, elems / 64));
    for (int i = 0; i < nsets; ++i) {
        CK(hipMalloc(&w[i], elems * 2)); CK(hipMemset(w[i], 0x11 + i, elems * 2));
        CK(hipMalloc(&pk[i], elems / 2)); CK(hipMemset(pk[i], 0x22 + i, elems / 2));
        CK(hipMalloc(&q8[i], elems)); CK(hipMemset(q8[i], 0x33 + i, elems));
    }
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    Timer T;
    const int iters = n <= 4096 ? 200 : 60;
    const double bw4 = (2.0 + 0.5 + 2.0 / 128) * elems, bq8 = 3.0 * elems;
    auto rep = [&](const char* name, double us, double bytes) { printf("%-44s %8.2f us  %7.1f GB/s  %5.1f %%\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0); fflush(stdout); };
    int* flag; CK(hipMalloc(&flag, 4));
    rep("null kernel back to back (boundary)", T.run([&](int) { hipLaunchKernelGGL(null_kernel, dim3(1), dim3(64), 0, 0, flag); }, 2000), 0);

#define COMP(B, F, STREAM) hipLaunchKernelGGL((comp_kernel<B, F>), dim3((unsigned)((lanes4 + (int64_t)B * F - 1) / ((int64_t)B * F))), dim3(B), 0, STREAM, \
        (const u32x4*)w[i % nsets], (const uint16_t*)scale, (u32x4*)pk[i % nsets], lanes4)
    rep("compress  block 256, 1 chunk  (product shape)", T.run([&](int i) { COMP(256, 1, 0); }, iters), bw4);
    rep("compress  block 256, 2 chunks", T.run([&](int i) { COMP(256, 2, 0); }, iters), bw4);
    rep("compress  block 256, 4 chunks", T.run([&](int i) { COMP(256, 4, 0); }, iters), bw4);
    rep("compress  block 512, 1 chunk", T.run([&](int i) { COMP(512, 1, 0); }, iters), bw4);
    rep("compress  block 1024, 1 chunk", T.run([&](int i) { COMP(1024, 1, 0); }, iters), bw4);
    rep("compress  block 128, 1 chunk", T.run([&](int i) { COMP(128, 1, 0); }, iters), bw4);
    rep("compress  block 256, 1 chunk, two streams", T.run([&](int i) { if (i & 1) COMP(256, 1, s2); else COMP(256, 1, s1); }, iters), bw4);

#define DECOMP(B, U, STREAM) hipLaunchKernelGGL((decomp_kernel<B, U, 4>), dim3((unsigned)((units + (int64_t)B * U - 1) / ((int64_t)B * U))), dim3(B), 0, STREAM, \
        (const void*)pk[(i + nsets / 2) % nsets], (const uint16_t*)scale, (u32x4*)w[i % nsets], units)
    rep("decompress block 256, U=1", T.run([&](int i) { DECOMP(256, 1, 0); }, iters), bw4);
    rep("decompress block 256, U=2 (product shape)", T.run([&](int i) { DECOMP(256, 2, 0); }, iters), bw4);
    rep("decompress block 256, U=4", T.run([&](int i) { DECOMP(256, 4, 0); }, iters), bw4);
    rep("decompress block 256, U=8", T.run([&](int i) { DECOMP(256, 8, 0); }, iters), bw4);
    rep("decompress block 512, U=2", T.run([&](int i) { DECOMP(512, 2, 0); }, iters), bw4);
    rep("decompress block 512, U=4", T.run([&](int i) { DECOMP(512, 4, 0); }, iters), bw4);
    rep("decompress block 1024, U=2", T.run([&](int i) { DECOMP(1024, 2, 0); }, iters), bw4);
    rep("decompress block 256, U=2, two streams", T.run([&](int i) { if (i & 1) DECOMP(256, 2, s2); else DECOMP(256, 2, s1); }, iters), bw4);

#define Q8Q(B, F) hipLaunchKernelGGL((q8q_kernel<B, F>), dim3((unsigned)((lanes2 + (int64_t)B * F - 1) / ((int64_t)B * F))), dim3(B), 0, 0, \
        (const u32x4*)w[i % nsets], (const uint16_t*)scale, (u32x4*)q8[i % nsets], lanes2)
    rep("q8 quant  block 256, 1 chunk  (product shape)", T.run([&](int i) { Q8Q(256, 1); }, iters), bq8);
    rep("q8 quant  block 256, 2 chunks", T.run([&](int i) { Q8Q(256, 2); }, iters), bq8);
    rep("q8 quant  block 256, 4 chunks", T.run([&](int i) { Q8Q(256, 4); }, iters), bq8);
    rep("q8 quant  block 512, 2 chunks", T.run([&](int i) { Q8Q(512, 2); }, iters), bq8);

#define Q8D(B, U) hipLaunchKernelGGL((decomp_kernel<B, U, 8>), dim3((unsigned)((units + (int64_t)B * U - 1) / ((int64_t)B * U))), dim3(B), 0, 0, \
        (const void*)q8[(i + nsets / 2) % nsets], (const uint16_t*)scale, (u32x4*)w[i % nsets], units)
    rep("q8 dequant block 256, U=1 (product shape)", T.run([&](int i) { Q8D(256, 1); }, iters), bq8);
    rep("q8 dequant block 256, U=2", T.run([&](int i) { Q8D(256, 2); }, iters), bq8);
    rep("q8 dequant block 256, U=4", T.run([&](int i) { Q8D(256, 4); }, iters), bq8);
    rep("q8 dequant block 512, U=2", T.run([&](int i) { Q8D(512, 2); }, iters), bq8);
    return 0;
}
