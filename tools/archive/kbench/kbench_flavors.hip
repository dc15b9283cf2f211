// kbench_flavors.hip — developer micro-benchmark: cache-policy flavours of the streaming stores /
// loads (plain, nt, sc1, sc0 sc1), size sweep of the copy ceiling (fixed overhead vs asymptote),
// XCD-contiguous block remap, software-pipelined persistent variants.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 kbench_flavors.hip -o kbench_flavors
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float bits_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef bf16_t b2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, b2));
}

// store flavours: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt
template <int F>
__device__ __forceinline__ void st16(u32x4* p, u32x4 v) {
    if constexpr (F == 0) *p = v;
    else if constexpr (F == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (F == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
// load flavours: 0 plain, 1 nt
template <int F>
__device__ __forceinline__ u32x4 ld16(const u32x4* p) {
    if constexpr (F == 0) return *p;
    else return __builtin_nontemporal_load(p);
}
template <int F>
__device__ __forceinline__ uint32_t ld4(const uint32_t* p) {
    if constexpr (F == 0) return *p;
    else return __builtin_nontemporal_load(p);
}

// remap blockIdx so that each XCD (b % 8) walks a contiguous eighth of the tensor
__device__ __forceinline__ int64_t xcd_block(bool remap) {
    if (!remap) return blockIdx.x;
    const int64_t per = gridDim.x >> 3;
    return (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

template <int U, int LF, int SF, bool REMAP>
__global__ __launch_bounds__(256) void copy16_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t n) {
    const int64_t base = xcd_block(REMAP) * 256 * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int i = 0; i < U; ++i) if (base + i * 256 < n) v[i] = ld16<LF>(in + base + i * 256);
#pragma unroll
    for (int i = 0; i < U; ++i) if (base + i * 256 < n) st16<SF>(out + base + i * 256, v[i]);
}

// decompress: lane = U units one block apart, 4 B loads, 16 B stores
template <int U, int LF, int SF, bool REMAP>
__global__ __launch_bounds__(256) void dq_kernel(const uint32_t* __restrict__ in, const uint16_t* __restrict__ scale, uint16_t* __restrict__ out, int64_t units) {
    const int64_t base = xcd_block(REMAP) * 256 * U + threadIdx.x;
    uint32_t w[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { int64_t u = base + (int64_t)i * 256; if (u < units) w[i] = ld4<LF>(in + u); }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        int64_t u = base + (int64_t)i * 256;
        if (u >= units) continue;
        float s = bits_f((uint32_t)scale[u >> 4] << 16);
        uint32_t ws[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = (float)((int)((w[i] >> (8 * j)) & 0xfu) - 8) * s;
            float b = (float)((int)((w[i] >> (8 * j + 4)) & 0xfu) - 8) * s;
            ws[j] = pk_bf16(a, b);
        }
        st16<SF>(reinterpret_cast<u32x4*>(out + u * 8), u32x4{ws[0], ws[1], ws[2], ws[3]});
    }
}

// decompress, persistent + software pipelined: grid = G blocks, each walks tiles of 256*U units with
// the next tile's loads issued before the current tile's math
template <int U, int SF>
__global__ __launch_bounds__(256) void dq_pipe_kernel(const uint32_t* __restrict__ in, const uint16_t* __restrict__ scale, uint16_t* __restrict__ out, int64_t units) {
    const int64_t tile = 256 * U;
    const int64_t ntiles = (units + tile - 1) / tile;
    int64_t t = blockIdx.x;
    if (t >= ntiles) return;
    uint32_t w[U], wn[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { int64_t u = t * tile + i * 256 + threadIdx.x; w[i] = u < units ? in[u] : 0; }
    for (; t < ntiles; t += gridDim.x) {
        const int64_t tn = t + gridDim.x;
        if (tn < ntiles) {
#pragma unroll
            for (int i = 0; i < U; ++i) { int64_t u = tn * tile + i * 256 + threadIdx.x; wn[i] = u < units ? in[u] : 0; }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            int64_t u = t * tile + (int64_t)i * 256 + threadIdx.x;
            if (u >= units) continue;
            float s = bits_f((uint32_t)scale[u >> 4] << 16);
            uint32_t ws[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = (float)((int)((w[i] >> (8 * j)) & 0xfu) - 8) * s;
                float b = (float)((int)((w[i] >> (8 * j + 4)) & 0xfu) - 8) * s;
                ws[j] = pk_bf16(a, b);
            }
            st16<SF>(reinterpret_cast<u32x4*>(out + u * 8), u32x4{ws[0], ws[1], ws[2], ws[3]});
        }
#pragma unroll
        for (int i = 0; i < U; ++i) w[i] = wn[i];
    }
}

__device__ __forceinline__ int cvt_i32_hw(float x) { int r; asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t q8_word_hw(const u32x4& raw, float rs) {
    const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t word = 0x88888888u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0 = bits_f(ws[j] << 16), x1 = bits_f(ws[j] & 0xffff0000u);
        uint32_t p = pk_bf16(x0 * rs, x1 * rs);
        float t0 = bits_f(p << 16), t1 = bits_f(p & 0xffff0000u);
        int c0 = cvt_i32_hw(__builtin_rintf(t0)), c1 = cvt_i32_hw(__builtin_rintf(t1));
        c0 = c0 < -8 ? -8 : (c0 > 7 ? 7 : c0);
        c1 = c1 < -8 ? -8 : (c1 > 7 ? 7 : c1);
        word += (uint32_t)c0 << (8 * j);
        word += (uint32_t)c1 << (8 * j + 4);
    }
    return word;
}
// compress: lane = 4 consecutive units (64 B in, 16 B out)
template <int LF, int SF, bool REMAP>
__global__ __launch_bounds__(256) void q_kernel(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale, u32x4* __restrict__ out, int64_t groups) {
    const int64_t g = xcd_block(REMAP) * 256 + threadIdx.x;
    if (g >= groups) return;
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = ld16<LF>(in + g * 4 + i);
    const float rs = 1.0f / bits_f((uint32_t)scale[g >> 2] << 16);
    st16<SF>(out + g, u32x4{q8_word_hw(r[0], rs), q8_word_hw(r[1], rs), q8_word_hw(r[2], rs), q8_word_hw(r[3], rs)});
}
// compress with 2 groups per lane one block apart (more bytes in flight per wave)
template <int LF, int SF>
__global__ __launch_bounds__(256) void q2_kernel(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale, u32x4* __restrict__ out, int64_t groups) {
    const int64_t g0 = (int64_t)blockIdx.x * 512 + threadIdx.x;
    u32x4 r[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) { int64_t g = g0 + k * 256; if (g < groups) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[k][i] = ld16<LF>(in + g * 4 + i); } }
#pragma unroll
    for (int k = 0; k < 2; ++k) { int64_t g = g0 + k * 256; if (g < groups) {
        const float rs = 1.0f / bits_f((uint32_t)scale[g >> 2] << 16);
        st16<SF>(out + g, u32x4{q8_word_hw(r[k][0], rs), q8_word_hw(r[k][1], rs), q8_word_hw(r[k][2], rs), q8_word_hw(r[k][3], rs)}); } }
}


// write-only: U x 16 B per lane, one block apart
template <int U, int SF, int BLOCK>
__global__ __launch_bounds__(BLOCK) void fill_kernel(u32x4* __restrict__ out, int64_t n, uint32_t seed) {
    const int64_t base = (int64_t)blockIdx.x * BLOCK * U + threadIdx.x;
#pragma unroll
    for (int i = 0; i < U; ++i) if (base + i * BLOCK < n) st16<SF>(out + base + i * BLOCK, u32x4{seed, seed + 1, (uint32_t)base, (uint32_t)i});
}
// read R bytes : write W bytes mixes, contiguous 16 B per lane both sides (traffic-shape ceilings)
template <int RU, int WU, int SF>
__global__ __launch_bounds__(256) void mix_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int64_t nblocks) {
    const int64_t b = blockIdx.x;
    u32x4 v[RU];
#pragma unroll
    for (int i = 0; i < RU; ++i) v[i] = in[(b * RU + i) * 256 + threadIdx.x];
    u32x4 x = v[0];
#pragma unroll
    for (int i = 1; i < RU; ++i) { x.x ^= v[i].x; x.y ^= v[i].y; x.z ^= v[i].z; x.w ^= v[i].w; }
#pragma unroll
    for (int i = 0; i < WU; ++i) { x.x += i; st16<SF>(out + (b * WU + i) * 256 + threadIdx.x, x); }
}

struct Bufs { void *w, *scale, *packed, *out; };

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 8192;
    const int64_t elems = N * N, units = elems / 8;
    const int NSETS = 12;
    std::vector<Bufs> sets(NSETS);
    std::vector<uint16_t> hw(elems), hs(elems / 128);
    srand(1);
    for (int64_t i = 0; i < elems; ++i) { float f = ((rand() & 0xffff) / 65536.0f - 0.5f) * 4.0f; uint32_t u; memcpy(&u, &f, 4); hw[i] = (uint16_t)(u >> 16); }
    for (int64_t i = 0; i < elems / 128; ++i) { float f = 0.25f + (rand() & 0xff) / 1024.0f; uint32_t u; memcpy(&u, &f, 4); hs[i] = (uint16_t)(u >> 16); }
    for (auto& b : sets) {
        CK(hipMalloc(&b.w, elems * 2)); CK(hipMalloc(&b.scale, elems / 128 * 2));
        CK(hipMalloc(&b.packed, elems / 2)); CK(hipMalloc(&b.out, elems * 2));
        CK(hipMemcpy(b.w, hw.data(), elems * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.scale, hs.data(), elems / 128 * 2, hipMemcpyHostToDevice));
        CK(hipMemset(b.packed, 0x5a, elems / 2));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alg = 2.0 * elems + 2.0 * elems / 128 + elems / 2.0;

    auto run = [&](const char* name, double bytes, std::function<void(const Bufs&)> fn) {
        for (int i = 0; i < 8; ++i) fn(sets[i % NSETS]);
        CK(hipDeviceSynchronize());
        float best = 1e30f, tot = 0;
        const int REP = 5, IT = 24;
        for (int r = 0; r < REP; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < IT; ++i) fn(sets[i % NSETS]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; tot += ms;
        }
        CK(hipGetLastError());
        double us = best * 1000.0 / IT, usavg = tot * 1000.0 / IT / REP;
        printf("%-40s  best %7.2f us  avg %7.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, us, usavg, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
    };
    auto G = [&](int64_t items, int per_block) { return dim3((unsigned)((items + per_block - 1) / per_block)); };

    printf("N=%lld  alg bytes/direction=%.0f\n", (long long)N, alg);

    {
        const int64_t n16 = elems * 2 / 16;
#define FILL(U, SF, BLK) run("fill 134MB U" #U " st" #SF " B" #BLK, 2.0 * elems, [&](const Bufs& b) { hipLaunchKernelGGL((fill_kernel<U, SF, BLK>), G(n16, BLK * U), dim3(BLK), 0, 0, (u32x4*)b.out, n16, 7u); })
        FILL(1, 0, 256); FILL(1, 1, 256); FILL(2, 1, 256); FILL(4, 1, 256); FILL(2, 2, 256); FILL(2, 4, 256); FILL(2, 1, 1024); FILL(4, 1, 1024); FILL(1, 1, 64);
        // mixes: bytes = (RU + WU) * 4 KiB per block
#define MIX(RU, WU, SF) { int64_t nb = (int64_t)(168.0e6 / ((RU + WU) * 4096)); if (nb * WU * 4096 > elems * 2) nb = elems * 2 / (WU * 4096); if (nb * RU * 4096 > elems * 2) nb = elems * 2 / (RU * 4096); \
        run("mix R" #RU ":W" #WU " st" #SF, (double)nb * (RU + WU) * 4096.0, [&](const Bufs& b) { hipLaunchKernelGGL((mix_kernel<RU, WU, SF>), dim3((unsigned)nb), dim3(256), 0, 0, (const u32x4*)b.w, (u32x4*)b.out, nb); }); }
        MIX(1, 4, 1); MIX(1, 4, 0); MIX(4, 1, 1); MIX(4, 1, 0); MIX(1, 1, 1); MIX(2, 2, 1); MIX(1, 2, 1); MIX(2, 1, 1);
    }

    // ---- copy ceilings by size (fixed overhead vs asymptote): bytes = 2 * n16 * 16
    for (int64_t mb : {32, 64, 84, 128, 256}) {
        int64_t n16 = mb * 1000000 / 16; if (n16 * 16 > elems * 2) n16 = elems * 2 / 16;
        char nm[64]; snprintf(nm, 64, "copy16 plain %lld MB -> %lld MB", (long long)(n16 * 16 / 1000000), (long long)(n16 * 16 / 1000000));
        run(nm, 2.0 * n16 * 16, [&](const Bufs& b) { hipLaunchKernelGGL((copy16_kernel<4, 0, 0, false>), G(n16, 1024), dim3(256), 0, 0, (const u32x4*)b.w, (u32x4*)b.out, n16); });
    }
    const int64_t n84 = (int64_t)(alg / 2 / 16);
#define COPY(LF, SF, RM) run("copy16 84MB ld" #LF " st" #SF " remap" #RM, alg, [&](const Bufs& b) { hipLaunchKernelGGL((copy16_kernel<4, LF, SF, RM>), G(n84, 1024), dim3(256), 0, 0, (const u32x4*)b.w, (u32x4*)b.out, n84); })
    COPY(0, 0, false); COPY(0, 1, false); COPY(0, 2, false); COPY(0, 3, false); COPY(0, 4, false); COPY(1, 0, false); COPY(1, 1, false); COPY(1, 2, false); COPY(1, 3, false);
    COPY(0, 0, true); COPY(1, 3, true);
#define DQ(U, LF, SF, RM) run("dq U" #U " ld" #LF " st" #SF " remap" #RM, alg, [&](const Bufs& b) { hipLaunchKernelGGL((dq_kernel<U, LF, SF, RM>), G(units, 256 * U), dim3(256), 0, 0, (const uint32_t*)b.packed, (const uint16_t*)b.scale, (uint16_t*)b.out, units); })
    DQ(2, 0, 0, false); DQ(2, 0, 1, false); DQ(2, 0, 2, false); DQ(2, 0, 3, false); DQ(2, 0, 4, false); DQ(2, 1, 0, false); DQ(2, 1, 1, false); DQ(2, 1, 2, false); DQ(2, 1, 3, false);
    DQ(2, 0, 0, true); DQ(2, 1, 3, true); DQ(4, 0, 3, false); DQ(4, 1, 3, false); DQ(1, 0, 3, false);
#define DQP(U, SF, GRID) run("dq_pipe U" #U " st" #SF " grid" #GRID, alg, [&](const Bufs& b) { hipLaunchKernelGGL((dq_pipe_kernel<U, SF>), dim3(GRID), dim3(256), 0, 0, (const uint32_t*)b.packed, (const uint16_t*)b.scale, (uint16_t*)b.out, units); })
    DQP(2, 0, 2048); DQP(2, 3, 2048); DQP(4, 0, 2048); DQP(4, 3, 2048); DQP(2, 0, 1024); DQP(4, 0, 1024); DQP(2, 0, 4096);
#define QK(LF, SF, RM) run("q Q4 ld" #LF " st" #SF " remap" #RM, alg, [&](const Bufs& b) { hipLaunchKernelGGL((q_kernel<LF, SF, RM>), G(units / 4, 256), dim3(256), 0, 0, (const u32x4*)b.w, (const uint16_t*)b.scale, (u32x4*)b.packed, units / 4); })
    QK(0, 0, false); QK(0, 1, false); QK(0, 2, false); QK(0, 3, false); QK(1, 0, false); QK(1, 1, false); QK(1, 3, false); QK(0, 0, true); QK(1, 3, true);
#define QK2(LF, SF) run("q2 2xQ4 ld" #LF " st" #SF, alg, [&](const Bufs& b) { hipLaunchKernelGGL((q2_kernel<LF, SF>), G(units / 4, 512), dim3(256), 0, 0, (const u32x4*)b.w, (const uint16_t*)b.scale, (u32x4*)b.packed, units / 4); })
    QK2(0, 0); QK2(1, 0); QK2(1, 3);
    return 0;
}
