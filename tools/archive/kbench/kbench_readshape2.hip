// kbench_readshape2.hip — developer micro-benchmark: the W4 compress / observer read shape.  A lane of those kernels reads 64
// CONTIGUOUS bytes (4 x 16 B: each wave load instruction touches every fourth 16-byte vector of a 4 KiB span); the alternative is
// lane-contiguous loads (each instruction 1 KiB contiguous) with the compress result transposed inside quads of lanes so that the
// store is still 16 bytes per lane (then in 4 segments of 256 B per wave instruction instead of 1 KiB contiguous).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t a, uint32_t b) { return (a * 0x9E3779B1u) ^ (b + 0x7F4A7C15u); }
__device__ __forceinline__ uint32_t red(const u32x4& v) { return mix(mix(v.x, v.y), mix(v.z, v.w)); }

// MODE 0: lane = 4 consecutive vectors (64 B), store out[g] (1 KiB contiguous per wave instruction)
// MODE 1: lane-contiguous loads (vector i*256 + tid of the workgroup's 1024), quad transpose with DPP, lane 4j+k stores the words of
//         vectors k*256 + 4j' .. of its quad: out index = (k * 256 + 4 * (tid >> 2)) / 4  -> 4 x 256 B segments per wave instruction
// MODE 2: lane-contiguous loads, NO store (read-only), MODE 3: 64 B per lane, NO store
template <int MODE>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint32_t* __restrict__ sink) {
    const int64_t base = (int64_t)blockIdx.x * 1024;  // vectors of this workgroup
    const int tid = threadIdx.x;
    u32x4 r[4];
    if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = in[base + 4 * tid + i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = in[base + i * 256 + tid];
    }
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = red(r[i]);
    if (MODE == 0) {
        __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, out + (base / 4) + tid);
    } else if (MODE == 1) {
        // 4 x 4 transpose inside each quad of lanes: lane q (= tid & 3) wants word i = q from its three neighbours
        uint32_t t[4];
        const int q = tid & 3;
#pragma unroll
        for (int s = 0; s < 4; ++s) {  // the value lane (q ^ s) holds for i = q ... done with two butterfly steps
            t[s] = w[s];
        }
        // butterfly: exchange with lane ^1 then lane ^2 (quad_perm DPP)
        uint32_t a0 = (q & 1) ? t[0] : t[1], a1 = (q & 1) ? t[2] : t[3];
        a0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)a0, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
        a1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)a1, 0xB1, 0xf, 0xf, false);
        if (q & 1) { t[0] = a0; t[2] = a1; } else { t[1] = a0; t[3] = a1; }
        uint32_t b0 = (q & 2) ? t[0] : t[2], b1 = (q & 2) ? t[1] : t[3];
        b0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)b0, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
        b1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)b1, 0x4E, 0xf, 0xf, false);
        if (q & 2) { t[0] = b0; t[1] = b1; } else { t[2] = b0; t[3] = b1; }
        // lane q of quad j now holds the four words of vectors q * 256 + 4 j .. + 3
        __builtin_nontemporal_store(u32x4{t[0], t[1], t[2], t[3]}, out + (base + q * 256 + 4 * (tid >> 2)) / 4);
    } else {
        const uint32_t acc = w[0] ^ w[1] ^ w[2] ^ w[3];
        if (acc == 0x12345678u) sink[tid] = acc;
    }
}

int main() {
    const int64_t bytes = 134217728, nvec = bytes / 16;
    const int nsets = 6;
    std::vector<u32x4*> in(nsets), out(nsets);
    for (int i = 0; i < nsets; ++i) { CK(hipMalloc(&in[i], bytes)); CK(hipMemset(in[i], 0x5a + i, bytes)); CK(hipMalloc(&out[i], bytes / 4)); }
    uint32_t* sink; CK(hipMalloc(&sink, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch, double total) {
        for (int i = 0; i < 300; ++i) launch(i);
        CK(hipDeviceSynchronize());
        std::vector<double> per;
        for (int blk = 0; blk < 5; ++blk) {
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < 60; ++i) launch(i);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); per.push_back(ms * 1000.0 / 60);
        }
        std::sort(per.begin(), per.end());
        printf("%-72s %7.2f us  %7.1f GB/s\n", name, per[2], total / per[2] / 1e3); fflush(stdout);
    };
#define L(MODE) [&](int i) { hipLaunchKernelGGL((k<MODE>), dim3((unsigned)(nvec / 1024)), dim3(256), 0, 0, (const u32x4*)in[i % nsets], out[i % nsets], sink); }
    run("read-only, 64 B per lane (4 strided instr)", L(3), (double)bytes);
    run("read-only, lane-contiguous (1 KiB per instr)", L(2), (double)bytes);
    run("compress-shaped, 64 B per lane -> 16 B store, 1 KiB contiguous", L(0), bytes * 1.25);
    run("compress-shaped, lane-contiguous loads, quad transpose -> 16 B store", L(1), bytes * 1.25);
    return 0;
}
