// kbench_readshape.hip — developer micro-benchmark: does a 134 MB streaming read care whether a wave's 16-byte load instruction
// covers 1 KiB contiguously (lane-contiguous) or every other 16 bytes of 2 KiB (lane = 32 contiguous bytes, two instructions)?
// (the marlin-24 front end reads 32 bytes per lane; DESIGN 5.1 measured 64-byte-per-lane strided reads 19 % slower than contiguous)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// MODE 0: lane-pair shape (vector 2*l and 2*l+1 per row pass), MODE 1: lane-contiguous (vector l and l+256)
template <int MODE, int PASSES>
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ in, uint32_t* __restrict__ out, int64_t nvec) {
    const int64_t base = (int64_t)blockIdx.x * 256 * 2 * PASSES;
    u32x4 v[PASSES][2];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int64_t b = base + (int64_t)p * 512;
        if (MODE == 0) { v[p][0] = in[b + 2 * threadIdx.x]; v[p][1] = in[b + 2 * threadIdx.x + 1]; }
        else { v[p][0] = in[b + threadIdx.x]; v[p][1] = in[b + threadIdx.x + 256]; }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) acc += v[p][0].x ^ v[p][0].w ^ v[p][1].y ^ v[p][1].z;
    if (acc == 0x12345678u) out[threadIdx.x] = acc;  // never true for the test data; keeps the loads alive
}

int main() {
    const int64_t bytes = 134217728, nvec = bytes / 16;
    const int nsets = 6;
    std::vector<u32x4*> bufs(nsets);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x5a, bytes)); }
    uint32_t* out; CK(hipMalloc(&out, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 200; ++i) launch(i);
        CK(hipDeviceSynchronize());
        std::vector<double> per;
        for (int blk = 0; blk < 5; ++blk) {
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < 60; ++i) launch(i);
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); per.push_back(ms * 1000.0 / 60);
        }
        std::sort(per.begin(), per.end());
        printf("%-52s %7.2f us  %7.1f GB/s\n", name, per[2], bytes / per[2] / 1e3); fflush(stdout);
    };
#define L(MODE, P) [&](int i) { hipLaunchKernelGGL((rd<MODE, P>), dim3((unsigned)(nvec / (512 * P))), dim3(256), 0, 0, (const u32x4*)bufs[i % nsets], out, nvec); }
    run("32 B per lane (2 strided instr), 4 passes (marlin)", L(0, 4));
    run("lane-contiguous 16 B, 4 passes (8 loads in flight)", L(1, 4));
    run("32 B per lane, 2 passes", L(0, 2));
    run("lane-contiguous 16 B, 2 passes", L(1, 2));
    run("32 B per lane, 1 pass", L(0, 1));
    run("lane-contiguous 16 B, 1 pass", L(1, 1));
    return 0;
}
