// microbenchmark: cost of unaligned LDS stores on gfx950 (clock ticks per wave-instruction, 8 waves per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) uint8_t lds8;
template <int W, int STRIDE>
__global__ __launch_bounds__(512) void k(int off, int iters, unsigned long long* out, uint32_t* sink) {
    __shared__ __attribute__((aligned(16))) uint8_t s[8][4096 + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* p = s[wave] + lane * STRIDE + off;
    unsigned long long v = threadIdx.x * 0x0101010101010101ull;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint8_t* q = p + ((j & 1) ? 2048 : 0);
            if (W == 8) asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)(lds8*)q), "v"(v) : "memory");
            if (W == 4) asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)(lds8*)q), "v"((uint32_t)v) : "memory");
            if (W == 2) asm volatile("ds_write_b16 %0, %1" ::"v"((uint32_t)(uintptr_t)(lds8*)q), "v"((uint32_t)v) : "memory");
            if (W == 1) asm volatile("ds_write_b8 %0, %1" ::"v"((uint32_t)(uintptr_t)(lds8*)q), "v"((uint32_t)v) : "memory");
            v += 1;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = s[wave][lane * 4];
}
template <int W, int STRIDE>
void run(const char* name, int off) {
    unsigned long long* out; uint32_t* sink;
    hipMalloc(&out, 8); hipMalloc(&sink, 256 * 512 * 4);
    const int iters = 2000;
    k<W, STRIDE><<<256, 512>>>(off, iters, out, sink);
    k<W, STRIDE><<<256, 512>>>(off, iters, out, sink);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    // wall_clock64 = 100 MHz; 8 waves share the CU's LDS: ns per store instruction per CU = ticks * 10 / (iters * 8 stores * 8 waves)
    printf("%-4s stride %2d off %d: %.2f ns per wave store instruction on a CU with 8 waves storing\n", name, STRIDE, off, (double)c * 10.0 / (iters * 8.0 * 8.0));
    hipFree(out); hipFree(sink);
}
int main() {
    for (int off = 0; off < 4; ++off) { run<8, 8>("b64", off); }
    run<8, 8>("b64", 4); run<8, 8>("b64", 7);
    for (int off = 0; off < 4; ++off) { run<4, 8>("b32", off); }
    for (int off = 0; off < 2; ++off) { run<4, 4>("b32", off); }
    for (int off = 0; off < 2; ++off) { run<2, 8>("b16", off); run<2, 2>("b16", off); }
    run<1, 8>("b8", 0); run<1, 8>("b8", 1); run<1, 1>("b8", 0);
    // the shape of the compaction: ~8 bytes per lane at byte-granular addresses
    run<8, 7>("b64", 0); run<8, 5>("b64", 0); run<4, 5>("b32", 0); run<4, 3>("b32", 0); run<2, 3>("b16", 0); run<1, 3>("b8", 0);
    return 0;
}
