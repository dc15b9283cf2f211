// kbench_w4d.hip — developer harness for the W4 decompress kernel's scale / zero-point fetch modes (it includes ct_quant.hip: the shipped
// arithmetic).  Per shape, symmetric and asymmetric: output of the per-lane mode (0), the DPP row-leader mode (1) and the scalar-load mode
// (2) compared byte for byte, then HBM-cold times (rotating inputs > 2 x the Infinity Cache, 5 blocks of 60 launches, median).
// Build: tools/kbench/build_w4d.sh.  Not part of the product.
#ifndef CT_QUANT_SRC
#define CT_QUANT_SRC "../../compressed_tensors_amd/csrc/ct_quant.hip"
#endif
#include CT_QUANT_SRC

#include <string.h>
#include <algorithm>
#include <functional>
#include <vector>

using namespace ct;
#define CK(x) do { hipError_t ck_err_ = (x); if (ck_err_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(ck_err_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16; return x; }
__global__ void fill_u32(uint32_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = hash32((uint32_t)i + seed);
}
__global__ void fill_scale(uint16_t* p, int64_t n, uint32_t seed, int dt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)i * 3u + seed);
        p[i] = dt == CT_BF16 ? (uint16_t)((h & 0x8000u) | (0x3000u + (h & 0xfffu))) : (uint16_t)((h & 0x8000u) | (0x1000u + (h & 0x3fffu)));
    }
}
__global__ void fill_zp(int8_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (int8_t)((int)(hash32((uint32_t)i * 7u + seed) % 16u) - 8);
}

static double timed(const std::function<void(int)>& fn, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(0);
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    for (int i = 0; i < 300; ++i) fn(i);
    CK(hipDeviceSynchronize());
    std::vector<double> per;
    for (int blk = 0; blk < 5; ++blk) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) fn(i);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        per.push_back(ms * 1000.0 / iters);
    }
    std::sort(per.begin(), per.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return per[2];
}

template <int DT, int U, bool ZP, int SM>
static void launch(const uint32_t* pk, const uint16_t* scale, const int8_t* zp, uint16_t* out, int64_t rows, int64_t cols) {
    W4Params w = make_w4(pk, scale, ZP ? zp : nullptr, ZP ? CT_I8 : -1, out, rows, cols, 1, 128, cols / 128);
    hipLaunchKernelGGL((w4_unpack_dequant_kernel<DT, U, ZP, SM>), dim3(w4_grid(w.units, U)), dim3(kBlock), 0, 0, w);
}

template <int DT, int U>
static bool shape(int64_t rows, int64_t cols, bool time_it) {
    const int64_t e = rows * cols;
    const int n = (int)std::max<int64_t>(4, (int64_t)(2 * 256 * 1048576LL) / (e / 2) + 1);  // the packed stream alone is 2 x the Infinity Cache
    const int nout = (int)std::max<int64_t>(2, (int64_t)(2 * 256 * 1048576LL) / (e * 2) + 1);
    std::vector<uint32_t*> pk; std::vector<uint16_t*> outs;
    uint16_t* scale; int8_t* zp;
    CK(hipMalloc(&scale, rows * (cols / 128) * 2)); CK(hipMalloc(&zp, rows * (cols / 128)));
    hipLaunchKernelGGL(fill_scale, dim3(256), dim3(256), 0, 0, scale, rows * (cols / 128), 5u, DT);
    hipLaunchKernelGGL(fill_zp, dim3(256), dim3(256), 0, 0, zp, rows * (cols / 128), 9u);
    for (int i = 0; i < n; ++i) { uint32_t* p; CK(hipMalloc(&p, e / 2)); hipLaunchKernelGGL(fill_u32, dim3(2048), dim3(256), 0, 0, p, e / 8, 77u * i + 1u); pk.push_back(p); }
    for (int i = 0; i < std::max(nout, 3); ++i) { uint16_t* o; CK(hipMalloc(&o, e * 2)); outs.push_back(o); }
    CK(hipDeviceSynchronize());
    bool ok = true;
    std::vector<uint16_t> h0(e), h1(e), h2(e);
    auto cmp = [&](const char* what) {
        CK(hipDeviceSynchronize()); CK(hipGetLastError());
        CK(hipMemcpy(h0.data(), outs[0], e * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), outs[1], e * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), outs[2], e * 2, hipMemcpyDeviceToHost));
        const bool a = h0 == h1, b = h0 == h2;
        printf("parity dt=%d U=%d %lldx%lld %-10s rowlead %s, scalar %s\n", DT, U, (long long)rows, (long long)cols, what, a ? "IDENTICAL" : "DIFFERENT", b ? "IDENTICAL" : "DIFFERENT");
        ok &= a && b;
    };
    for (int i = 0; i < 3; ++i) CK(hipMemset(outs[i], 0xa5, e * 2));
    launch<DT, U, true, 0>(pk[0], scale, zp, outs[0], rows, cols); launch<DT, U, true, 1>(pk[0], scale, zp, outs[1], rows, cols); launch<DT, U, true, 2>(pk[0], scale, zp, outs[2], rows, cols);
    cmp("asymmetric");
    for (int i = 0; i < 3; ++i) CK(hipMemset(outs[i], 0xa5, e * 2));
    launch<DT, U, false, 0>(pk[1], scale, zp, outs[0], rows, cols); launch<DT, U, false, 1>(pk[1], scale, zp, outs[1], rows, cols); launch<DT, U, false, 2>(pk[1], scale, zp, outs[2], rows, cols);
    cmp("symmetric");
    if (time_it) {
        const double bs = (2.0 + 0.5 + 2.0 / 128) * e, ba = bs + e / 128.0;
        const int it = e <= (1 << 24) ? 200 : 60;
        for (int rep = 0; rep < 2; ++rep) {
#define T(ZP, SM, name, bytes) { double us = timed([&](int i) { launch<DT, U, ZP, SM>(pk[i % n], scale, zp, outs[i % outs.size()], rows, cols); }, it); \
            printf("time dt=%d U=%d %lldx%lld %-22s %7.2f us  %7.1f GB/s  %5.2f %%\n", DT, U, (long long)rows, (long long)cols, name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0); fflush(stdout); }
            T(false, 0, "sym  per-lane (ships)", bs) T(false, 2, "sym  scalar", bs)
            T(true, 1, "asym row-leader (ships)", ba) T(true, 2, "asym scalar", ba) T(true, 0, "asym per-lane", ba)
#undef T
        }
    }
    for (auto p : pk) hipFree(p); for (auto o : outs) hipFree(o); hipFree(scale); hipFree(zp);
    return ok;
}

template <int U>
static void unroll_sweep(int64_t rows, int64_t cols) {  // scalar mode, symmetric and asymmetric, U units per lane
    const int64_t e = rows * cols;
    const int n = (int)std::max<int64_t>(4, (int64_t)(2 * 256 * 1048576LL) / (e / 2) + 1);
    std::vector<uint32_t*> pk; std::vector<uint16_t*> outs;
    uint16_t* scale; int8_t* zp;
    CK(hipMalloc(&scale, rows * (cols / 128) * 2)); CK(hipMalloc(&zp, rows * (cols / 128)));
    hipLaunchKernelGGL(fill_scale, dim3(256), dim3(256), 0, 0, scale, rows * (cols / 128), 5u, (int)CT_BF16);
    hipLaunchKernelGGL(fill_zp, dim3(256), dim3(256), 0, 0, zp, rows * (cols / 128), 9u);
    for (int i = 0; i < n; ++i) { uint32_t* p; CK(hipMalloc(&p, e / 2)); hipLaunchKernelGGL(fill_u32, dim3(2048), dim3(256), 0, 0, p, e / 8, 77u * i + 1u); pk.push_back(p); }
    for (int i = 0; i < 4; ++i) { uint16_t* o; CK(hipMalloc(&o, e * 2)); outs.push_back(o); }
    CK(hipDeviceSynchronize());
    const double bs = (2.0 + 0.5 + 2.0 / 128) * e;
    for (int rep = 0; rep < 2; ++rep) {
        double a = timed([&](int i) { launch<CT_BF16, U, false, 2>(pk[i % n], scale, zp, outs[i % 4], rows, cols); }, 60);
        double b = timed([&](int i) { launch<CT_BF16, U, true, 2>(pk[i % n], scale, zp, outs[i % 4], rows, cols); }, 60);
        printf("sweep U=%d %lldx%lld scalar: sym %7.2f us (%5.2f %%)  asym %7.2f us\n", U, (long long)rows, (long long)cols, a, bs / a / 1e3 / 80.0, b); fflush(stdout);
    }
    for (auto p : pk) hipFree(p); for (auto o : outs) hipFree(o); hipFree(scale); hipFree(zp);
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "sweep")) { unroll_sweep<1>(8192, 8192); unroll_sweep<2>(8192, 8192); unroll_sweep<4>(8192, 8192); unroll_sweep<8>(8192, 8192); return 0; }
    bool ok = true;
    ok &= shape<CT_BF16, 2>(2048, 2048, false);
    ok &= shape<CT_F16, 2>(1024, 4096, false);
    ok &= shape<CT_BF16, 4>(512, 1024, false);
    ok &= shape<CT_BF16, 2>(8192, 8192, true);
    ok &= shape<CT_BF16, 4>(4096, 4096, true);
    printf("PARITY %s\n", ok ? "ALL IDENTICAL" : "FAILURES");
    return ok ? 0 : 1;
}
