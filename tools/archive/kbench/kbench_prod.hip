// kbench_prod.hip — developer micro-benchmark over the PRODUCT's own kernel templates (it includes ct_quant.hip, so every variant
// below is the shipped arithmetic with a different launch shape): which (rows per workgroup, units per lane) is fastest for the
// activation-ordered W4 kernels, and which units-per-lane the W4 / int8 decompress side should use at which tensor size.
// HBM-cold rotation, HIP events, median of 5 blocks.  Not part of the product; nothing here is linked into libct_hip.so.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 <the flags of __graft_entry__.HIP_FLAGS> -I../../include -I../../compressed_tensors_amd/csrc kbench_prod.hip -o kbench_prod
#include "../../compressed_tensors_amd/csrc/ct_quant.hip"

#include <string.h>
#include <algorithm>
#include <functional>
#include <numeric>
#include <random>
#include <vector>

using namespace ct;
#define CK(x) do { hipError_t ck_err_ = (x); if (ck_err_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(ck_err_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void fill_bf16(uint16_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = ((uint32_t)i + seed) * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
        p[i] = (uint16_t)((h & 0x8000u) | (0x3D00u + ((h >> 16) & 0x3ffu)));  // |x| in [2^-5, 2^2)
    }
}
__global__ void fill_u32(uint32_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = ((uint32_t)i + seed) * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
        p[i] = h;
    }
}

static double timed(const std::function<void(int)>& fn, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(0);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    for (int i = 0; i < 600; ++i) fn(i);
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
static void rep(const char* name, double us, double bytes) { printf("%-58s %8.2f us  %7.1f GB/s  %5.1f %%\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0); fflush(stdout); }

struct Sets {
    int64_t rows, cols; int n;
    std::vector<uint16_t*> w, out; std::vector<uint32_t*> pk; std::vector<int8_t*> q8;
    uint16_t* scale; int8_t* zp;
    Sets(int64_t r, int64_t c, bool with_q8) : rows(r), cols(c) {
        const int64_t e = r * c;
        n = (int)std::max<int64_t>(6, (int64_t)(2 * 256 * 1048576LL) / (e / 2) + 1);
        CK(hipMalloc(&scale, r * (c / 128) * 2)); CK(hipMalloc(&zp, r * (c / 128)));
        CK(hipMemset(zp, 0, r * (c / 128)));
        std::vector<uint16_t> s(r * (c / 128), 0x3e80);  // 0.25 in bf16
        CK(hipMemcpy(scale, s.data(), s.size() * 2, hipMemcpyHostToDevice));
        for (int i = 0; i < n; ++i) {
            uint16_t* a; uint32_t* p; uint16_t* o;
            CK(hipMalloc(&a, e * 2)); hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, a, e, 1000u * i);
            CK(hipMalloc(&p, e / 2)); hipLaunchKernelGGL(fill_u32, dim3(4096), dim3(256), 0, 0, p, e / 8, 77u * i);
            CK(hipMalloc(&o, e * 2));
            w.push_back(a); pk.push_back(p); out.push_back(o);
            if (with_q8) { int8_t* q; CK(hipMalloc(&q, e)); hipLaunchKernelGGL(fill_u32, dim3(4096), dim3(256), 0, 0, (uint32_t*)q, e / 4, 31u * i); q8.push_back(q); }
        }
        CK(hipDeviceSynchronize());
        printf("# sets ready: %lldx%lld x %d\n", (long long)r, (long long)c, n); fflush(stdout);
    }
    ~Sets() { for (auto p : w) hipFree(p); for (auto p : pk) hipFree(p); for (auto p : out) hipFree(p); for (auto p : q8) hipFree(p); hipFree(scale); hipFree(zp); }
};

template <int R, int UL>
static void gidx_compress(Sets& S, const int32_t* cg, bool zp) {
    char name[128];
    const int64_t upr = S.cols / 8;
    if (upr % UL) return;
    const int chunks = (int)cdiv64(upr, (int64_t)UL * kBlock);
    dim3 g((unsigned)(cdiv64(S.rows, R) * chunks));
    printf("# g_idx compress R=%d UL=%d grid %u\n", R, UL, g.x); fflush(stdout);
    const double bytes = (2.0 + 0.5 + 2.0 / 128 + (zp ? 1.0 / 128 : 0)) * S.rows * S.cols;
    double us = timed([&](int i) {
        W4Params w = make_w4(S.w[i % S.n], S.scale, zp ? S.zp : nullptr, CT_I8, S.pk[i % S.n], S.rows, S.cols, 1, S.cols, S.cols / 128);
        if (zp) hipLaunchKernelGGL((w4_gidx_rows_kernel<CT_BF16, true, true, R, UL, kGidxSmallGroups>), g, dim3(kBlock), 0, 0, w, cg, chunks, S.rows);
        else hipLaunchKernelGGL((w4_gidx_rows_kernel<CT_BF16, false, true, R, UL, kGidxSmallGroups>), g, dim3(kBlock), 0, 0, w, cg, chunks, S.rows);
    }, 40);
    snprintf(name, sizeof name, "g_idx compress  %lldx%lld %s R=%d UL=%d", (long long)S.rows, (long long)S.cols, zp ? "asym" : "sym ", R, UL);
    rep(name, us, bytes);
}
template <int R, int UL>
static void gidx_decompress(Sets& S, const int32_t* cg, bool zp) {
    char name[128];
    const int64_t upr = S.cols / 8;
    const int chunks = (int)cdiv64(upr, (int64_t)UL * kBlock);
    dim3 g((unsigned)(cdiv64(S.rows, R) * chunks));
    const double bytes = (2.0 + 0.5 + 2.0 / 128 + (zp ? 1.0 / 128 : 0)) * S.rows * S.cols;
    double us = timed([&](int i) {
        W4Params w = make_w4(S.pk[(i + S.n / 2) % S.n], S.scale, zp ? S.zp : nullptr, CT_I8, S.out[i % S.n], S.rows, S.cols, 1, S.cols, S.cols / 128);
        if (zp) hipLaunchKernelGGL((w4_gidx_rows_kernel<CT_BF16, true, false, R, UL, kGidxSmallGroups>), g, dim3(kBlock), 0, 0, w, cg, chunks, S.rows);
        else hipLaunchKernelGGL((w4_gidx_rows_kernel<CT_BF16, false, false, R, UL, kGidxSmallGroups>), g, dim3(kBlock), 0, 0, w, cg, chunks, S.rows);
    }, 40);
    snprintf(name, sizeof name, "g_idx decompress %lldx%lld %s R=%d UL=%d", (long long)S.rows, (long long)S.cols, zp ? "asym" : "sym ", R, UL);
    rep(name, us, bytes);
}

template <int U>
static void w4_decompress(Sets& S) {
    char name[128];
    const double bytes = (2.0 + 0.5 + 2.0 / 128) * S.rows * S.cols;
    double us = timed([&](int i) {
        W4Params w = make_w4(S.pk[(i + S.n / 2) % S.n], S.scale, nullptr, -1, S.out[i % S.n], S.rows, S.cols, 1, 128, S.cols / 128);
        hipLaunchKernelGGL((w4_unpack_dequant_kernel<CT_BF16, U, false, false>), dim3(w4_grid(w.units, U)), dim3(kBlock), 0, 0, w);
    }, S.rows * S.cols <= (1 << 24) ? 200 : 60);
    snprintf(name, sizeof name, "W4 decompress %lldx%lld U=%d (%u workgroups)", (long long)S.rows, (long long)S.cols, U, w4_grid(S.rows * S.cols / 8, U));
    rep(name, us, bytes);
}
static void w4_compress(Sets& S) {
    char name[128];
    const double bytes = (2.0 + 0.5 + 2.0 / 128) * S.rows * S.cols;
    const int64_t groups = S.rows * S.cols / 32;
    double us = timed([&](int i) {
        hipLaunchKernelGGL((w4_quant_pack_lean_kernel<CT_BF16, true>), dim3((unsigned)cdiv64(groups, kBlock)), dim3(kBlock), 0, 0, (const u32x4*)S.w[i % S.n], S.scale,
                           S.zp, (u32x4*)S.pk[i % S.n], groups, 2);
    }, S.rows * S.cols <= (1 << 24) ? 200 : 60);
    snprintf(name, sizeof name, "W4 compress (lean) %lldx%lld", (long long)S.rows, (long long)S.cols);
    rep(name, us, bytes);
}
template <int U>
static void q8_dequant(Sets& S) {
    char name[128];
    const double bytes = 3.0 * S.rows * S.cols;
    double us = timed([&](int i) {
        W4Params w = make_w4(S.q8[(i + S.n / 2) % S.n], S.scale, nullptr, -1, S.out[i % S.n], S.rows, S.cols, S.rows, S.cols, 1);  // per-tensor scale
        hipLaunchKernelGGL((q8_dequant_kernel<CT_BF16, U, false, 0>), dim3(w4_grid(w.units, U)), dim3(kBlock), 0, 0, w);
    }, 200);
    snprintf(name, sizeof name, "int8 dequantize %lldx%lld per-tensor U=%d", (long long)S.rows, (long long)S.cols, U);
    rep(name, us, bytes);
}
static void q8_quant(Sets& S) {
    char name[128];
    const double bytes = 3.0 * S.rows * S.cols;
    double us = timed([&](int i) {
        W4Params w = make_w4(S.w[i % S.n], S.scale, nullptr, -1, S.q8[i % S.n], S.rows, S.cols, S.rows, S.cols, 1);
        hipLaunchKernelGGL((q8_quant_kernel<CT_BF16, false, true, 0>), dim3(w4_grid(w.units / 2, 1)), dim3(kBlock), 0, 0, w, -128, 127);
    }, 200);
    snprintf(name, sizeof name, "int8 quantize %lldx%lld per-tensor (product shape)", (long long)S.rows, (long long)S.cols);
    rep(name, us, bytes);
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    const bool all = !strcmp(what, "all");
    if (all || !strcmp(what, "gidx")) {
        Sets S(8192, 8192, false);
        std::vector<int32_t> cg(8192);
        std::iota(cg.begin(), cg.end(), 0);
        std::shuffle(cg.begin(), cg.end(), std::mt19937(7));
        for (auto& v : cg) v /= 128;
        int32_t* d_cg; CK(hipMalloc(&d_cg, 8192 * 4)); CK(hipMemcpy(d_cg, cg.data(), 8192 * 4, hipMemcpyHostToDevice));
        gidx_compress<4, 2>(S, d_cg, false); gidx_compress<8, 2>(S, d_cg, false); gidx_compress<2, 2>(S, d_cg, false); gidx_compress<4, 4>(S, d_cg, false);
        gidx_compress<8, 1>(S, d_cg, false); gidx_compress<4, 1>(S, d_cg, false);
        gidx_compress<4, 2>(S, d_cg, true); gidx_compress<8, 2>(S, d_cg, true); gidx_compress<4, 4>(S, d_cg, true);
        gidx_decompress<4, 1>(S, d_cg, false); gidx_decompress<8, 1>(S, d_cg, false); gidx_decompress<8, 2>(S, d_cg, false); gidx_decompress<4, 2>(S, d_cg, false);
        gidx_decompress<4, 1>(S, d_cg, true); gidx_decompress<8, 1>(S, d_cg, true); gidx_decompress<4, 2>(S, d_cg, true);
    }
    if (all || !strcmp(what, "small")) {
        for (auto sh : {std::pair<int64_t, int64_t>{4096, 4096}, {2048, 5632}, {8192, 4096}, {8192, 8192}}) {
            Sets S(sh.first, sh.second, false);
            w4_compress(S);
            w4_decompress<1>(S); w4_decompress<2>(S); w4_decompress<4>(S); w4_decompress<8>(S);
        }
        {
            Sets S(4096, 4096, true);
            q8_quant(S);
            q8_dequant<1>(S); q8_dequant<2>(S); q8_dequant<4>(S); q8_dequant<8>(S);
        }
        {
            Sets S(8192, 8192, true);
            q8_dequant<1>(S); q8_dequant<2>(S); q8_dequant<4>(S);
        }
    }
    return 0;
}
