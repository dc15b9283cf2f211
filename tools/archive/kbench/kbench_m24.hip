// kbench_m24.hip — developer harness over the marlin-24 lean kernel (it includes ct_marlin24.hip, so this is the shipped arithmetic; no
// Python, a run takes seconds).  (1) PARITY of the lean kernel against the GENERAL fused kernel of the same file (marlin24_fused_w4_kernel:
// the exact per-element path, IEEE divides) on clean 2:4 tensors and on "dirty" ones (NaN / inf / out-of-range weights, scales outside the
// lean range, non-zero zero points, 2:4 violations), all four weight / scale dtype pairs, groups of 32 / 48 / 128 / k / 2 / channel-wise:
// packed words, metadata and the violation flag must be identical (round 5: the variant bits of the instruction-count work — 16-bit reads in
// the packing phase, msad flags, pair selectors, saddr addressing, NaN marker — were each compared byte for byte against the previous kernel
// this way, scale_packed included, before they replaced it: gpurun_out/r05m1/kbench_m24.txt, copied to profiles/r05_marlin_variants.txt);
// (2) HBM-cold time at 8192^2 bf16 g128 (6 rotating inputs, 5 blocks of 60 launches, median).  Not part of the product.
// Build: tools/kbench/build_m24.sh
#ifndef CT_M24_SRC  // build_m24.sh <source> <binary>: an experimental copy of the kernel source instead of the product's
#define CT_M24_SRC "../../compressed_tensors_amd/csrc/ct_marlin24.hip"
#endif
#include CT_M24_SRC

#include <string.h>
#include <algorithm>
#include <functional>
#include <vector>

using namespace ct;
#define CK(x) do { hipError_t ck_err_ = (x); if (ck_err_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(ck_err_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16; return x; }

// 2:4-structured bf16 weights: two of every four are zero (positions from a hash), the others |x| in [2^-7, 2^0) with random sign.
// dirty != 0: one element in ~2000 becomes NaN / +-inf / +-65536 / +-1e30 / a subnormal, and one quad in ~50000 gets a third non-zero.
__global__ void fill_w24(uint16_t* p, int64_t quads, uint32_t seed, int dirty, int dt) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)q + seed * 0x10001u);
        const int a = h & 3, b0 = (h >> 2) % 3;
        const int b = (a + 1 + b0) & 3;  // a != b
        uint16_t v[4] = {0, 0, 0, 0};
        for (int j = 0; j < 4; ++j) {
            if (j != a && j != b) continue;
            const uint32_t g = hash32(h + 77u * j + 1u);
            if (dt == CT_BF16) v[j] = (uint16_t)((g & 0x8000u) | (0x3c00u + ((g >> 16) & 0x3ffu)));       // exponents 0x78..0x7f
            else v[j] = (uint16_t)((g & 0x8000u) | (0x2000u + ((g >> 16) & 0x1fffu)));                    // fp16: 2^-7 .. 2^1
        }
        if (dirty) {
            const uint32_t d = hash32(h ^ 0xabcdef01u);
            if ((d & 0x7ff) == 0) {
                const int j = (d >> 11) & 3;
                static const uint16_t bf[8] = {0x7fc0, 0x7f80, 0xff80, 0x4780, 0xc780, 0x7149, 0x0001, 0x8040};
                static const uint16_t hf[8] = {0x7e00, 0x7c00, 0xfc00, 0x7bff, 0xfbff, 0x7a00, 0x0001, 0x8200};
                v[j] = dt == CT_BF16 ? bf[(d >> 13) & 7] : hf[(d >> 13) & 7];
            }
            if ((d >> 16) % 50000u == 1u) { for (int j = 0; j < 4; ++j) if (v[j] == 0) { v[j] = dt == CT_BF16 ? 0x3f00 : 0x3800; break; } }
        }
        for (int j = 0; j < 4; ++j) p[q * 4 + j] = v[j];
    }
}
// scales: ~ max|x| / 7 (about 0.14) with a random mantissa; dirty: some are 0, negative, tiny (2^-20), huge (2^17), NaN
__global__ void fill_scale(uint16_t* p, int64_t n, uint32_t seed, int dirty, int dt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)i * 3u + seed);
        uint16_t s = dt == CT_BF16 ? (uint16_t)(0x3e00u + (h & 0x7fu)) : (uint16_t)(0x3000u + (h & 0x3ffu));  // [0.125, 0.25)
        if (dirty && (h >> 20) % 97u == 0u) {
            static const uint16_t bf[8] = {0x0000, 0xbe10, 0x3580, 0x4800, 0x7fc0, 0x3980, 0x4700, 0x3e01};
            static const uint16_t hf[8] = {0x0000, 0xb100, 0x0010, 0x7bff, 0x7e00, 0x0c00, 0x7800, 0x3001};
            s = dt == CT_BF16 ? bf[(h >> 8) & 7] : hf[(h >> 8) & 7];
        }
        p[i] = s;
    }
}
__global__ void fill_zp(int8_t* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = hash32((uint32_t)i * 7u + seed);
        p[i] = (h % 13u == 0u) ? (int8_t)((h >> 8) % 15u - 7) : (int8_t)0;
    }
}

static double timed(const std::function<void(int)>& fn, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(0);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    for (int i = 0; i < 400; ++i) fn(i);
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

struct Out {
    int32_t* packed; uint16_t* meta; uint16_t* sp; int* bad;
    size_t n_packed, n_meta, n_sp;
    Out(int64_t m, int64_t k, int64_t groups) : n_packed((size_t)(m * k / 16)), n_meta((size_t)(m * k / 16)), n_sp((size_t)(groups * m)) {
        CK(hipMalloc(&packed, n_packed * 4)); CK(hipMalloc(&meta, n_meta * 2)); CK(hipMalloc(&sp, n_sp * 2)); CK(hipMalloc(&bad, 4));
    }
    void poison() { CK(hipMemset(packed, 0xa5, n_packed * 4)); CK(hipMemset(meta, 0xa5, n_meta * 2)); CK(hipMemset(sp, 0xa5, n_sp * 2)); CK(hipMemset(bad, 0, 4)); }
    void fetch(std::vector<uint8_t>& h) {
        h.resize(n_packed * 4 + n_meta * 2 + n_sp * 2 + 4);
        CK(hipMemcpy(h.data(), packed, n_packed * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h.data() + n_packed * 4, meta, n_meta * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h.data() + n_packed * 4 + n_meta * 2, sp, n_sp * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h.data() + n_packed * 4 + n_meta * 2 + n_sp * 2, bad, 4, hipMemcpyDeviceToHost));
    }
};

template <int XDT, int SDT>
static void launch(const uint16_t* w, const uint16_t* scale, const int8_t* zp, int64_t m, int64_t k, int64_t c, Out& o, int scale_single) {
    const unsigned tg = (unsigned)((m / 64) * (k / 256));
    const int xcd_rows = (m / 64) % 8 == 0 ? 1 : 0;
    const uint32_t tc_magic = k / 256 == 1 ? 0xffffffffu : (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(k / 256));
    hipLaunchKernelGGL((marlin24_fused_w4_lean_kernel<XDT, SDT>), dim3(tg), dim3(kBlock), 0, 0, w, scale, zp, m, k, c, k / c, o.packed, o.meta, o.bad, o.sp,
                       scale_single, xcd_rows, (unsigned int*)nullptr, (long long*)nullptr, tc_magic);
}
template <int XDT>
static void launch_general(const uint16_t* w, const uint16_t* scale, int sdt, const int8_t* zp, int64_t m, int64_t k, int64_t c, Out& o) {
    const unsigned tg = (unsigned)((m / 64) * (k / 256));
    hipLaunchKernelGGL((marlin24_fused_w4_kernel<XDT>), dim3(tg), dim3(kBlock), 0, 0, (const void*)w, (const void*)scale, sdt, (const void*)zp, (int)CT_I8, m, k, c, k / c,
                       o.packed, o.meta, o.bad);
}

template <int XDT, int SDT>
static bool parity_case(const char* what, int64_t m, int64_t k, int64_t c, int dirty, bool with_zp, uint32_t seed) {
    uint16_t* w; uint16_t* s; int8_t* z = nullptr;
    const int64_t groups = k / c;
    CK(hipMalloc(&w, m * k * 2)); CK(hipMalloc(&s, m * groups * 2));
    hipLaunchKernelGGL(fill_w24, dim3(4096), dim3(256), 0, 0, w, m * k / 4, seed, dirty, XDT);
    hipLaunchKernelGGL(fill_scale, dim3(256), dim3(256), 0, 0, s, m * groups, seed + 5u, dirty, SDT);
    if (with_zp) { CK(hipMalloc(&z, m * groups)); hipLaunchKernelGGL(fill_zp, dim3(256), dim3(256), 0, 0, z, m * groups, seed + 9u); }
    Out a(m, k, groups), b(m, k, groups);
    a.poison(); b.poison();
    launch_general<XDT>(w, s, SDT, z, m, k, c, a);
    launch<XDT, SDT>(w, s, z, m, k, c, b, groups * 2 < k / 2 ? 0 : 1);
    CK(hipMemcpy(a.sp, b.sp, a.n_sp * 2, hipMemcpyDeviceToDevice));  // the general kernel does not write scale_packed
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    std::vector<uint8_t> ha, hb;
    a.fetch(ha); b.fetch(hb);
    int bad_a, bad_b; memcpy(&bad_a, ha.data() + ha.size() - 4, 4); memcpy(&bad_b, hb.data() + hb.size() - 4, 4);
    ha.resize(ha.size() - 4); hb.resize(hb.size() - 4);
    const bool ok = ha == hb && (bad_a != 0) == (bad_b != 0);
    size_t first = 0;
    if (ha != hb) for (; first < ha.size() && ha[first] == hb[first]; ++first) {}
    printf("parity x=%d s=%d %-34s %lldx%lld g%lld dirty=%d zp=%d : %s (flags %d / %d)\n", XDT, SDT, what, (long long)m, (long long)k, (long long)c, dirty, (int)with_zp,
           ok ? "IDENTICAL" : "DIFFERENT", bad_a, bad_b);
    if (ha != hb) printf("   first difference at byte %zu of %zu (packed %zu bytes, meta %zu)\n", first, ha.size(), a.n_packed * 4, a.n_meta * 2);
    fflush(stdout);
    hipFree(w); hipFree(s); if (z) hipFree(z);
    hipFree(a.packed); hipFree(a.meta); hipFree(a.sp); hipFree(a.bad); hipFree(b.packed); hipFree(b.meta); hipFree(b.sp); hipFree(b.bad);
    return ok;
}

static bool parity_all() {
    bool ok = true;
    ok &= parity_case<CT_BF16, CT_BF16>("clean", 2048, 2048, 128, 0, false, 1);
    ok &= parity_case<CT_BF16, CT_BF16>("dirty", 2048, 2048, 128, 1, false, 2);
    ok &= parity_case<CT_BF16, CT_BF16>("dirty + zero points", 2048, 2048, 128, 1, true, 3);
    ok &= parity_case<CT_BF16, CT_F16>("dirty, fp16 scales", 1024, 2048, 128, 1, true, 4);
    ok &= parity_case<CT_F16, CT_BF16>("dirty, fp16 weights", 1024, 2048, 128, 1, true, 5);
    ok &= parity_case<CT_F16, CT_F16>("dirty, fp16 both", 1024, 2048, 128, 1, false, 6);
    ok &= parity_case<CT_BF16, CT_BF16>("group 32 (512 entries per tile)", 512, 1024, 32, 1, true, 7);
    ok &= parity_case<CT_BF16, CT_BF16>("group 48 (not a power of two)", 512, 768, 48, 1, false, 8);
    ok &= parity_case<CT_BF16, CT_BF16>("channel-wise", 512, 1024, 1024, 1, false, 9);
    ok &= parity_case<CT_BF16, CT_BF16>("group = k / 2", 448, 512, 256, 0, false, 10);
    ok &= parity_case<CT_BF16, CT_BF16>("rows not a multiple of 512", 192, 256, 128, 1, true, 11);
    ok &= parity_case<CT_BF16, CT_BF16>("full size", 8192, 8192, 128, 0, false, 12);
    return ok;
}

static void time_lean(const std::vector<uint16_t*>& ws, const uint16_t* scale, Out* outs, int64_t n) {
    const double bytes = 2.0 * n * n + n * n / 4.0 + n * n / 8.0 + 2.0 * 2.0 * n * (n / 128);
    double us = timed([&](int i) { launch<CT_BF16, CT_BF16>(ws[i % ws.size()], scale, nullptr, n, n, 128, outs[i & 1], 0); }, 60);
    printf("time   lean kernel %lldx%lld bf16 g128 : %7.2f us  %7.1f GB/s  %5.2f %% of 8 TB/s\n", (long long)n, (long long)n, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int64_t n = 8192;
    const bool ok = parity_all();
    printf("PARITY %s\n", ok ? "ALL IDENTICAL" : "FAILURES");
    std::vector<uint16_t*> ws;
    for (int i = 0; i < 6; ++i) {
        uint16_t* w; CK(hipMalloc(&w, n * n * 2));
        hipLaunchKernelGGL(fill_w24, dim3(4096), dim3(256), 0, 0, w, n * n / 4, 100u + i, 0, CT_BF16);
        ws.push_back(w);
    }
    uint16_t* scale; CK(hipMalloc(&scale, n * (n / 128) * 2));
    hipLaunchKernelGGL(fill_scale, dim3(256), dim3(256), 0, 0, scale, n * (n / 128), 55u, 0, CT_BF16);
    CK(hipDeviceSynchronize());
    Out o0(n, n, n / 128), o1(n, n, n / 128);
    Out outs[2] = {o0, o1};
    for (int rep = 0; rep < 3; ++rep) time_lean(ws, scale, outs, n);
    return ok ? 0 : 1;
}
