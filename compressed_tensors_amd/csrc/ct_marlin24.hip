// ct_marlin24.hip — CUTLASS 2:4 semi-structured conversion and marlin-24 packing for gfx950.
//
// Reference: utils/semi_structured_conversions.py:33-298 (compress/decompress + metadata
// reordering), utils/permutations_24.py:20-53 (marlin-24 permutation tables).  The
// Marlin24Compressor class itself is absent from the reference snapshot;
// the packing restated here follows its historical definition and is pinned against the CPU
// oracle, which in turn is pinned against the surviving reference primitives.
#include "ct_common.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace ct {

// destination linear offset of meta element (r, c) (semi_structured_conversions.py:33-60)
__host__ __device__ __forceinline__ int64_t meta_reorder_offset(int64_t r, int64_t c, int64_t m, int meta_itemsize) {
    const int64_t group_x = 64, group_y = meta_itemsize == 2 ? 32 : 16;
    int64_t dr = r / group_x * group_x + (r % 2) * 2 + (r % 8) / 4 + ((r % group_y) % 4) / 2 * 32 + ((r % group_x) / 8) * 4;
    int64_t dc = c;
    const int topright = (dr % 2 == 0) && (dc % 2 == 1);
    const int bottomleft = (dr % 2 == 1) && (dc % 2 == 0);
    dr += topright - bottomleft;
    dc -= topright - bottomleft;
    return (dc / 2) * m * 2 + dr * 2 + dc % 2;
}

// the 2:4 structure verdict: every violating lane stores the same 1 (idempotent, no read-modify-write), at system scope — the slot may
// live in pinned host memory (ct_mailbox_alloc: the default, check-per-call mode of Marlin24Compressor) where a device atomic
// would need PCIe atomics
// the ticket tree of the verdict mode (marlin24_fused_w4_lean_kernel): root at word 0, leaf i at word 32 * (1 + i) — one 128-byte line each.
// It lives in the CALLER's workspace (CT_M24_VERDICT_WORKSPACE_BYTES of include/ct_hip.h): the library holds no device state of its own.
constexpr unsigned kM24TreeWords = 32 * 65;
static_assert(kM24TreeWords * sizeof(unsigned int) == CT_M24_VERDICT_WORKSPACE_BYTES, "include/ct_hip.h states the workspace size");

__device__ __forceinline__ void raise_flag(int* bad) { __hip_atomic_store(bad, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// 4-bit code of a quad from its non-zero flags (:111-153)
__device__ __forceinline__ uint32_t quad_code(bool m0, bool m1, bool m3) {
    const bool e0 = m0 && m1, e1 = !m0 && m1, e2 = !m0 && !m1;
    const uint32_t bit0 = e1, bit1 = e2, bit2 = e0 || e2 || m3, bit3 = e1 || !m1;
    return bit0 | (bit1 << 1) | (bit2 << 2) | (bit3 << 3);
}

// one lane per metadata word: Q quads = 16 (16-bit inputs) or 32 (int8 inputs) dense elements
// = 32 bytes in, 16 bytes of kept values + one meta word out
template <int ES>
__global__ __launch_bounds__(kBlock) void cutlass24_from_dense_kernel(const void* __restrict__ dense, bool is_float, int64_t m, int64_t k,
                                                                      void* __restrict__ sparse, void* __restrict__ meta) {
    constexpr int MI = ES == 1 ? 4 : 2;  // meta itemsize
    constexpr int Q = MI * 2;            // quads per meta word
    const int64_t meta_ncols = k / (4 * Q);
    const int64_t total = m * meta_ncols;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / meta_ncols, mc = i - r * meta_ncols;
        const u32x4* in = reinterpret_cast<const u32x4*>(static_cast<const uint8_t*>(dense) + (r * k + mc * 4 * Q) * ES);
        const u32x4 a = in[0], b = in[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t word = 0;
        uint32_t outw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int qd = 0; qd < Q; ++qd) {
            uint32_t e[4];
            if constexpr (ES == 2) {
                e[0] = w[2 * qd] & 0xffffu; e[1] = w[2 * qd] >> 16; e[2] = w[2 * qd + 1] & 0xffffu; e[3] = w[2 * qd + 1] >> 16;
            } else {
                e[0] = w[qd] & 0xffu; e[1] = (w[qd] >> 8) & 0xffu; e[2] = (w[qd] >> 16) & 0xffu; e[3] = w[qd] >> 24;
            }
            const uint32_t zmask = (ES == 2 && is_float) ? 0x7fffu : 0xffffffffu;
            const uint32_t code = quad_code((e[0] & zmask) != 0, (e[1] & zmask) != 0, (e[3] & zmask) != 0);
            word |= code << (4 * qd);
            const uint32_t i0 = code & 3u, i1 = (code >> 2) & 3u;
            const uint32_t v0 = i0 == 0 ? e[0] : (i0 == 1 ? e[1] : (i0 == 2 ? e[2] : e[3]));
            const uint32_t v1 = i1 == 0 ? e[0] : (i1 == 1 ? e[1] : (i1 == 2 ? e[2] : e[3]));
            if constexpr (ES == 2) outw[qd] = v0 | (v1 << 16);
            else outw[qd >> 1] |= (v0 | (v1 << 8)) << (16 * (qd & 1));
        }
        *reinterpret_cast<u32x4*>(static_cast<uint8_t*>(sparse) + (r * (k / 2) + mc * 2 * Q) * ES) = u32x4{outw[0], outw[1], outw[2], outw[3]};
        const int64_t off = meta_reorder_offset(r, mc, m, MI);
        if constexpr (MI == 2) static_cast<uint16_t*>(meta)[off] = (uint16_t)word;
        else static_cast<uint32_t*>(meta)[off] = word;
    }
}

template <int ES>
__global__ __launch_bounds__(kBlock) void cutlass24_to_dense_kernel(const void* __restrict__ sparse, const void* __restrict__ meta, int64_t m,
                                                                    int64_t k /*sparse cols*/, void* __restrict__ dense) {
    constexpr int MI = ES == 1 ? 4 : 2;
    constexpr int Q = MI * 2;
    const int64_t meta_ncols = 2 * k / (4 * Q);
    const int64_t total = m * meta_ncols;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / meta_ncols, mc = i - r * meta_ncols;
        const int64_t off = meta_reorder_offset(r, mc, m, MI);
        const uint32_t word = MI == 2 ? (uint32_t)static_cast<const uint16_t*>(meta)[off] : static_cast<const uint32_t*>(meta)[off];
        const u32x4 s = *reinterpret_cast<const u32x4*>(static_cast<const uint8_t*>(sparse) + (r * k + mc * 2 * Q) * ES);
        const uint32_t sw[4] = {s.x, s.y, s.z, s.w};
        uint32_t ow[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int qd = 0; qd < Q; ++qd) {
            const uint32_t code = (word >> (4 * qd)) & 0xfu;
            const uint32_t i0 = code & 3u, i1 = (code >> 2) & 3u;
            uint32_t v0, v1;
            if constexpr (ES == 2) { v0 = sw[qd] & 0xffffu; v1 = sw[qd] >> 16; }
            else { const uint32_t h = (sw[qd >> 1] >> (16 * (qd & 1))) & 0xffffu; v0 = h & 0xffu; v1 = h >> 8; }
            // later writes win, exactly like the reference's scatter_ with duplicate indices
            // cannot happen here: i0 != i1 for every code the encoder emits
            if constexpr (ES == 2) {
                uint32_t lo = 0, hi = 0;  // elements 0,1 | 2,3 of the quad
                auto put = [&](uint32_t idx, uint32_t v) {
                    if (idx == 0) lo = (lo & 0xffff0000u) | v; else if (idx == 1) lo = (lo & 0xffffu) | (v << 16);
                    else if (idx == 2) hi = (hi & 0xffff0000u) | v; else hi = (hi & 0xffffu) | (v << 16);
                };
                put(i0, v0); put(i1, v1);
                ow[2 * qd] = lo; ow[2 * qd + 1] = hi;
            } else {
                uint32_t q4 = 0;
                q4 = (q4 & ~(0xffu << (8 * i0))) | (v0 << (8 * i0));
                q4 = (q4 & ~(0xffu << (8 * i1))) | (v1 << (8 * i1));
                ow[qd] = q4;
            }
        }
        u32x4* o = reinterpret_cast<u32x4*>(static_cast<uint8_t*>(dense) + (r * 2 * k + mc * 4 * Q) * ES);
        o[0] = u32x4{ow[0], ow[1], ow[2], ow[3]};
        o[1] = u32x4{ow[4], ow[5], ow[6], ow[7]};
    }
}



// fp16(x / s) without the 11-instruction IEEE divide: rs = fl(1 / s) (one correctly rounded reciprocal
// per 16 elements), q0 = x * rs, one Newton correction q1 = fma(fma(-q0, s, x), rs, q0).  q1 is within
// half an fp32 ulp (+ a vanishing term) of the exact quotient; the quotient of two 11-bit significands is
// never closer than 2^-23 relative to an fp16 rounding boundary and never on one, so the rounding to
// fp16 is the same for every normal result.  Checked exhaustively on the device over all 65536 x 65536
// fp16 pairs by ct_selftest_f16_div (tests/test_gpu_parity.py): identical values everywhere except
// quotients below 2^-13 (fp16-subnormal results can differ in the last place; they all quantize to
// code 0).  Scales outside [2^-14, 2^15] (and 0, inf, NaN) keep the IEEE divide (rs = 0).
__device__ __forceinline__ float f16_fast_rcp(float s16) {
    const float as = __builtin_fabsf(s16);
    return ((as >= 0x1p-14f) && (as <= 0x1p15f)) ? 1.0f / s16 : 0.0f;
}
// round two floats to fp16 and back: one v_cvt_pk_f16_f32 + two v_cvt_f32_f16 (3 ops for 2 elements)
__device__ __forceinline__ void round2_f16(float& a, float& b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef f16_t h2 __attribute__((ext_vector_type(2)));
    const f2 r = __builtin_convertvector(__builtin_convertvector(f2{a, b}, h2), f2);
    a = r.x; b = r.y;
}

// unrounded quotient (the caller rounds pairs with round2_f16)
template <bool FAST>
__device__ __forceinline__ float f16_quotient_raw(float x16, float s16, float rs) {
    if constexpr (!FAST) return x16 / s16;
    const float q0 = x16 * rs;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, s16, x16), rs, q0);
    return __builtin_isfinite(q0) ? q1 : q0;  // x = +-inf: the correction would make NaN
}

template <bool FAST>
__device__ __forceinline__ float f16_quotient(float x16, float s16, float rs) {
    if constexpr (!FAST) return round_to<CT_F16>(x16 / s16);
    const float q0 = x16 * rs;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, s16, x16), rs, q0);
    return round_to<CT_F16>(__builtin_isfinite(q0) ? q1 : q0);  // x = +-inf: the correction would make NaN
}

__global__ __launch_bounds__(kBlock) void selftest_f16_div_kernel(uint32_t s_lo, uint32_t s_hi, unsigned long long* mismatches) {
    unsigned long long local = 0;
    for (uint32_t sb = s_lo + blockIdx.x; sb < s_hi; sb += gridDim.x) {
        const float s = f16_bits_to_f(sb);
        const float rs = f16_fast_rcp(s);
        if (rs == 0.0f) continue;  // outside the fast-path range
        for (uint32_t xb = threadIdx.x; xb < 65536u; xb += kBlock) {
            const float x = f16_bits_to_f(xb);
            const float a = f16_quotient<true>(x, s, rs);
            const float b = round_to<CT_F16>(x / s);
            const bool an = a != a, bn = b != b;
            // value comparison (-0 == +0: the sign of a zero never reaches an integer code); results that are
            // both below 2^-13 (fp16 subnormal quotients, where an exact fp32 quotient can sit on an fp16 tie
            // and one fp32 ulp flips it) round to code 0 with or without a zero point: not counted
            const bool tiny = __builtin_fabsf(a) < 0x1p-13f && __builtin_fabsf(b) < 0x1p-13f;
            const bool bad = (an != bn) || (!an && !bn && a != b && !tiny);
            local += bad ? 1ull : 0ull;
        }
    }
    if (local) atomicAdd(mismatches, local);
}

// ------------------------------------------------------------------------------------------
// marlin-24 front end, fused: weight (bf16 / fp16) + scale -> fp16 quantize -> 2:4 compress.
// Replaces `weight.to(fp16)`, `scale.to(fp16)`, quantize(...) kept in fp16, the 2:4 structure
// check and sparse_semi_structured_from_dense_cutlass of the restated pipeline (four full-size
// intermediates, ~900 us of framework kernels at 8192^2) by ONE pass: 2 B/element in, 0.5 B of
// kept int8 codes + 1/8 B of metadata out.  Arithmetic = the fp16 eager sequence of the
// reference's quantize (every op rounded to fp16): x16 = fp16(x); t = fp16(x16 / s16);
// [t = fp16(t + fp16(zp))]; clamp; rint.  A quad with more than two non-zero codes sets *bad.
// One lane per metadata word = 4 quads = 16 elements (32 B in, 8 B of codes + one int16 out).
// ------------------------------------------------------------------------------------------
// one metadata word: 16 elements (8 dwords) of a row -> 8 kept int8 codes + the 4 quad codes.
// FAST selects the reciprocal + Newton quotient; the choice is made once per word (one scale), so the
// IEEE divide sequence is not interleaved with every element
// v_cvt_i32_f32: saturating, NaN -> 0 (spelled as an instruction so that clang does not expand the conversion)
__device__ __forceinline__ int m24_cvt_i32(float x) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int XDT, bool FAST, bool HAS_ZP>
__device__ __forceinline__ bool marlin24_word(const uint32_t (&ws)[8], float s16, float z16, float rs, float qmin, float qmax, u32x2& codes,
                                              uint32_t& word) {
    // quad_code as a table over idx = m0 | m1 << 1 | m3 << 2 (4 bits per entry)
    constexpr uint32_t kQuadLut = []() constexpr {
        uint32_t lut = 0;
        for (int idx = 0; idx < 8; ++idx) {
            const bool m0 = idx & 1, m1 = (idx >> 1) & 1, m3 = (idx >> 2) & 1;
            const bool e0 = m0 && m1, e1 = !m0 && m1, e2 = !m0 && !m1;
            const uint32_t bit0 = e1, bit1 = e2, bit2 = e0 || e2 || m3, bit3 = e1 || !m1;
            lut |= (bit0 | (bit1 << 1) | (bit2 << 2) | (bit3 << 3)) << (4 * idx);
        }
        return lut;
    }();
    const int iqmin = (int)qmin, iqmax = (int)qmax;
    uint32_t lo = 0, hi = 0;
    word = 0;
    int worst = 0;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        uint32_t nz[4], packed4 = 0;
        float tq[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // element pairs: the fp16 roundings go through the packed conversion
            const uint32_t pair = ws[2 * qd + h];
            float x0, x1;
            if constexpr (XDT == CT_BF16) {
                x0 = bf16_bits_to_f(pair & 0xffffu); x1 = bits_f(pair & 0xffff0000u);
                round2_f16(x0, x1);  // weight.to(fp16)
            } else {
                x0 = f16_bits_to_f(pair & 0xffffu); x1 = f16_bits_to_f(pair >> 16);
            }
            float t0 = f16_quotient_raw<FAST>(x0, s16, rs), t1 = f16_quotient_raw<FAST>(x1, s16, rs);
            round2_f16(t0, t1);
            if (HAS_ZP) { t0 += z16; t1 += z16; round2_f16(t0, t1); }
            tq[2 * h] = t0; tq[2 * h + 1] = t1;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = tq[e];
            // rint then clamp == clamp then rint for integer bounds; NaN: non-zero (torch `!= 0`), code 0 (the int cast)
            const float tr = __builtin_rintf(t);
            nz[e] = (tr != 0.0f) ? 1u : 0u;
            int c = m24_cvt_i32(tr);
            c = c < iqmin ? iqmin : (c > iqmax ? iqmax : c);  // v_med3_i32
            // a NaN is clamped by torch as NaN and cast to 0, an out-of-range value to the bound: both reproduced
            packed4 |= ((uint32_t)c & 0xffu) << (8 * e);
        }
        const uint32_t cnt = nz[0] + nz[1] + nz[2] + nz[3];
        worst = (int)cnt > worst ? (int)cnt : worst;
        const uint32_t idx = nz[0] | (nz[1] << 1) | (nz[3] << 2);
        const uint32_t qc = (kQuadLut >> (4 * idx)) & 0xfu;
        word |= qc << (4 * qd);
        // the two kept codes: bytes (qc & 3) and (qc >> 2) of packed4
        const uint32_t sel = 0x0c0c0000u | (qc & 3u) | ((qc >> 2) << 8);
        const uint32_t two = __builtin_amdgcn_perm(0u, packed4, sel);
        if (qd < 2) lo |= two << (16 * qd); else hi |= two << (16 * (qd - 2));
    }
    codes = u32x2{lo, hi};
    return worst > 2;
}

// ---- packed-fp16 back end of the fast path ---------------------------------------------------------------------
// The exact form above spends ~20 VALU per element and the fused kernel is VALU-bound (60 us at 8192^2 against ~30 us
// of memory time; dropping 4 ops per element moved it to 52.6 us).  Same arithmetic, fewer instructions:
//   * the quotient stays the proven fp32 reciprocal + Newton step, but its fp16 rounding is the packed conversion
//     (v_cvt_pk_f16_f32) and everything after it works on fp16 PAIRS;
//   * clamp = v_pk_max_f16 / v_pk_min_f16; round-half-even + integer cast in ONE v_pk_add_f16: for t in [-128, 127],
//     fl16(t + 1536) = 1536 + rint(t) exactly (ulp(1536) = 1, ties to the even significand = the even integer), and
//     the low byte of each half of the sum IS the two's-complement code;
//   * the four codes of a quad are gathered with one v_perm_b32, their non-zero flags with a carry trick, the table
//     index and the count with two v_dot4_u32_u8.
// What the packed clamp cannot reproduce is NaN (maxnum drops it) and the inf / NaN guard of the Newton step: any
// non-finite first quotient in the 16 elements sends that lane to the exact form (rare, divergent).
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

template <int XDT, bool HAS_ZP>
__device__ __forceinline__ bool marlin24_word_packed(const uint32_t (&ws)[8], float s16, float z16, float rs, float qmin, float qmax, u32x2& codes,
                                                     uint32_t& word, bool& special) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr uint32_t kQuadLut = []() constexpr {
        uint32_t lut = 0;
        for (int idx = 0; idx < 8; ++idx) {
            const bool m0 = idx & 1, m1 = (idx >> 1) & 1, m3 = (idx >> 2) & 1;
            const bool e0 = m0 && m1, e1 = !m0 && m1, e2 = !m0 && !m1;
            const uint32_t bit0 = e1, bit1 = e2, bit2 = e0 || e2 || m3, bit3 = e1 || !m1;
            lut |= (bit0 | (bit1 << 1) | (bit2 << 2) | (bit3 << 3)) << (4 * idx);
        }
        return lut;
    }();
    const h2_t lo2 = {(_Float16)qmin, (_Float16)qmin}, hi2 = {(_Float16)qmax, (_Float16)qmax};
    const h2_t magic = {(_Float16)1536.0f, (_Float16)1536.0f};
    const h2_t z2 = {(_Float16)z16, (_Float16)z16};
    uint32_t lo = 0, hi = 0;
    word = 0;
    uint32_t worst = 0;
    special = false;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        uint32_t u[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t pair = ws[2 * qd + h];
            float x0, x1;
            if constexpr (XDT == CT_BF16) {
                x0 = bf16_bits_to_f(pair & 0xffffu); x1 = bits_f(pair & 0xffff0000u);
                round2_f16(x0, x1);  // weight.to(fp16)
            } else {
                x0 = f16_bits_to_f(pair & 0xffffu); x1 = f16_bits_to_f(pair >> 16);
            }
            const float a0 = x0 * rs, a1 = x1 * rs;
            special |= !__builtin_isfinite(a0) | !__builtin_isfinite(a1);
            const float t0 = __builtin_fmaf(__builtin_fmaf(-a0, s16, x0), rs, a0), t1 = __builtin_fmaf(__builtin_fmaf(-a1, s16, x1), rs, a1);
            h2_t t = __builtin_convertvector(f2{t0, t1}, h2_t);  // fp16(x16 / s16)
            if (HAS_ZP) t = t + z2;                              // fp16(t + zp)
            t = __builtin_elementwise_min(__builtin_elementwise_max(t, lo2), hi2);
            u[h] = __builtin_bit_cast(uint32_t, t + magic);
        }
        const uint32_t packed4 = __builtin_amdgcn_perm(u[1], u[0], 0x06040200u);  // codes of elements 0..3
        const uint32_t nzb = ((((packed4 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | packed4) & 0x80808080u) >> 7;  // one 0/1 byte per element
        const uint32_t cnt = __builtin_amdgcn_udot4(nzb, 0x01010101u, 0u, false);
        const uint32_t idx = __builtin_amdgcn_udot4(nzb, 0x04000201u, 0u, false);
        worst = cnt > worst ? cnt : worst;
        const uint32_t qc = (kQuadLut >> (4 * idx)) & 0xfu;
        word |= qc << (4 * qd);
        const uint32_t sel = 0x0c0c0000u | (qc & 3u) | ((qc >> 2) << 8);
        const uint32_t two = __builtin_amdgcn_perm(0u, packed4, sel);
        if (qd < 2) lo |= two << (16 * qd); else hi |= two << (16 * (qd - 2));
    }
    codes = u32x2{lo, hi};
    return worst > 2;
}

template <int XDT>
__device__ __forceinline__ bool marlin24_item(const void* __restrict__ w, const void* __restrict__ scale, int sdt, const void* __restrict__ zp, int zdt,
                                              int64_t r, int64_t mc, int64_t k, int64_t cdiv, int64_t scale_cols, float qmin, float qmax,
                                              u32x2& codes, uint32_t& word) {
    // group of the word's 16 columns: cdiv is a multiple of 16 (or the whole row), so this is mc / (cdiv / 16) — a shift for
    // the usual power-of-two groups, a 32-bit divide otherwise (the 64-bit software divide cost ~80 VALU per word)
    const uint32_t per = (uint32_t)(cdiv >> 4);
    const uint32_t grp = (per & (per - 1)) == 0 ? ((uint32_t)mc >> __builtin_ctz(per)) : ((uint32_t)mc / per);
    const int64_t si = r * scale_cols + grp;
    // the scale first: its reciprocal is computed while the 32 bytes of weights are in flight
    const float s16 = sdt == CT_F16 ? f16_bits_to_f(static_cast<const uint16_t*>(scale)[si]) : round_to<CT_F16>(load_rt(scale, sdt, si));
    const bool has_zp = zp != nullptr;
    const float z16 = !has_zp ? 0.0f : (zdt == CT_I8 ? (float)static_cast<const int8_t*>(zp)[si] : round_to<CT_F16>(load_rt(zp, zdt, si)));
    const u32x4* in = reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(w) + r * k + mc * 16);
    const u32x4 a = in[0], b = in[1];
    const uint32_t ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const float rs = f16_fast_rcp(s16);
    // an all-zero zero point adds nothing (t is already rounded to fp16): wave-uniform skip, as in the W4 kernel
    const bool use_zp = has_zp && (__builtin_amdgcn_ballot_w64(z16 != 0.0f) != 0);
    if (rs != 0.0f) {
        bool special;
        const bool v = use_zp ? marlin24_word_packed<XDT, true>(ws, s16, z16, rs, qmin, qmax, codes, word, special)
                              : marlin24_word_packed<XDT, false>(ws, s16, z16, rs, qmin, qmax, codes, word, special);
        if (!special) return v;
        return use_zp ? marlin24_word<XDT, true, true>(ws, s16, z16, rs, qmin, qmax, codes, word)
                      : marlin24_word<XDT, true, false>(ws, s16, z16, rs, qmin, qmax, codes, word);
    }
    return use_zp ? marlin24_word<XDT, false, true>(ws, s16, z16, rs, qmin, qmax, codes, word)
                  : marlin24_word<XDT, false, false>(ws, s16, z16, rs, qmin, qmax, codes, word);
}

// generic: one lane per metadata word, the reordered int16 goes straight to its (scattered) place
template <int XDT>
__global__ __launch_bounds__(kBlock) void marlin24_quant_compress_kernel(const void* __restrict__ w, const void* __restrict__ scale, int sdt,
                                                                         const void* __restrict__ zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                                                                         int64_t scale_cols, float qmin, float qmax, int8_t* __restrict__ comp,
                                                                         uint16_t* __restrict__ meta, int* __restrict__ bad) {
    const int64_t meta_ncols = k / 16;
    const int64_t total = m * meta_ncols;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / meta_ncols, mc = i - r * meta_ncols;
        u32x2 codes;
        uint32_t word;
        const bool violation = marlin24_item<XDT>(w, scale, sdt, zp, zdt, r, mc, k, cdiv, scale_cols, qmin, qmax, codes, word);
        stream_store8(comp + r * (k / 2) + mc * 8, codes);
        meta[meta_reorder_offset(r, mc, m, 2)] = (uint16_t)word;
        if (violation) raise_flag(bad);
    }
}

// tiled (k % 256 == 0): a workgroup owns 64 rows x 16 metadata columns.  The reorder keeps the 64 rows of a
// column PAIR inside one contiguous 256-byte run of the output, but consecutive column pairs are m*4 bytes
// apart — written lane by lane (generic kernel) every 2-byte store lands in its own cache line (80 us).  Here
// the words are put in DESTINATION order in LDS and leave as 8 contiguous 256-byte runs.
template <int XDT>
__global__ __launch_bounds__(kBlock) void marlin24_quant_compress_tiled_kernel(const void* __restrict__ w, const void* __restrict__ scale, int sdt,
                                                                               const void* __restrict__ zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                                                                               int64_t scale_cols, float qmin, float qmax, int8_t* __restrict__ comp,
                                                                               uint16_t* __restrict__ meta, int* __restrict__ bad) {
    __shared__ __attribute__((aligned(16))) uint16_t s_meta[8][128];
    const int64_t tiles_c = k / 256;
    const int64_t tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x - tile_r * tiles_c;
    const int tid = threadIdx.x;
    bool violation = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = it * kBlock + tid;
        const int rl = item >> 4, cl = item & 15;
        const int64_t r = tile_r * 64 + rl, mc = tile_c * 16 + cl;
        u32x2 codes;
        uint32_t word;
        violation |= marlin24_item<XDT>(w, scale, sdt, zp, zdt, r, mc, k, cdiv, scale_cols, qmin, qmax, codes, word);
        stream_store8(comp + r * (k / 2) + mc * 8, codes);
        // meta_reorder_offset(r, mc, m, 2) relative to the pair's base, in 32-bit local terms: the row permutation only
        // involves the row inside its 64-row group, the column swap stays inside the column pair
        int dr = (rl & 1) * 2 + ((rl & 7) >> 2) + ((rl & 3) >> 1) * 32 + (rl >> 3) * 4;
        const int adj = (((dr & 1) == 0) && (cl & 1)) - (((dr & 1) == 1) && !(cl & 1));
        dr += adj;
        s_meta[cl >> 1][dr * 2 + ((cl - adj) & 1)] = (uint16_t)word;
    }
    if (violation) raise_flag(bad);
    __syncthreads();
    {
        const int pair = tid >> 5, chunk = tid & 31;  // 8 pairs x 32 chunks of 4 int16
        const int64_t pair_base = (tile_c * 8 + pair) * m * 2 + tile_r * 128;
        const u32x2 v = *reinterpret_cast<const u32x2*>(&s_meta[pair][chunk * 4]);
        stream_store8(meta + pair_base + chunk * 4, v);
    }
}

// entry `within` of the marlin-24 weight permutation (permutations_24.py:20-45), computed
// arithmetically instead of from a 1024-entry table
__host__ __device__ constexpr int marlin24_perm_entry(int within, int bits) {
    const int ilen = bits == 4 ? 8 : 4;
    const int grp = within / ilen, jj = within % ilen;
    const int src_in_grp = bits == 4 ? ((jj < 4) ? 2 * jj : 2 * (jj - 4) + 1) : ((jj == 0) ? 0 : (jj == 1 ? 2 : (jj == 2 ? 1 : 3)));
    const int idx = grp * ilen + src_in_grp;  // index into the un-interleaved list
    // list layout: for i in 0..31: for j in 0..3: for q in 0..7: perm1_i[q] + j
    const int i = idx / 32, rem = idx % 32, j = rem / 8, q = rem % 8;
    const int col = i / 4, col_o = col / 2, block = q / 4, t = q % 4;
    const int row = (t == 0) ? 2 * (i % 4) : (t == 1) ? 2 * (i % 4) + 1 : (t == 2) ? 2 * (i % 4 + 4) : 2 * (i % 4 + 4) + 1;
    return 16 * row + col_o * 256 + 8 * (col % 2) + 4 * block + j;
}


// byte offsets into the fused kernel's code tile (row stride 136) of the 8 source codes of the word at
// position jj inside a marlin chunk: the permutation evaluated at compile time (it cost 240 VALU ops per thread)
struct Marlin4SrcTable {
    uint16_t off[128][8];
};
constexpr Marlin4SrcTable make_marlin4_src_table() {
    Marlin4SrcTable t{};
    for (int jj = 0; jj < 128; ++jj)
        for (int e = 0; e < 8; ++e) {
            const int pe = marlin24_perm_entry(jj * 8 + e, 4);
            const int nt = pe >> 8, rem = pe & 255;
            t.off[jj][e] = (uint16_t)((nt * 16 + (rem & 15)) * (128 + 8) + (rem >> 4));
        }
    return t;
}
__device__ const Marlin4SrcTable kMarlin4Src = make_marlin4_src_table();

// fully fused int4 path (k % 256 == 0): the tiled front end above, with the kept codes held in LDS instead
// of HBM and the marlin-24 tile permutation + nibble packing done by the same workgroup.  A workgroup's
// 64 rows x 128 compressed columns are 8 k-tiles of exactly one 1024-element marlin chunk each (a chunk =
// 4 n-tiles of 16 rows x one k-tile of 16 columns), i.e. 8 runs of 128 consecutive output words.  Word
// w = tid + 256 * it has the same position jj = tid % 128 inside its chunk for every it, so a thread
// computes its 8 permutation entries once (arithmetic, permutations_24.py:20-45) and reuses them 4 times.
// No int8 intermediate (33.5 MB written + read back at 8192^2) and no separate packing launch.
template <int XDT>
__global__ __launch_bounds__(kBlock) void marlin24_fused_w4_kernel(const void* __restrict__ w, const void* __restrict__ scale, int sdt,
                                                                   const void* __restrict__ zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                                                                   int64_t scale_cols, int32_t* __restrict__ packed, uint16_t* __restrict__ meta,
                                                                   int* __restrict__ bad) {
    __shared__ __attribute__((aligned(16))) uint16_t s_meta[8][128];
    __shared__ __attribute__((aligned(16))) uint8_t s_code[64][128 + 8];  // +8: rows start on different banks
    const int64_t tiles_c = k / 256;
    const int64_t tile_r = blockIdx.x / tiles_c, tile_c = blockIdx.x - tile_r * tiles_c;
    const int tid = threadIdx.x;
    // this thread's 8 source positions inside a chunk (byte offsets into s_code for k-tile 0): one 16-byte load
    const u32x4 so = *reinterpret_cast<const u32x4*>(&kMarlin4Src.off[tid & 127][0]);
    const uint32_t src_off[8] = {so.x & 0xffffu, so.x >> 16, so.y & 0xffffu, so.y >> 16, so.z & 0xffffu, so.z >> 16, so.w & 0xffffu, so.w >> 16};
    bool violation = false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = it * kBlock + tid;
        const int rl = item >> 4, cl = item & 15;
        const int64_t r = tile_r * 64 + rl, mc = tile_c * 16 + cl;
        u32x2 codes;
        uint32_t word;
        violation |= marlin24_item<XDT>(w, scale, sdt, zp, zdt, r, mc, k, cdiv, scale_cols, -8.0f, 7.0f, codes, word);
        *reinterpret_cast<u32x2*>(&s_code[rl][cl * 8]) = codes;
        const int64_t off = meta_reorder_offset(r, mc, m, 2);
        const int pair = cl >> 1;
        const int64_t pair_base = (tile_c * 8 + pair) * m * 2 + tile_r * 128;
        s_meta[pair][(int)(off - pair_base)] = (uint16_t)word;
    }
    if (violation) raise_flag(bad);
    __syncthreads();
    {
        const int pair = tid >> 5, chunk = tid & 31;  // 8 pairs x 32 chunks of 4 int16
        const int64_t pair_base = (tile_c * 8 + pair) * m * 2 + tile_r * 128;
        stream_store8(meta + pair_base + chunk * 4, *reinterpret_cast<const u32x2*>(&s_meta[pair][chunk * 4]));
    }
    const uint8_t* sc = &s_code[0][0];
    const int64_t wpr = m * 2;  // packed words per k-tile row (size_n * 16 * 4 / 32)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int t = (tid >> 7) + 2 * it;  // k-tile inside the workgroup tile
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t code = (uint32_t)((int)(int8_t)sc[src_off[e] + t * 16] + 8);
            word |= code << (4 * e);
        }
        packed[(tile_c * 8 + t) * wpr + tile_r * 128 + (tid & 127)] = (int32_t)word;
    }
}

// ---- lean int4 front end (round 2) --------------------------------------------------------------------------------
// The packed-fp16 form above still spends ~21 VALU per element (1360 instructions per wave for 64 elements per lane) and the
// fused kernel was VALU-bound: 49 us against a ~34 us traffic floor.  This form needs ~9 per element.  What changed:
//   * no bf16 -> fp16 -> fp32 round trip and no per-element finiteness test: a bf16 weight with 2^-14 <= |x| <= 65280 IS its
//     own fp16 image, and with a scale >= 2^-12 anything smaller quantizes to code 0 either way (|x / s| <= 1/4).  The one
//     range / inf / NaN test for all 16 elements is a sum of squares kept by v_dot2c_f32_bf16 (one instruction per element
//     PAIR, straight from the raw dword): sum <= 65280^2 implies every |x| <= 65280, and inf / NaN propagate.  A lane that
//     fails the test (or has a scale outside [2^-12, 2^15], or a non-zero zero point) redoes its word in the exact form;
//   * bf16 weight AND bf16 scale: the quotient of two 8-bit significands is never within 2^-20 (relative) of an fp16
//     rounding boundary, so fp16(x * fl(1/s)) == fp16(fl(x/s)) without the Newton step (checked over all 3840 x 3457
//     in-range pairs on the CPU and by ct_selftest_m24_div on the device); fp16 weights or scales keep the proven
//     reciprocal + Newton form (ct_selftest_f16_div).  Multiplies and FMAs are the packed fp32 instructions;
//   * the magic-number rounding adds 1544 = 1536 + 8, so the low byte of each half is the UNSIGNED nibble c + 8 the marlin
//     word wants (the packing phase no longer sign-extends and re-biases), and a zero code is the byte 8;
//   * the 2:4 selection runs on all four quads of a metadata word together: per quad one v_dot4_u32_u8 of the non-zero
//     flags with the weights (9, 10, 8, 12) returns 16 * (idx + 8 * count) — table index and violation count in one
//     number —, the four 3-bit indices go through v_perm_b32 used as an 8-entry byte table (twice: the positions of the
//     first and the second kept element), and two more v_perm_b32 per quad pick the kept codes.
template <bool HI>
__host__ __device__ constexpr uint64_t quad_position_table() {  // byte idx = position (0..3) of the first / second kept element
    uint64_t t = 0;
    for (int idx = 0; idx < 8; ++idx) {
        const bool m0 = idx & 1, m1 = (idx >> 1) & 1, m3 = (idx >> 2) & 1;
        const bool e0 = m0 && m1, e1 = !m0 && m1, e2 = !m0 && !m1;
        const uint32_t bit0 = e1, bit1 = e2, bit2 = e0 || e2 || m3, bit3 = e1 || !m1;
        const uint64_t pos = HI ? (bit2 | (bit3 << 1)) : (bit0 | (bit1 << 1));
        t |= pos << (8 * idx);
    }
    return t;
}

__device__ __forceinline__ float m24_sumsq(uint32_t d, float acc, std::integral_constant<int, CT_BF16>) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, d), __builtin_bit_cast(b2, d), acc, false);
}
__device__ __forceinline__ float m24_sumsq(uint32_t d, float acc, std::integral_constant<int, CT_F16>) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, d), __builtin_bit_cast(h2_t, d), acc, false);
}

// 16 elements -> 8 kept codes as unsigned nibbles-in-bytes (c + 8), the 16-bit metadata word, 8 * count (+ the table index) folded
// into vmax, the sum of squares of the inputs.  No zero point.
template <int XDT, bool NEWTON>
__device__ __forceinline__ void marlin24_word_lean(const uint32_t (&ws)[8], float s16, float rs, u32x2& codes, uint32_t& word, uint32_t& vmax, float& sumsq) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 rs2 = {rs, rs}, s2 = {s16, s16};
    const h2_t lo2 = {(_Float16)-8.0f, (_Float16)-8.0f}, hi2 = {(_Float16)7.0f, (_Float16)7.0f};
    const h2_t magic = {(_Float16)1544.0f, (_Float16)1544.0f};
    uint32_t P[4], R[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t u[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t d = ws[2 * q + h];
            sumsq = m24_sumsq(d, sumsq, std::integral_constant<int, XDT>{});
            f2 x;
            if constexpr (XDT == CT_BF16) x = f2{bits_f(d << 16), bits_f(d & 0xffff0000u)};
            else x = __builtin_convertvector(__builtin_bit_cast(h2_t, d), f2);
            f2 t = x * rs2;
            if constexpr (NEWTON) t = __builtin_elementwise_fma(__builtin_elementwise_fma(-t, s2, x), rs2, t);
            h2_t t16 = __builtin_convertvector(t, h2_t);  // fp16(x16 / s16), RNE
            t16 = __builtin_elementwise_min(__builtin_elementwise_max(t16, lo2), hi2) + magic;
            u[h] = __builtin_bit_cast(uint32_t, t16);
        }
        P[q] = __builtin_amdgcn_perm(u[1], u[0], 0x06040200u);  // byte j = c_j + 8
        // v_msad_u8(a, ref, acc) = acc + the sum over the bytes with ref != 0 of |a - ref|: with ref = the code bytes (zero code = byte 0)
        // and a = ref + weight (no carries: 15 + 12 < 256) every NON-ZERO code adds exactly its weight — the non-zero flags and their weighted
        // sum in three instructions (xor, add, msad; round 4 took four: xor, add, and, v_dot4_u32_u8 — and its R came scaled by 16)
        const uint32_t z = P[q] ^ 0x08080808u;
        R[q] = __builtin_amdgcn_msad_u8(z + 0x0c080a09u, z, 0u);  // m0 + 2 m1 + 4 m3 + 8 * count
    }
    const uint32_t r01 = R[0] > R[1] ? R[0] : R[1], r23 = R[2] > R[3] ? R[2] : R[3];
    vmax = vmax > r01 ? vmax : r01;
    vmax = vmax > r23 ? vmax : r23;
    constexpr uint64_t kLo = quad_position_table<false>(), kHi = quad_position_table<true>();
    // R < 64: OR-ed together at byte distance, bits 0-2 of byte q are quad q's table index (three shift-ors and one mask; round 4 extracted
    // and placed the four indices one by one: eight instructions)
    const uint32_t idx4 = (R[0] | (R[1] << 8) | (R[2] << 16) | (R[3] << 24)) & 0x07070707u;
    const uint32_t lo4 = __builtin_amdgcn_perm((uint32_t)(kLo >> 32), (uint32_t)kLo, idx4);  // first kept position of each quad
    const uint32_t hi4 = __builtin_amdgcn_perm((uint32_t)(kHi >> 32), (uint32_t)kHi, idx4);  // second kept position
    // the kept codes of TWO quads in one v_perm over the register pair (P[q + 1], P[q]): selector bytes = (first, second) position of quad q,
    // the same of quad q + 1 with 4 added — three instructions per pair (round 4: a selector and a pick per quad plus a merge, five)
    const uint32_t s01 = __builtin_amdgcn_perm(hi4, lo4, 0x05010400u) | 0x04040000u;
    const uint32_t s23 = __builtin_amdgcn_perm(hi4, lo4, 0x07030602u) | 0x04040000u;
    codes = u32x2{__builtin_amdgcn_perm(P[1], P[0], s01), __builtin_amdgcn_perm(P[3], P[2], s23)};
    const uint32_t qc4 = (hi4 << 2) | lo4;  // one 4-bit quad code per byte
    const uint32_t w = qc4 | __builtin_amdgcn_alignbit(qc4, qc4, 4);
    word = __builtin_amdgcn_perm(0u, w, 0x0c0c0200u);
}

// floor(n / d) for a wave-uniform n < 2^32 from magic = floor(2^32 / d) (0xffffffff for d = 1): one multiply-high and one correction
// (floor(n * magic / 2^32) >= floor(n / d) - 1)
__device__ __forceinline__ uint32_t udiv_magic(uint32_t n, uint32_t d, uint32_t magic) {
    const uint32_t q = __umulhi(n, magic);
    return q + ((n - q * d) >= d ? 1u : 0u);
}

// the lean scale range: see above (2^-12 keeps sub-fp16-normal weights at code 0)
__device__ __forceinline__ float m24_lean_rcp_ieee(float s16) {  // the reference form: the correctly rounded quotient (kept for the selftest)
    const float as = __builtin_fabsf(s16);
    return ((as >= 0x1p-12f) && (as <= 0x1p15f)) ? 1.0f / s16 : 0.0f;
}
// Round 5: v_rcp_f32 + two Newton steps instead of the IEEE divide sequence (div_scale, rcp, four fma, div_fmas, div_fixup + range fix-ups:
// ~14 instructions -> 5).  In the lean range no scaling is needed, and for every fp16 scale the result is BIT-IDENTICAL to `1.0f / s16`:
// ct_selftest_m24_div(mode 2) compares the two over all 65536 patterns on the device (tests/test_gpu_parity.py), modes 0 / 1 still check
// every quotient that is built on it.
__device__ __forceinline__ float m24_lean_rcp(float s16) {
    const float as = __builtin_fabsf(s16);
    float r = __builtin_amdgcn_rcpf(s16);
    r = __builtin_fmaf(__builtin_fmaf(-s16, r, 1.0f), r, r);
    r = __builtin_fmaf(__builtin_fmaf(-s16, r, 1.0f), r, r);
    return ((as >= 0x1p-12f) && (as <= 0x1p15f)) ? r : 0.0f;
}
template <int XDT> struct m24_limit;  // largest sum of 16 squares that proves every element is an in-range finite value
template <> struct m24_limit<CT_BF16> { static constexpr float v = 65280.0f * 65280.0f; };
template <> struct m24_limit<CT_F16> { static constexpr float v = 16.0f * 65504.0f * 65504.0f; };

constexpr int64_t kM24LeanMaxDim = (int64_t)1 << 24;  // rows / columns the lean kernel's 32-bit lane offsets cover (see the loads)
template <int XDT, int SDT>
__global__ __launch_bounds__(kBlock) void marlin24_fused_w4_lean_kernel(const uint16_t* __restrict__ w, const uint16_t* __restrict__ scale,
                                                                        const int8_t* __restrict__ zp, int64_t m, int64_t k, int64_t cdiv,
                                                                        int64_t scale_cols, int32_t* __restrict__ packed, uint16_t* __restrict__ meta,
                                                                        int* __restrict__ bad, uint16_t* __restrict__ scale_packed, int scale_single, int xcd_rows,
                                                                        unsigned int* __restrict__ tickets, long long* __restrict__ verdict_word, uint32_t tc_magic) {
    constexpr bool NEWTON = !(XDT == CT_BF16 && SDT == CT_BF16);
    __shared__ __attribute__((aligned(16))) uint16_t s_meta[8][128];
    __shared__ __attribute__((aligned(16))) uint8_t s_code[64][128 + 8];  // +8: rows start on different banks
    __shared__ float s_rs[64][16];                 // reciprocal of the (row, group) scale, 0 = the words of that group take the exact path
    __shared__ float s_s16[NEWTON ? 64 : 1][16];   // the fp16 scale itself (Newton step of the fp16 forms)
    __shared__ uint16_t s_sbits[64][16];           // its fp16 bit pattern: what scale_packed stores (round 4: no second read of the scale matrix)
    const int tiles_c = (int)(k / 256);
    // Round 4: workgroup b runs on XCD b % 8 (observed placement; speed only).  With `xcd_rows` the eight XCDs take ADJACENT row blocks —
    // XCD x walks the tiles of row blocks x, x + 8, ... column by column — so that the 32 tiles sharing a row block's scale / zero-point
    // lines meet in ONE L2 while the chip as a whole still streams one contiguous 512-row band of the weight: PMC traffic 173.2 -> 162.2 MB
    // (1.072x -> 1.004x the algorithmic bytes: the scale and zero-point matrices are no longer fetched once per XCD), 30.4 -> 29.9 us.
    // (Each XCD on a contiguous EIGHTH of the tiles — eight far-apart streams — measured 36.5 us.)
    // Round 5: no runtime division on the way to the first load.  The quotients by `tiles_c` come from the host's floor(2^32 / tiles_c) and one
    // scalar multiply-high each (a wave-uniform `/` is a 12-instruction float-reciprocal sequence in the VECTOR unit, two of them quarter-rate
    // multiplies, and three of those sat in front of the address of the first weight load; the 64-bit `/ cdiv` pair of the scale_packed tail
    // was another 156 vector + 343 scalar instructions per wave: of the 716 vector instructions a wave executed, ~180 were index arithmetic).
    int tile_r, tile_c;
    if (xcd_rows) {
        const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        const uint32_t q = udiv_magic(i, (uint32_t)tiles_c, tc_magic);
        tile_r = (int)(q * 8u + x);
        tile_c = (int)(i - q * (uint32_t)tiles_c);
    } else {
        tile_r = (int)udiv_magic(blockIdx.x, (uint32_t)tiles_c, tc_magic);
        tile_c = (int)(blockIdx.x - (uint32_t)tile_r * (uint32_t)tiles_c);
    }
    const int tid = threadIdx.x;
    const uint32_t per = (uint32_t)(cdiv >> 4);
    const bool per_pow2 = (per & (per - 1)) == 0;
    const int per_shift = __builtin_ctz(per);
    const int cl = tid & 15, rl0 = tid >> 4;
    const uint32_t mc = (uint32_t)tile_c * 16u + (uint32_t)cl;
    uint32_t grp, g_first, g_last;  // this lane's group, the tile's first and last group
    if (per_pow2) {  // wave-uniform; kept a BRANCH (hipcc turns `pow2 ? shift : divide` into both + a select: the divide then always runs)
        grp = mc >> per_shift;
        g_first = ((uint32_t)tile_c * 16u) >> per_shift;
        g_last = ((uint32_t)tile_c * 16u + 15u) >> per_shift;
    } else {
        grp = mc / per;
        g_first = ((uint32_t)tile_c * 16u) / per;
        g_last = ((uint32_t)tile_c * 16u + 15u) / per;
    }
    // position of this lane's metadata word inside its column pair's 256-byte run (meta_reorder_offset in local terms): the
    // row permutation only involves the row inside its 64-row group, the column swap stays inside the column pair
    int dr0 = (rl0 & 1) * 2 + ((rl0 & 7) >> 2) + ((rl0 & 3) >> 1) * 32 + (rl0 >> 3) * 4;
    const int adj = (((dr0 & 1) == 0) && (cl & 1)) - (((dr0 & 1) == 1) && !(cl & 1));
    const int mpos0 = (dr0 + adj) * 2 + ((cl - adj) & 1);
    uint32_t vmax = 0;
    bool violation = false;
    // Round 3: the tile's scales — 64 rows x (at most 16, for group 128 two) groups — are converted ONCE, cooperatively: scale.to(fp16),
    // the lean-range test and the IEEE reciprocal used to be recomputed by every thread for each of its four words (1024 divisions per
    // tile for 128 distinct scales, plus eight small loads per thread).  A group whose zero point is not zero gets reciprocal 0, which
    // sends its words to the exact path like an out-of-range scale does.
    // Round 4: the scale (and zero point) load is issued BEFORE the weight loads and is unconditional straight-line code (entry index
    // clamped; a missing zero-point tensor reads a byte of the scale instead and ignores it).  Vector-memory results return in order:
    // issued after the 128 bytes of weights — as in round 3 — the scale could only be used behind `s_waitcnt vmcnt(0)`, i.e. the
    // "conversion while the weights are in flight" waited for every weight first and the whole load latency of a workgroup was exposed
    // (the ISA showed it).  Now the conversion (a divide) and the LDS hand-over run under the weight loads, and the four words are
    // consumed behind vmcnt(6 / 4 / 2 / 0) as they land.
    const int ng = (int)(g_last - g_first) + 1;  // <= 16
    const int n_entries = 64 * ng;
    const int e0 = tid < n_entries ? tid : n_entries - 1;
    int rl_e;
    if ((ng & (ng - 1)) == 0) rl_e = e0 >> __builtin_ctz((unsigned)ng);  // ng: 1, 2, 4, ... for power-of-two groups (wave-uniform branch)
    else rl_e = e0 / ng;
    const int gi_e = e0 - rl_e * ng;
    // (inline asm: hipcc sinks an ordinary small load to its first use, below the eight wide ones, and a volatile one is waited for
    // on the spot; these two are issued here and waited for by hand — `vmcnt(8)`: everything but the eight weight loads issued after them)
    uint32_t sb_e;
    int32_t zb_e;
    // Round 5 (second half): every address of the hot path is a wave-uniform 64-bit base (scalar registers) plus a 32-bit lane offset —
    // the `saddr` form of the global instructions — instead of 64-bit vector arithmetic per access (v_mad_u64_u32 + two v_mul_lo_u32 +
    // v_add3 + v_lshl_add_u64 in front of each of the four packed stores, the same in front of the loads).  The launcher takes this
    // kernel for m, k < 2^24 only (a tile's lane offsets then stay below 2^32).
    u32x4 wa[4], wb[4];
    {
        const int64_t s_uni = (int64_t)tile_r * 64 * scale_cols + g_first;                   // wave-uniform part of the entry's index
        const uint32_t s_lane = (uint32_t)rl_e * (uint32_t)scale_cols + (uint32_t)gi_e;       // < 64 * scale_cols
        const uint16_t* sbase = scale + s_uni;
        const int8_t* zbase = zp != nullptr ? zp + s_uni : reinterpret_cast<const int8_t*>(sbase);
        const uint32_t soff = s_lane * 2u, zoff = zp != nullptr ? s_lane : soff;
        asm volatile("global_load_ushort %0, %1, %2" : "=v"(sb_e) : "v"(soff), "s"(sbase) : "memory");
        asm volatile("global_load_sbyte %0, %1, %2" : "=v"(zb_e) : "v"(zoff), "s"(zbase) : "memory");
        // all 128 bytes of this lane's four words are requested before the first one is used
        const char* wbase = reinterpret_cast<const char*>(w + ((int64_t)tile_r * 64 * k + (int64_t)tile_c * 256));
        const uint32_t woff = ((uint32_t)rl0 * (uint32_t)k + (uint32_t)cl * 16u) * 2u;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const u32x4* in = reinterpret_cast<const u32x4*>(wbase + (size_t)(woff + (uint32_t)it * 32u * (uint32_t)k));
            wa[it] = in[0];
            wb[it] = in[1];
        }
    }
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(sb_e), "+v"(zb_e) : : "memory");
    {
        const float s16 = SDT == CT_F16 ? f16_bits_to_f(sb_e) : round_to<CT_F16>(bf16_bits_to_f(sb_e));  // scale.to(fp16)
        float rs = m24_lean_rcp(s16);
        if (zp != nullptr && zb_e != 0) rs = 0.0f;
        rs = rs == 0.0f ? __builtin_nanf("") : rs;  // the "exact path" marker in LDS is a NaN: it poisons the word's sum of squares by itself
        if (tid < n_entries) {
            s_rs[rl_e][gi_e] = rs;
            if (NEWTON) s_s16[rl_e][gi_e] = s16;
            s_sbits[rl_e][gi_e] = SDT == CT_BF16 ? (uint16_t)f_to_f16_bits(s16) : (uint16_t)sb_e;
        }
    }
    for (int e = tid + kBlock; e < n_entries; e += kBlock) {  // more than 256 (row, group) entries: groups narrower than 64 columns
        const int rl = e / ng, gi = e - rl * ng;
        const int64_t si = ((int64_t)tile_r * 64 + rl) * scale_cols + g_first + gi;
        const uint32_t sb = scale[si];
        const float s16 = SDT == CT_F16 ? f16_bits_to_f(sb) : round_to<CT_F16>(bf16_bits_to_f(sb));
        float rs = m24_lean_rcp(s16);
        if (zp != nullptr && zp[si] != 0) rs = 0.0f;
        rs = rs == 0.0f ? __builtin_nanf("") : rs;
        s_rs[rl][gi] = rs;
        if (NEWTON) s_s16[rl][gi] = s16;
        s_sbits[rl][gi] = SDT == CT_BF16 ? (uint16_t)f_to_f16_bits(s16) : (uint16_t)sb;
    }
    __syncthreads();
    const int gl = (int)(grp - g_first);
    const uint32_t g_first_tile = g_first;
    uint32_t redo = 0;  // words the range test rejected
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rl = it * 16 + rl0;  // (rl >> 3) advances by 2 per iteration -> the metadata row by 8
        const uint32_t ws[8] = {wa[it].x, wa[it].y, wa[it].z, wa[it].w, wb[it].x, wb[it].y, wb[it].z, wb[it].w};
        const float rs = s_rs[rl][gl];
        const float s16 = NEWTON ? s_s16[rl][gl] : 0.0f;  // only the Newton step reads the scale itself
        u32x2 codes;
        uint32_t word;
        // (a NaN reciprocal — scale outside the lean range, non-zero zero point — starts the sum as NaN, so the one range test below also
        // sends those words to the exact path: no separate compare of rs per word)
        float sumsq = rs * 0.0f;
        uint32_t vm = 0;
        marlin24_word_lean<XDT, NEWTON>(ws, s16, rs, codes, word, vm, sumsq);
        const bool special = !(sumsq <= m24_limit<XDT>::v);
        redo |= special ? (1u << it) : 0u;
        vmax = (!special && vm > vmax) ? vm : vmax;
        *reinterpret_cast<u32x2*>(&s_code[rl][cl * 8]) = codes;
        s_meta[cl >> 1][mpos0 + it * 16] = (uint16_t)word;
    }
    if (redo != 0) {  // rare and divergent: the exact form (IEEE divide, torch's NaN / inf / zero-point behaviour), one copy, not unrolled
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            if (!((redo >> it) & 1u)) continue;
            const int rl = it * 16 + rl0;
            u32x2 c8;
            uint32_t word;
            violation |= marlin24_item<XDT>(w, scale, SDT, zp, CT_I8, (int64_t)tile_r * 64 + rl, (int64_t)mc, k, cdiv, scale_cols, -8.0f, 7.0f, c8, word);
            *reinterpret_cast<u32x2*>(&s_code[rl][cl * 8]) = u32x2{(c8.x & 0x0f0f0f0fu) ^ 0x08080808u, (c8.y & 0x0f0f0f0fu) ^ 0x08080808u};  // int8 code c -> c + 8
            s_meta[cl >> 1][mpos0 + it * 16] = (uint16_t)word;
        }
    }
    const bool lane_bad = violation || vmax >= 24u;  // a quad with three or more non-zero codes
    // the permutation row of the packing phase is requested here — the weight registers are dead (and so is the rare exact path, which
    // needs the registers itself: requested above it, the row cost the fifth wave per SIMD) — so that its ~1 us (an L2 hit) passes
    // under the barrier and the metadata store instead of in front of the packing loop, where round 3 fetched it
    // (entries 0-3 only: entry e + 4 is the byte behind entry e, see the packing loop)
    const u32x2 so = *reinterpret_cast<const u32x2*>(&kMarlin4Src.off[tid & 127][0]);
    // Round 5, verdict mode (`tickets` != nullptr; ct_marlin24_compress_w4_verdict): the host wants "does the WHOLE tensor keep 2:4?" as early
    // as the device knows it, without waiting for the launch to drain (hipStreamSynchronize added ~15 us to the default-mode class call).
    // Every workgroup reports once, through a two-level ticket tree: leaf counter b % leaves (count in the low half, violating workgroups in
    // the high half of ONE 32-bit atomic), the last arrival of a leaf reports to the root, the last arrival at the root stores
    // 1 | (violated << 1) into the caller's (pinned, host-visible) word at system scope and leaves every counter it closed at zero for the
    // next launch.  A flat counter would take 2048 same-address atomics at ~40 ns each (they execute at the memory side on this multi-XCD
    // part: DESIGN.md 5.4) — longer than the kernel; <= 64 leaves of <= ~32 arrivals close in parallel.  The verdict travels IN the atomics'
    // values, so the arrivals need no fence.  The atomic is issued here and its result first read after the packing stores below.
    // Round 6: the tree is the caller's (`tickets` = the workspace argument of the entry), and the resets are ORDERED before the verdict:
    // a closer zeroes its counter with a returning exchange and reports upward only when that has come back, so by the time the host
    // can see the verdict every counter of the tree is zero at the memory side — the workspace may be handed to the next launch (on any
    // stream) as soon as the verdict has been read.
    unsigned int leaf_old = 0;
    int wg_bad = 0;
    if (tickets != nullptr) {  // workgroup-uniform
        wg_bad = __syncthreads_or(lane_bad ? 1 : 0);
        if (tid == 0) {
            const unsigned leaves = gridDim.x < 64u ? gridDim.x : 64u;
            leaf_old = __hip_atomic_fetch_add(tickets + 32u * (1u + blockIdx.x % leaves), 1u + (wg_bad ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (lane_bad) raise_flag(bad);
        __syncthreads();
    }
    {
        const int pair = tid >> 5, chunk = tid & 31;  // 8 pairs x 32 chunks of 4 int16
        char* mbase = reinterpret_cast<char*>(meta + ((int64_t)tile_c * 8 * m * 2 + (int64_t)tile_r * 128));  // wave-uniform
        const uint32_t moff = ((uint32_t)pair * (uint32_t)m * 2u + (uint32_t)chunk * 4u) * 2u;
        stream_store8(mbase + (size_t)moff, *reinterpret_cast<const u32x2*>(&s_meta[pair][chunk * 4]));
    }
    // the tree is closed HERE, ahead of the packing loop (the leaf atomic was issued above the metadata store; only wave 0 waits for it): the
    // verdict of the launch's last workgroup then travels while that workgroup packs, instead of starting its two round trips after it
    if (tickets != nullptr && tid == 0) {
        const unsigned leaves = gridDim.x < 64u ? gridDim.x : 64u, leaf = blockIdx.x % leaves;
        const unsigned leaf_size = gridDim.x / leaves + (leaf < gridDim.x % leaves ? 1u : 0u);
        if ((leaf_old & 0xffffu) + 1u == leaf_size) {  // the leaf's last arrival (leaf sizes stay below 2^16: the entry refuses launches of 64 x 65535 tiles and more)
            const unsigned leaf_bad = (leaf_old >> 16) + (wg_bad ? 1u : 0u);
            // (the exchange's result feeds the next atomic's operand — `& 0` with a zero the compiler cannot see — so the report upward is issued
            // only after the reset has been performed at the memory side, where every agent-scope atomic of this multi-XCD part executes)
            unsigned hidden_zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(hidden_zero));
            const unsigned z_leaf = __hip_atomic_exchange(tickets + 32u * (1u + leaf), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned root_old = __hip_atomic_fetch_add(tickets, 1u + (leaf_bad ? 0x10000u : 0u) + (z_leaf & hidden_zero),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((root_old & 0xffffu) + 1u == leaves) {
                const bool any_bad = (root_old >> 16) != 0u || leaf_bad != 0u;
                const unsigned z_root = __hip_atomic_exchange(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(verdict_word, (any_bad ? 3ll : 1ll) + (long long)(z_root & hidden_zero), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    const uint8_t* sc = &s_code[0][0];
    const uint32_t src_off[4] = {so.x & 0xffffu, so.x >> 16, so.y & 0xffffu, so.y >> 16};
    const int64_t wpr = m * 2;  // packed words per k-tile row (size_n * 16 * 4 / 32)
    // (a thread building FOUR consecutive words of one k-tile — four table rows, one 16-byte streaming store — measured slower:
    // 38.9 vs 36.2 us; the same word position in four k-tiles reuses one table row and its 4-byte stores are 512-byte runs.
    // Round 3: staging the words through 4 KB of LDS so that they leave as one 16-byte store per thread: 35.3 us, no change;
    // 5 instead of 4 waves per SIMD: 35.6 us, no change; the 32-byte-per-lane read shape against lane-contiguous 16-byte reads
    // (tools/kbench/kbench_readshape.hip): 21.6 against 21.9 us for the 134 MB, no difference.)
    // Round 5: the byte addresses are formed ONCE — table offset + this thread's k-tile half — and the four k-tiles of a thread are the
    // immediate offsets 0 / 32 / 64 / 96 of the reads.  Nibbles e and e + 4 of a word are the codes of compressed columns 2a and 2a + 1
    // of ONE row (permutations_24.py:26-33: the interleave [0, 2, 4, 6, 1, 3, 5, 7] over rows 2a, 2a + 1, 2a + 8, 2a + 9 of two 4-column
    // blocks), i.e. two adjacent bytes of the code tile at an even address: four 16-bit reads instead of eight byte reads,
    // v = n_e | n_(e+4) << 8.  v0 | v1 << 4 is the byte pair (n0 n1, n4 n5), v2 | v3 << 4 the pair (n2 n3, n6 n7), one v_perm interleaves
    // them: 3 vector + 4 LDS instructions per word for 8 + 8.
    uint32_t src_base[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) src_base[e] = src_off[e] + (uint32_t)(tid >> 7) * 16u;
    char* pbase = reinterpret_cast<char*>(packed + ((int64_t)tile_c * 8 * wpr + (int64_t)tile_r * 128));  // wave-uniform
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int t = (tid >> 7) + 2 * it;  // k-tile inside the workgroup tile
        uint32_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const uint16_t*>(sc + src_base[e] + (uint32_t)(it * 32));
        const uint32_t a = v[0] | (v[1] << 4), b = v[2] | (v[3] << 4);
        const uint32_t word = __builtin_amdgcn_perm(b, a, 0x05010400u);
        // streaming store: a plain one leaves the line dirty in the XCD's L2 and the kernel ends with a write-back bubble
        const uint32_t poff = ((uint32_t)t * (uint32_t)wpr + (uint32_t)(tid & 127)) * 4u;
        __builtin_nontemporal_store((int32_t)word, reinterpret_cast<int32_t*>(pbase + (size_t)poff));
    }
    // scale_packed (marlin24_pack_scales_kernel fused in): the 64 rows of this tile are one 64-entry row of the transposed
    // (groups, size_n) matrix per group, permuted inside itself — a contiguous 128-byte run.  The tile that holds a group's
    // first column writes it (groups of <= 256 columns: all of the tile's; wider groups / channel-wise: one tile in several).
    if (scale_packed != nullptr) {
        const uint32_t col0 = (uint32_t)tile_c * 256u, cd = (uint32_t)cdiv;  // cdiv <= k < 2^31 (the launcher clamps it)
        const int cshift = per_shift + 4;  // cdiv = 16 * per
        const int64_t g_first = per_pow2 ? ((col0 + cd - 1u) >> cshift) : ((col0 + cd - 1u) / cd);  // first group starting at or after col0
        const int n_here = (int)((per_pow2 ? ((col0 + 256u + cd - 1u) >> cshift) : ((col0 + 256u + cd - 1u) / cd)) - (uint32_t)g_first);  // groups starting inside [col0, col0 + 256)
        for (int e = tid; e < n_here * 64; e += kBlock) {
            const int gi = e >> 6, j = e & 63;
            const int pj = scale_single ? j : ((j & ~7) + (((j & 7) >> 1) | ((j & 1) << 2)));  // scale_perm: [0, 4, 1, 5, 2, 6, 3, 7] per 8
            const int64_t g = g_first + gi;
            // the tile's own LDS copy (a group that starts in this tile is one of the tile's groups): fp16(scale), converted once above
            scale_packed[g * m + (int64_t)tile_r * 64 + j] = s_sbits[pj][(int)(g - (int64_t)g_first_tile)];
        }
    }
}

// exhaustive check of the lean quotients against the IEEE divide: mode 0 = fp16 x, any fp16 scale in the lean range, reciprocal + Newton;
// mode 1 = bf16 x in the fp16-exact range, bf16 scale in the lean range, single multiply.  Same pass criterion as selftest_f16_div_kernel.
__global__ __launch_bounds__(kBlock) void selftest_m24_div_kernel(int mode, uint32_t s_lo, uint32_t s_hi, unsigned long long* mismatches) {
    unsigned long long local = 0;
    if (mode == 2) {  // the kernel's reciprocal against the IEEE divide, bit for bit, for every fp16 pattern in [s_lo, s_hi)
        for (uint32_t sb = s_lo + blockIdx.x * kBlock + threadIdx.x; sb < s_hi; sb += gridDim.x * kBlock) {
            const float s = f16_bits_to_f(sb);
            local += f_bits(m24_lean_rcp(s)) != f_bits(m24_lean_rcp_ieee(s)) ? 1ull : 0ull;
        }
        if (local) atomicAdd(mismatches, local);
        return;
    }
    for (uint32_t sb = s_lo + blockIdx.x; sb < s_hi; sb += gridDim.x) {
        const float s = mode == 0 ? f16_bits_to_f(sb) : round_to<CT_F16>(bf16_bits_to_f(sb));
        const float rs = m24_lean_rcp(s);
        if (rs == 0.0f || (mode == 1 && s != bf16_bits_to_f(sb))) continue;
        for (uint32_t xb = threadIdx.x; xb < 65536u; xb += kBlock) {
            const float x = mode == 0 ? f16_bits_to_f(xb) : bf16_bits_to_f(xb);
            if (!__builtin_isfinite(x) || (mode == 1 && __builtin_fabsf(x) > 65280.0f)) continue;  // the sum-of-squares test sends these to the exact form
            const float x16 = round_to<CT_F16>(x);  // what the reference divides
            float t = x * rs;
            if (mode == 0) t = __builtin_fmaf(__builtin_fmaf(-t, s, x), rs, t);
            const float a = round_to<CT_F16>(t), b = round_to<CT_F16>(x16 / s);
            // the codes must agree everywhere; the fp16 values wherever either is at least 2^-13 (smaller ones all round to code 0)
            const float ca = __builtin_rintf(__builtin_fminf(__builtin_fmaxf(a, -8.0f), 7.0f)), cb = __builtin_rintf(__builtin_fminf(__builtin_fmaxf(b, -8.0f), 7.0f));
            // (a bf16 weight below 2^-14 is not its own fp16 image: there only the code — always 0, |x / s| <= 1/4 — is compared)
            const bool tiny = (__builtin_fabsf(a) < 0x1p-13f && __builtin_fabsf(b) < 0x1p-13f) || (mode == 1 && __builtin_fabsf(x) < 0x1p-14f);
            local += (ca != cb || (a != b && !tiny)) ? 1ull : 0ull;
        }
    }
    if (local) atomicAdd(mismatches, local);
}

__device__ __forceinline__ int load_code(const void* q, int dt, int64_t i, int add) {
    switch (dt) {
        case CT_I32: return static_cast<const int32_t*>(q)[i] + add;
        case CT_I8: return (int)static_cast<const int8_t*>(q)[i] + add;
        case CT_F16: return (int)f16_bits_to_f(static_cast<const uint16_t*>(q)[i]) + add;
        case CT_BF16: return (int)bf16_bits_to_f(static_cast<const uint16_t*>(q)[i]) + add;
        case CT_F32: return (int)static_cast<const float*>(q)[i] + add;
    }
    return 0;
}

// one lane per packed word.  q holds the codes either as (size_k, size_n) [transposed == 0]
// or as the un-transposed compressed matrix (size_n, size_k) [transposed == 1]
__global__ __launch_bounds__(kBlock) void marlin24_pack_kernel(const void* __restrict__ q, int dt, int transposed, int add, int64_t size_k,
                                                               int64_t size_n, int bits, int32_t* __restrict__ packed) {
    const int pf = 32 / bits;
    const int64_t trow = size_n * 16;
    const int64_t wpr = trow / pf;
    const int64_t total = (size_k / 16) * wpr;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t kt = i / wpr, j = i - kt * wpr;
        uint32_t w = 0;
        for (int e = 0; e < pf; ++e) {
            const int64_t pos = j * pf + e;
            const int64_t chunk = pos >> 10;
            const int within = (int)(pos & 1023);
            const int64_t src = (chunk << 10) + marlin24_perm_entry(within, bits);
            const int64_t nt = src >> 8;
            const int rem = (int)(src & 255);
            const int64_t kk = kt * 16 + (rem >> 4), nn = nt * 16 + (rem & 15);
            const int64_t idx = transposed ? (nn * size_k + kk) : (kk * size_n + nn);
            w |= (uint32_t)load_code(q, dt, idx, add) << (bits * e);
        }
        packed[i] = (int32_t)w;
    }
}

// marlin-24 scale packing: scales (size_n, groups) -> transpose -> reshape(-1, 64)[:, perm]
// -> (groups, size_n).  perm: group table or identity ("single", channel-wise)
template <bool BF16_TO_F16>
__global__ __launch_bounds__(kBlock) void marlin24_pack_scales_kernel(const uint16_t* __restrict__ scale, int64_t size_n, int64_t groups, int single,
                                                                      uint16_t* __restrict__ out) {
    const int64_t total = size_n * groups;
    for (int64_t f = (int64_t)blockIdx.x * kBlock + threadIdx.x; f < total; f += (int64_t)gridDim.x * kBlock) {
        const int64_t i = f >> 6;
        const int j = (int)(f & 63);
        const int tbl[8] = {0, 4, 1, 5, 2, 6, 3, 7};
        const int pj = single ? j : (8 * (j >> 3) + tbl[j & 7]);
        const int64_t src = (i << 6) + pj;  // flat index into the transposed (groups, size_n) matrix
        const int64_t g = src / size_n, n = src - g * size_n;
        const uint16_t v = scale[n * groups + g];
        out[f] = BF16_TO_F16 ? (uint16_t)f_to_f16_bits(bf16_bits_to_f(v)) : v;  // scale.to(torch.float16), RNE
    }
}

static unsigned grid_1d(int64_t items) {
    int64_t g = cdiv64(items, kBlock);
    int64_t cap = (int64_t)kCUs * 32;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace ct

using namespace ct;

extern "C" {

int ct_cutlass24_from_dense(const void* dense, int dt, int64_t m, int64_t k, void* sparse, void* meta, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_F16 || dt == CT_BF16 || dt == CT_I8, "Invalid datatype code %d of dense matrix", dt);
    const int mi = dt == CT_I8 ? 4 : 2, Q = mi * 2;
    CT_REQUIRE(m >= 0 && k >= 0, "negative shape");
    CT_REQUIRE(m % 64 == 0, "Number of rows of dense matrix %lld must be divisible by 64", (long long)m);
    CT_REQUIRE(k % (4 * Q) == 0, "Number of columns of dense matrix %lld must be divisible by %d", (long long)k, 4 * Q);
    CT_REQUIRE(aligned16(dense) && aligned16(sparse), "dense/sparse buffers must be 16-byte aligned");
    if (m == 0 || k == 0) return CT_OK;
    const int64_t total = m * (k / (4 * Q));
    if (dt == CT_I8) hipLaunchKernelGGL((cutlass24_from_dense_kernel<1>), dim3(grid_1d(total)), dim3(kBlock), 0, as_stream(stream), dense, false, m, k, sparse, meta);
    else hipLaunchKernelGGL((cutlass24_from_dense_kernel<2>), dim3(grid_1d(total)), dim3(kBlock), 0, as_stream(stream), dense, true, m, k, sparse, meta);
    CT_LAUNCH_CHECK("ct_cutlass24_from_dense");
}

int ct_cutlass24_to_dense(const void* sparse, int dt, const void* meta, int meta_itemsize, int64_t m, int64_t k, void* dense, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_F16 || dt == CT_BF16 || dt == CT_I8, "Invalid datatype code %d of sparse matrix", dt);
    const int mi = dt == CT_I8 ? 4 : 2, Q = mi * 2;
    CT_REQUIRE(meta_itemsize == mi, "Invalid datatype of meta matrix (itemsize %d, expected %d)", meta_itemsize, mi);
    CT_REQUIRE(m >= 0 && k >= 0 && m % 64 == 0 && (2 * k) % (4 * Q) == 0, "bad sparse shape (%lld, %lld)", (long long)m, (long long)k);
    CT_REQUIRE(aligned16(dense) && aligned16(sparse), "dense/sparse buffers must be 16-byte aligned");
    if (m == 0 || k == 0) return CT_OK;
    const int64_t total = m * (2 * k / (4 * Q));
    if (dt == CT_I8) hipLaunchKernelGGL((cutlass24_to_dense_kernel<1>), dim3(grid_1d(total)), dim3(kBlock), 0, as_stream(stream), sparse, meta, m, k, dense);
    else hipLaunchKernelGGL((cutlass24_to_dense_kernel<2>), dim3(grid_1d(total)), dim3(kBlock), 0, as_stream(stream), sparse, meta, m, k, dense);
    CT_LAUNCH_CHECK("ct_cutlass24_to_dense");
}

int ct_marlin24_quant_compress(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                               int bits, int8_t* comp, int16_t* meta, int* bad, ct_stream_t stream) {
    CT_REQUIRE(wdt == CT_F16 || wdt == CT_BF16, "marlin-24 weights must be 16-bit floats, got dtype %d", wdt);
    CT_REQUIRE(is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(bits == 4 || bits == 8, "num_bits must be 4 or 8, got %d", bits);
    CT_REQUIRE(m >= 0 && k >= 0 && m % 64 == 0 && k % 16 == 0, "marlin-24 needs rows %% 64 == 0 and cols %% 16 == 0, got (%lld, %lld)", (long long)m,
               (long long)k);
    CT_REQUIRE(cdiv >= 16 && (cdiv % 16 == 0 || cdiv >= k), "group size %lld must be a multiple of 16", (long long)cdiv);
    CT_REQUIRE(aligned16(w) && (reinterpret_cast<uintptr_t>(comp) & 7u) == 0 && bad != nullptr, "misaligned buffers");
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), as_stream(stream));
    if (e != hipSuccess) return hip_check(e, "ct_marlin24_quant_compress memset");
    if (m == 0 || k == 0) return CT_OK;
    const int64_t c = cdiv > k ? k : cdiv;
    const int64_t scale_cols = k / c;
    const float qmax = (float)((1 << bits) / 2 - 1), qmin = -(float)((1 << bits) / 2);
    const int64_t total = m * (k / 16);
    const unsigned grid = (unsigned)(cdiv64(total, kBlock) < ((int64_t)1 << 30) ? cdiv64(total, kBlock) : ((int64_t)1 << 30));
    if (k % 256 == 0 && (reinterpret_cast<uintptr_t>(meta) & 7u) == 0 && (m / 64) * (k / 256) < ((int64_t)1 << 31)) {
        const unsigned tg = (unsigned)((m / 64) * (k / 256));
        if (wdt == CT_BF16)
            hipLaunchKernelGGL((marlin24_quant_compress_tiled_kernel<CT_BF16>), dim3(tg), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c,
                               scale_cols, qmin, qmax, comp, reinterpret_cast<uint16_t*>(meta), bad);
        else
            hipLaunchKernelGGL((marlin24_quant_compress_tiled_kernel<CT_F16>), dim3(tg), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c,
                               scale_cols, qmin, qmax, comp, reinterpret_cast<uint16_t*>(meta), bad);
        CT_LAUNCH_CHECK("ct_marlin24_quant_compress[tiled]");
    }
    if (wdt == CT_BF16)
        hipLaunchKernelGGL((marlin24_quant_compress_kernel<CT_BF16>), dim3(grid), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c, scale_cols,
                           qmin, qmax, comp, reinterpret_cast<uint16_t*>(meta), bad);
    else
        hipLaunchKernelGGL((marlin24_quant_compress_kernel<CT_F16>), dim3(grid), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c, scale_cols,
                           qmin, qmax, comp, reinterpret_cast<uint16_t*>(meta), bad);
    CT_LAUNCH_CHECK("ct_marlin24_quant_compress");
}

// returns CT_OK + 1 when the launch also wrote scale_packed
static int marlin24_compress_w4_impl(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                                     int32_t* packed, int16_t* meta, int* bad, bool clear_bad, void* scale_packed, int scale_single, bool* fused_scales,
                                     ct_stream_t stream, long long* verdict_word = nullptr, void* workspace = nullptr, bool clear_workspace = false) {
    if (fused_scales) *fused_scales = false;
    CT_REQUIRE(wdt == CT_F16 || wdt == CT_BF16, "marlin-24 weights must be 16-bit floats, got dtype %d", wdt);
    CT_REQUIRE(is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(m >= 0 && k >= 0 && m % 64 == 0 && k % 256 == 0, "the fused marlin-24 path needs rows %% 64 == 0 and cols %% 256 == 0, got (%lld, %lld)",
               (long long)m, (long long)k);
    CT_REQUIRE(cdiv >= 16 && (cdiv % 16 == 0 || cdiv >= k), "group size %lld must be a multiple of 16", (long long)cdiv);
    CT_REQUIRE(aligned16(w) && (reinterpret_cast<uintptr_t>(meta) & 7u) == 0 && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0 && bad != nullptr,
               "misaligned buffers");
    CT_REQUIRE((m / 64) * (k / 256) < ((int64_t)1 << 31), "tensor too large for one launch");
    unsigned int* tickets = nullptr;
    if (verdict_word != nullptr) {
        const bool lean_ok = (sdt == CT_F16 || sdt == CT_BF16) && (zp == nullptr || zdt == CT_I8) && m < kM24LeanMaxDim && k < kM24LeanMaxDim;
        if (!lean_ok || (m / 64) * (k / 256) >= (int64_t)64 * 65535 || m == 0 || k == 0) {
            CT_UNSUPPORTED("ct_marlin24_compress_w4_verdict: layout outside the one-launch kernel (16-bit scales, int8 or no zero point, < 4.2 M tiles, sides < 2^24)");
        }
        // the ticket tree is the caller's: all-zero when the launch starts (cleared here, in stream order, if the caller asks), all-zero again
        // when the verdict has been stored — nothing of it lives in the library, so any number of launches may be in flight, one per workspace
        tickets = static_cast<unsigned int*>(workspace);
        if (clear_workspace) {
            hipError_t e = hipMemsetAsync(workspace, 0, CT_M24_VERDICT_WORKSPACE_BYTES, as_stream(stream));
            if (e != hipSuccess) return hip_check(e, "ct_marlin24_compress_w4_verdict workspace memset");
        }
    }
    if (clear_bad) {
        hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), as_stream(stream));
        if (e != hipSuccess) return hip_check(e, "ct_marlin24_compress_w4 memset");
    }
    if (m == 0 || k == 0) return CT_OK;
    const int64_t c = cdiv > k ? k : cdiv;
    const unsigned tg = (unsigned)((m / 64) * (k / 256));
    // (the lean kernel forms its lane offsets in 32 bits: sides below 2^24; anything larger takes the general fused kernel)
    const bool lean = (sdt == CT_F16 || sdt == CT_BF16) && (zp == nullptr || zdt == CT_I8) && m < kM24LeanMaxDim && k < kM24LeanMaxDim;
    if (lean) {
#define CT_M24_LEAN(X, S)                                                                                                                      \
    hipLaunchKernelGGL((marlin24_fused_w4_lean_kernel<X, S>), dim3(tg), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(w), \
                       static_cast<const uint16_t*>(scale), static_cast<const int8_t*>(zp), m, k, c, k / c, packed, reinterpret_cast<uint16_t*>(meta), bad, \
                       static_cast<uint16_t*>(scale_packed), scale_single, xcd_rows, tickets, verdict_word, tc_magic)
        const int xcd_rows = (m / 64) % 8 == 0 ? 1 : 0;  // the row blocks divide evenly over the eight XCDs
        const uint32_t tc_magic = k / 256 == 1 ? 0xffffffffu : (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(k / 256));
        if (wdt == CT_BF16 && sdt == CT_BF16) CT_M24_LEAN(CT_BF16, CT_BF16);
        else if (wdt == CT_BF16) CT_M24_LEAN(CT_BF16, CT_F16);
        else if (sdt == CT_BF16) CT_M24_LEAN(CT_F16, CT_BF16);
        else CT_M24_LEAN(CT_F16, CT_F16);
#undef CT_M24_LEAN
        if (fused_scales) *fused_scales = scale_packed != nullptr;
    } else if (wdt == CT_BF16)
        hipLaunchKernelGGL((marlin24_fused_w4_kernel<CT_BF16>), dim3(tg), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c, k / c, packed,
                           reinterpret_cast<uint16_t*>(meta), bad);
    else
        hipLaunchKernelGGL((marlin24_fused_w4_kernel<CT_F16>), dim3(tg), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c, k / c, packed,
                           reinterpret_cast<uint16_t*>(meta), bad);
    CT_LAUNCH_CHECK("ct_marlin24_compress_w4");
}

int ct_marlin24_compress_w4(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                            int32_t* packed, int16_t* meta, int* bad, ct_stream_t stream) {
    return marlin24_compress_w4_impl(w, wdt, scale, sdt, zp, zdt, m, k, cdiv, packed, meta, bad, true, nullptr, 0, nullptr, stream);
}

int ct_marlin24_compress_w4_full(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                                 int group_perm, int32_t* packed, int16_t* meta, void* scale_packed, int* bad, int clear_bad, ct_stream_t stream) {
    CT_REQUIRE(sdt == CT_F16 || sdt == CT_BF16, "marlin-24 scales must be 16-bit floats, got dtype %d", sdt);
    CT_REQUIRE(scale_packed != nullptr, "scale_packed is NULL");
    bool fused = false;
    const int rc = marlin24_compress_w4_impl(w, wdt, scale, sdt, zp, zdt, m, k, cdiv, packed, meta, bad, clear_bad != 0, scale_packed, group_perm ? 0 : 1, &fused, stream);
    if (rc != CT_OK || m == 0 || k == 0 || fused) return rc;
    const int64_t c = cdiv > k ? k : cdiv;
    const int64_t groups = k / c;
    CT_REQUIRE((m * groups) % 64 == 0, "scale count must be a multiple of 64");
    if (sdt == CT_BF16)
        hipLaunchKernelGGL(marlin24_pack_scales_kernel<true>, dim3(grid_1d(m * groups)), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(scale), m,
                           groups, group_perm ? 0 : 1, static_cast<uint16_t*>(scale_packed));
    else
        hipLaunchKernelGGL(marlin24_pack_scales_kernel<false>, dim3(grid_1d(m * groups)), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(scale), m,
                           groups, group_perm ? 0 : 1, static_cast<uint16_t*>(scale_packed));
    CT_LAUNCH_CHECK("ct_marlin24_compress_w4_full");
}

int ct_marlin24_compress_w4_verdict(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                                    int group_perm, int32_t* packed, int16_t* meta, void* scale_packed, int64_t* verdict_word, void* workspace,
                                    int clear_workspace, ct_stream_t stream) {
    CT_REQUIRE(sdt == CT_F16 || sdt == CT_BF16, "marlin-24 scales must be 16-bit floats, got dtype %d", sdt);
    CT_REQUIRE(scale_packed != nullptr && verdict_word != nullptr && (reinterpret_cast<uintptr_t>(verdict_word) & 7u) == 0, "scale_packed / verdict_word NULL or misaligned");
    CT_REQUIRE(workspace != nullptr && (reinterpret_cast<uintptr_t>(workspace) & 127u) == 0, "workspace NULL or not 128-byte aligned (CT_M24_VERDICT_WORKSPACE_BYTES of device memory)");
    static int unused_flag;  // the kernel's `bad` argument is not touched in verdict mode; the shared argument check wants a non-null pointer
    return marlin24_compress_w4_impl(w, wdt, scale, sdt, zp, zdt, m, k, cdiv, packed, meta, &unused_flag, false, scale_packed, group_perm ? 0 : 1, nullptr, stream,
                                     reinterpret_cast<long long*>(verdict_word), workspace, clear_workspace != 0);
}

int ct_selftest_m24_div(int mode, uint32_t s_lo_bits, uint32_t s_hi_bits, unsigned long long* mismatches, ct_stream_t stream) {
    CT_REQUIRE((mode == 0 || mode == 1 || mode == 2) && s_lo_bits <= s_hi_bits && s_hi_bits <= 65536u, "bad mode / scale bit range");
    hipError_t e = hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), as_stream(stream));
    if (e != hipSuccess) return hip_check(e, "ct_selftest_m24_div memset");
    if (s_lo_bits == s_hi_bits) return CT_OK;
    const unsigned n = s_hi_bits - s_lo_bits;
    hipLaunchKernelGGL(selftest_m24_div_kernel, dim3(n < 4096 ? n : 4096), dim3(kBlock), 0, as_stream(stream), mode, s_lo_bits, s_hi_bits, mismatches);
    CT_LAUNCH_CHECK("ct_selftest_m24_div");
}

int ct_selftest_f16_div(uint32_t s_lo_bits, uint32_t s_hi_bits, unsigned long long* mismatches, ct_stream_t stream) {
    CT_REQUIRE(s_lo_bits <= s_hi_bits && s_hi_bits <= 65536u, "bad scale bit range");
    hipError_t e = hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), as_stream(stream));
    if (e != hipSuccess) return hip_check(e, "ct_selftest_f16_div memset");
    if (s_lo_bits == s_hi_bits) return CT_OK;
    const unsigned n = s_hi_bits - s_lo_bits;
    hipLaunchKernelGGL(selftest_f16_div_kernel, dim3(n < 4096 ? n : 4096), dim3(kBlock), 0, as_stream(stream), s_lo_bits, s_hi_bits, mismatches);
    CT_LAUNCH_CHECK("ct_selftest_f16_div");
}

int ct_marlin24_pack_weights(const void* q, int dt, int transposed, int add_offset, int64_t size_k, int64_t size_n, int bits, int32_t* packed,
                             ct_stream_t stream) {
    CT_REQUIRE(bits == 4 || bits == 8, "num_bits must be 4 or 8, got %d", bits);
    CT_REQUIRE(dt == CT_I32 || dt == CT_I8 || is_float_dt(dt), "unsupported code dtype %d", dt);
    CT_REQUIRE(size_k >= 0 && size_n >= 0 && size_k % 16 == 0 && size_n % 64 == 0, "marlin-24 needs size_k %% 16 == 0 and size_n %% 64 == 0, got (%lld, %lld)",
               (long long)size_k, (long long)size_n);
    if (size_k == 0 || size_n == 0) return CT_OK;
    const int64_t total = (size_k / 16) * (size_n * 16 * bits / 32);
    hipLaunchKernelGGL(marlin24_pack_kernel, dim3(grid_1d(total)), dim3(kBlock), 0, as_stream(stream), q, dt, transposed, add_offset ? (1 << (bits - 1)) : 0,
                       size_k, size_n, bits, packed);
    CT_LAUNCH_CHECK("ct_marlin24_pack_weights");
}

int ct_marlin24_pack_scales(const void* scale, int dt, int64_t size_n, int64_t groups, int single, void* out, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_F16 || dt == CT_BF16, "marlin-24 scales must be 16-bit floats, got dtype %d", dt);
    CT_REQUIRE(size_n >= 0 && groups >= 0 && (size_n * groups) % 64 == 0, "scale count must be a multiple of 64");
    if (size_n == 0 || groups == 0) return CT_OK;
    hipLaunchKernelGGL(marlin24_pack_scales_kernel<false>, dim3(grid_1d(size_n * groups)), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(scale),
                       size_n, groups, single, static_cast<uint16_t*>(out));
    CT_LAUNCH_CHECK("ct_marlin24_pack_scales");
}

int ct_marlin24_pack_scales_f16(const void* scale, int dt, int64_t size_n, int64_t groups, int single, void* out, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_F16 || dt == CT_BF16, "marlin-24 scales must be 16-bit floats, got dtype %d", dt);
    CT_REQUIRE(size_n >= 0 && groups >= 0 && (size_n * groups) % 64 == 0, "scale count must be a multiple of 64");
    if (size_n == 0 || groups == 0) return CT_OK;
    if (dt == CT_BF16)
        hipLaunchKernelGGL(marlin24_pack_scales_kernel<true>, dim3(grid_1d(size_n * groups)), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(scale),
                           size_n, groups, single, static_cast<uint16_t*>(out));
    else
        hipLaunchKernelGGL(marlin24_pack_scales_kernel<false>, dim3(grid_1d(size_n * groups)), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(scale),
                           size_n, groups, single, static_cast<uint16_t*>(out));
    CT_LAUNCH_CHECK("ct_marlin24_pack_scales_f16");
}

}  // extern "C"
