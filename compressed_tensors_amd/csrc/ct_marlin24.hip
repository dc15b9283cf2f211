// ct_marlin24.hip — CUTLASS 2:4 semi-structured conversion and marlin-24 packing for gfx950.
//
// Reference: utils/semi_structured_conversions.py:33-298 (compress/decompress + metadata
// reordering), utils/permutations_24.py:20-53 (marlin-24 permutation tables).  The
// Marlin24Compressor class itself is absent from the reference snapshot;
// the packing restated here follows its historical definition and is pinned against the CPU
// oracle, which in turn is pinned against the surviving reference primitives.
#include "ct_common.h"

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
        if (violation) atomicOr(bad, 1);
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
    if (violation) atomicOr(bad, 1);
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
    if (violation) atomicOr(bad, 1);
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

int ct_marlin24_compress_w4(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt, int64_t m, int64_t k, int64_t cdiv,
                            int32_t* packed, int16_t* meta, int* bad, ct_stream_t stream) {
    CT_REQUIRE(wdt == CT_F16 || wdt == CT_BF16, "marlin-24 weights must be 16-bit floats, got dtype %d", wdt);
    CT_REQUIRE(is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(m >= 0 && k >= 0 && m % 64 == 0 && k % 256 == 0, "the fused marlin-24 path needs rows %% 64 == 0 and cols %% 256 == 0, got (%lld, %lld)",
               (long long)m, (long long)k);
    CT_REQUIRE(cdiv >= 16 && (cdiv % 16 == 0 || cdiv >= k), "group size %lld must be a multiple of 16", (long long)cdiv);
    CT_REQUIRE(aligned16(w) && (reinterpret_cast<uintptr_t>(meta) & 7u) == 0 && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0 && bad != nullptr,
               "misaligned buffers");
    CT_REQUIRE((m / 64) * (k / 256) < ((int64_t)1 << 31), "tensor too large for one launch");
    hipError_t e = hipMemsetAsync(bad, 0, sizeof(int), as_stream(stream));
    if (e != hipSuccess) return hip_check(e, "ct_marlin24_compress_w4 memset");
    if (m == 0 || k == 0) return CT_OK;
    const int64_t c = cdiv > k ? k : cdiv;
    const unsigned tg = (unsigned)((m / 64) * (k / 256));
    if (wdt == CT_BF16)
        hipLaunchKernelGGL((marlin24_fused_w4_kernel<CT_BF16>), dim3(tg), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c, k / c, packed,
                           reinterpret_cast<uint16_t*>(meta), bad);
    else
        hipLaunchKernelGGL((marlin24_fused_w4_kernel<CT_F16>), dim3(tg), dim3(kBlock), 0, as_stream(stream), w, scale, sdt, zp, zdt, m, k, c, k / c, packed,
                           reinterpret_cast<uint16_t*>(meta), bad);
    CT_LAUNCH_CHECK("ct_marlin24_compress_w4");
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
