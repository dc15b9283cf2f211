// ct_quant_lean.h — the per-word arithmetic of the lean 16-bit compress kernels (moved out of ct_quant.hip in round 6, verbatim, so that
// the any-width lean kernels of ct_quant_wb.hip share it): the hardware float->int conversion, the fast-path predicates (reciprocal
// instead of the IEEE divide where that is proven bit-identical), the packed-fp16 back end and the W4 word builder.
#pragma once
#include "ct_quant_core.h"

namespace ct {

// v_cvt_i32_f32: saturating, NaN -> 0.  Spelled as an instruction so that clang does not expand
// the (well-defined under -fno-strict-float-cast-overflow) conversion into compare/select chains.
__device__ __forceinline__ int cvt_i32_hw(float x) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// two elements of a 16-bit pair as floats
template <int DT>
__device__ __forceinline__ void unpack2(uint32_t w, float& a, float& b) {
    if constexpr (DT == CT_BF16) { a = bits_f(w << 16); b = bits_f(w & 0xffff0000u); }
    else { a = f16_bits_to_f(w & 0xffffu); b = f16_bits_to_f(w >> 16); }
}

// round two floats to DT and back (one v_cvt_pk_bf16_f32 for bf16)
template <int DT>
__device__ __forceinline__ void round2(float& a, float& b) {
    if constexpr (DT == CT_BF16) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef bf16_t b2 __attribute__((ext_vector_type(2)));
        const uint32_t p = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, b2));
        a = bits_f(p << 16); b = bits_f(p & 0xffff0000u);
    } else {
        a = round_to<DT>(a); b = round_to<DT>(b);
    }
}

// ---- fast-path predicates --------------------------------------------------------------------------------------
// bf16: x * fl(1/s) == x / s after the rounding to bf16 for 2^-64 <= |s| <= 2^64 (ct_selftest_bf16_div).
// fp16: reciprocal + one Newton step == the IEEE quotient after the rounding to fp16 for 2^-14 <= |s| <= 2^15 and every
// FINITE x (ct_selftest_f16_div; quotients below 2^-13 may differ in the last subnormal place and all become code 0,
// with or without an integer zero point) — the finiteness of a lane's 32 weights is one v_dot2c_f32_f16 per pair.
template <int DT>
__device__ __forceinline__ bool fast_scale_ok(float s) {
    const float as = __builtin_fabsf(s);
    if constexpr (DT == CT_BF16) return (as >= 0x1p-64f) && (as <= 0x1p64f);
    else if constexpr (DT == CT_F16) return (as >= 0x1p-14f) && (as <= 0x1p15f);
    else return false;
}
typedef _Float16 qh2_t __attribute__((ext_vector_type(2)));
template <int DT, int Q>
__device__ __forceinline__ bool fast_data_ok(const u32x4 (&r)[Q]) {
    if constexpr (DT != CT_F16) return true;
    else {
        float acc = 0.0f;  // <= 32 * 65504^2 when everything is finite; inf / NaN propagate
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(qh2_t, ws[j]), __builtin_bit_cast(qh2_t, ws[j]), acc, false);
        }
        return acc <= 3.0e38f;
    }
}

// fp16 weights, fast path: everything after the fp32 quotient works on fp16 PAIRS (derivation: ct_marlin24.hip) —
// v_cvt_pk_f16_f32 is the rounding to T, the zero-point add is v_pk_add_f16 (the reference adds in fp16 too), clamp =
// v_pk_max_f16 / v_pk_min_f16 (no NaN can reach it: fast_data_ok), round-half-even + integer cast + bias in ONE
// v_pk_add_f16: for |t| <= 128, fl16(t + MAGIC) = MAGIC + rint(t) exactly (ulp = 1 there) and the low byte of each half is
// the code plus MAGIC's low byte.  5.5 VALU per element instead of ~20 with the IEEE divide.
// pairs[j] = 0x66cc66cc-style halves; returns them un-gathered
template <bool ZP, int MAGIC>
__device__ __forceinline__ void quant_pairs_f16(const u32x4& raw, float s, float rs, float z, float qmin, float qmax, uint32_t (&u)[4]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 rs2 = {rs, rs}, s2 = {s, s};
    const qh2_t lo2 = {(_Float16)qmin, (_Float16)qmin}, hi2 = {(_Float16)qmax, (_Float16)qmax};
    const qh2_t magic = {(_Float16)(float)MAGIC, (_Float16)(float)MAGIC}, z2 = {(_Float16)z, (_Float16)z};
    const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f2 x = __builtin_convertvector(__builtin_bit_cast(qh2_t, ws[j]), f2);
        f2 t = x * rs2;
        t = __builtin_elementwise_fma(__builtin_elementwise_fma(-t, s2, x), rs2, t);
        qh2_t t16 = __builtin_convertvector(t, qh2_t);
        if (ZP) t16 = t16 + z2;
        t16 = __builtin_elementwise_min(__builtin_elementwise_max(t16, lo2), hi2) + magic;
        u[j] = __builtin_bit_cast(uint32_t, t16);
    }
}

template <bool ZP>
__device__ __forceinline__ uint32_t w4_quant_word_f16(const u32x4& raw, float s, float rs, float z) {
    uint32_t u[4];
    quant_pairs_f16<ZP, 1544>(raw, s, rs, z, -8.0f, 7.0f, u);  // 1544 = 1536 + 8: the low byte is the biased nibble
    const uint32_t p0 = __builtin_amdgcn_perm(u[1], u[0], 0x06040200u), p1 = __builtin_amdgcn_perm(u[3], u[2], 0x06040200u);
    const uint32_t a = p0 | __builtin_amdgcn_alignbit(p0, p0, 4), b = p1 | __builtin_amdgcn_alignbit(p1, p1, 4);
    return __builtin_amdgcn_perm(b, a, 0x06040200u);  // bytes 0 / 2 of a and b: nibble pairs (0,1) (2,3) (4,5) (6,7)
}

// 8 weights (16 B) -> one packed word.  FAST: x * (1/s) instead of x / s — bit-identical after the
// rounding to bf16 for every bf16 x and every bf16 s with 2^-64 <= |s| <= 2^64 (no quotient of two
// 8-bit significands lies within 2^-17 relative of a bf16 rounding boundary, while the
// two-rounding error of x * fl(1/s) is < 2^-22 relative); proven exhaustively on the device by
// ct_selftest_bf16_div (tests/test_gpu_parity.py).  The codes are accumulated as
// 0x88888888 + sum(code_k << 4k): code_k in [-8, 7], so the biased nibbles never carry.
template <int DT, bool FAST, bool ZP>
__device__ __forceinline__ uint32_t w4_quant_word(const u32x4& raw, float s, float rs, float z) {
    if constexpr (DT == CT_F16 && FAST) return w4_quant_word_f16<ZP>(raw, s, rs, z);
    const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t word = 0x88888888u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0, x1;
        unpack2<DT>(ws[j], x0, x1);
        float t0 = FAST ? x0 * rs : x0 / s, t1 = FAST ? x1 * rs : x1 / s;
        round2<DT>(t0, t1);
        if (ZP) {
            t0 += z; t1 += z;
            round2<DT>(t0, t1);
        }
        int c0 = cvt_i32_hw(__builtin_rintf(t0)), c1 = cvt_i32_hw(__builtin_rintf(t1));
        c0 = c0 < -8 ? -8 : (c0 > 7 ? 7 : c0);  // v_med3_i32
        c1 = c1 < -8 ? -8 : (c1 > 7 ? 7 : c1);
        word += (uint32_t)c0 << (8 * j);
        word += (uint32_t)c1 << (8 * j + 4);
    }
    return word;
}

}  // namespace ct
