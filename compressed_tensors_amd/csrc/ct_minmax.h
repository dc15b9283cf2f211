// ct_minmax.h — min/max reduction and calculate_qparams device helpers shared by ct_qparams.hip (the observer
// kernel) and ct_quant.hip (the fused round-to-nearest compress).  Reference: quantization/utils/helpers.py:50-137.
#pragma once
#include "ct_common.h"

namespace ct {

struct MinMax {
    float mn, mx;
    int nan;
};

__device__ __forceinline__ MinMax mm_merge(MinMax a, MinMax b) {
    MinMax r;
    r.mn = __builtin_fminf(a.mn, b.mn);
    r.mx = __builtin_fmaxf(a.mx, b.mx);
    r.nan = a.nan | b.nan;
    return r;
}

// scale (already rounded to XDT) and zero point (an integral value in [-128, 127], or 0 for NaN) of one group
template <int XDT>
__device__ __forceinline__ void compute_qparams(MinMax m, int bits, int symmetric, float& s_out, float& z_out) {
    const float bit_max = (float)((1 << bits) / 2 - 1), bit_min = -(float)((1 << bits) / 2);
    const float bit_range = bit_max - bit_min;
    const float eps = XDT == CT_BF16 ? 0.0078125f : (XDT == CT_F16 ? 0.0009765625f : 1.1920928955078125e-07f);
    float mn = m.mn, mx = m.mx;
    if (m.nan) {
        mn = mx = __builtin_nanf("");
    } else {
        mn = mn < 0.0f ? mn : 0.0f;
        mx = mx > 0.0f ? mx : 0.0f;
    }
    float s, z;
    if (symmetric) {
        const float a = __builtin_fabsf(mn), b = __builtin_fabsf(mx);
        const float mm = m.nan ? mn : (a > b ? a : b);
        s = round_to<XDT>(mm / (bit_range / 2.0f));
        z = 0.0f;
    } else {
        s = round_to<XDT>(round_to<XDT>(mx - mn) / bit_range);
        const float zz = round_to<XDT>(bit_min - round_to<XDT>(mn / s));
        z = clamp_nan(zz, bit_min, bit_max);
    }
    if (s == 0.0f) s = eps;
    float zc = clamp_nan(z, -128.0f, 127.0f);
    zc = __builtin_rintf(zc);
    s_out = s;
    z_out = (zc != zc) ? 0.0f : zc;
}

template <int XDT>
__device__ __forceinline__ void emit_qparams(MinMax m, int bits, int symmetric, void* scale_out, int8_t* zp_out, int64_t idx) {
    float s, z;
    compute_qparams<XDT>(m, bits, symmetric, s, z);
    store1<XDT>(scale_out, idx, s);
    if (zp_out) zp_out[idx] = (int8_t)(int)z;
}

// all-lanes reduction over aligned groups of `lpg` lanes (power of two).  Inside a 16-lane row the
// exchange is DPP (no LDS crossbar traffic): quad_perm xor 1, xor 2, then row_half_mirror and
// row_mirror, which pair the quads / halves — any pairing that covers the group reduces it.
// Wider groups finish with ds_bpermute (xor 16, 32).  The shuffle form cost 12 ds_bpermute per
// 8-element unit and held the kernel at 36.6 us for 134 MB; the read-only ceiling is 21.3 us.
template <int CTRL>
__device__ __forceinline__ MinMax mm_dpp(MinMax m) {
    MinMax o;
    o.mn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m.mn), CTRL, 0xf, 0xf, false));
    o.mx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m.mx), CTRL, 0xf, 0xf, false));
    o.nan = __builtin_amdgcn_update_dpp(0, m.nan, CTRL, 0xf, 0xf, false);
    return mm_merge(m, o);
}

__device__ __forceinline__ MinMax group_reduce(MinMax m, int lpg) {
    if (lpg >= 2) m = mm_dpp<0xB1>(m);   // quad_perm [1,0,3,2]
    if (lpg >= 4) m = mm_dpp<0x4E>(m);   // quad_perm [2,3,0,1]
    if (lpg >= 8) m = mm_dpp<0x141>(m);  // row_half_mirror
    if (lpg >= 16) m = mm_dpp<0x140>(m); // row_mirror
    for (int d = 16; d < lpg; d <<= 1) {
        MinMax o;
        o.mn = __shfl_xor(m.mn, d, 64);
        o.mx = __shfl_xor(m.mx, d, 64);
        o.nan = __shfl_xor(m.nan, d, 64);
        m = mm_merge(m, o);
    }
    return m;
}

// ---- symmetric schemes on 16-bit weights: the observer only needs max |x| --------------------------------------
// For a symmetric scheme amax = max(|min(mn, 0)|, |max(mx, 0)|) = max |x|, and for non-negative floats the integer order
// of the bit patterns IS the float order, with every NaN above inf.  So the reduction runs on the raw 16-bit PAIRS:
// one v_and_b32 (drop the sign bits) and one v_pk_max_u16 per two elements — 1 VALU per element instead of 5
// (unpack, NaN test, min, max) — and one packed max per DPP step instead of three.  The result converts back exactly.
typedef uint16_t u16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t absmax_acc(uint32_t acc, uint32_t pair_bits) {
    const u16x2_t a = __builtin_bit_cast(u16x2_t, acc), b = __builtin_bit_cast(u16x2_t, pair_bits & 0x7fff7fffu);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(a, b));
}

template <int CTRL>
__device__ __forceinline__ uint32_t absmax_dpp(uint32_t acc) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, CTRL, 0xf, 0xf, false);
    return absmax_acc(acc, o);
}

__device__ __forceinline__ uint32_t absmax_group_reduce(uint32_t acc, int lpg) {
    if (lpg >= 2) acc = absmax_dpp<0xB1>(acc);
    if (lpg >= 4) acc = absmax_dpp<0x4E>(acc);
    if (lpg >= 8) acc = absmax_dpp<0x141>(acc);
    if (lpg >= 16) acc = absmax_dpp<0x140>(acc);
    for (int d = 16; d < lpg; d <<= 1) acc = absmax_acc(acc, (uint32_t)__shfl_xor((int)acc, d, 64));
    return acc;
}

// ---- asymmetric schemes on 16-bit weights: min AND max on the raw pairs (round 6) ---------------------------------
// key = b ^ ((b >> 15, arithmetic) & 0x7fff) maps a sign-magnitude 16-bit float onto the int16 whose order is the float order (negative
// values: -1 - magnitude; -0.0 -> -1 < +0.0 -> 0, which no quantization parameter can tell apart), a positive NaN above +inf and a
// negative one below -inf — so a NaN anywhere in the group ends up as the group's maximum or minimum and is seen at the end.  Three packed
// ops for the key and one packed min / max each per two elements: 2.5 VALU per element against 5 (unpack, NaN test, fmin, fmax), and two
// packed ops per DPP step against five.  The key map is an involution.
typedef int16_t i16x2_t __attribute__((ext_vector_type(2)));
struct MinMaxKey {
    uint32_t mn, mx;  // two int16 keys each
};
__device__ __forceinline__ uint32_t mm_key_pair(uint32_t pair_bits) {
    const i16x2_t sign = __builtin_bit_cast(i16x2_t, pair_bits) >> 15;
    return pair_bits ^ (__builtin_bit_cast(uint32_t, sign) & 0x7fff7fffu);
}
__device__ __forceinline__ MinMaxKey mmk_init() { return MinMaxKey{0x7fff7fffu, 0x80008000u}; }
__device__ __forceinline__ MinMaxKey mmk_merge(MinMaxKey a, uint32_t kmn, uint32_t kmx) {
    a.mn = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(i16x2_t, a.mn), __builtin_bit_cast(i16x2_t, kmn)));
    a.mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, a.mx), __builtin_bit_cast(i16x2_t, kmx)));
    return a;
}
__device__ __forceinline__ MinMaxKey mmk_acc(MinMaxKey a, uint32_t pair_bits) {
    const uint32_t k = mm_key_pair(pair_bits);
    return mmk_merge(a, k, k);
}
template <int CTRL>
__device__ __forceinline__ MinMaxKey mmk_dpp(MinMaxKey a) {
    return mmk_merge(a, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.mn, CTRL, 0xf, 0xf, false), (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.mx, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ MinMaxKey mmk_group_reduce(MinMaxKey a, int lpg) {
    if (lpg >= 2) a = mmk_dpp<0xB1>(a);
    if (lpg >= 4) a = mmk_dpp<0x4E>(a);
    if (lpg >= 8) a = mmk_dpp<0x141>(a);
    if (lpg >= 16) a = mmk_dpp<0x140>(a);
    for (int d = 16; d < lpg; d <<= 1) a = mmk_merge(a, (uint32_t)__shfl_xor((int)a.mn, d, 64), (uint32_t)__shfl_xor((int)a.mx, d, 64));
    return a;
}
// the MinMax the float reduction would have produced (up to the sign of a zero)
template <int XDT>
__device__ __forceinline__ MinMax mmk_finish(MinMaxKey a) {
    static_assert(XDT == CT_BF16 || XDT == CT_F16, "16-bit weights only");
    const int mn_lo = (int)(int16_t)(a.mn & 0xffffu), mn_hi = (int)(int16_t)(a.mn >> 16);
    const int mx_lo = (int)(int16_t)(a.mx & 0xffffu), mx_hi = (int)(int16_t)(a.mx >> 16);
    const int kmn = mn_lo < mn_hi ? mn_lo : mn_hi, kmx = mx_lo > mx_hi ? mx_lo : mx_hi;
    const uint32_t bmn = ((uint32_t)kmn ^ ((uint32_t)(kmn >> 15) & 0x7fffu)) & 0xffffu, bmx = ((uint32_t)kmx ^ ((uint32_t)(kmx >> 15) & 0x7fffu)) & 0xffffu;
    constexpr uint32_t inf = XDT == CT_BF16 ? 0x7f80u : 0x7c00u;
    MinMax m;
    m.nan = ((bmn & 0x7fffu) > inf) | ((bmx & 0x7fffu) > inf);
    m.mn = XDT == CT_BF16 ? bits_f(bmn << 16) : f16_bits_to_f(bmn);
    m.mx = XDT == CT_BF16 ? bits_f(bmx << 16) : f16_bits_to_f(bmx);
    return m;
}

// the MinMax the float reduction would have produced, as far as a symmetric scheme can tell: {0, amax, nan}
template <int XDT>
__device__ __forceinline__ MinMax absmax_finish(uint32_t acc) {
    static_assert(XDT == CT_BF16 || XDT == CT_F16, "16-bit weights only");
    const uint32_t h = (acc & 0xffffu) > (acc >> 16) ? (acc & 0xffffu) : (acc >> 16);
    MinMax m;
    m.nan = h > (XDT == CT_BF16 ? 0x7f80u : 0x7c00u);
    m.mn = 0.0f;
    m.mx = XDT == CT_BF16 ? bits_f(h << 16) : f16_bits_to_f(h);
    return m;
}

// calculate_qparams of the symmetric FLOAT schemes (helpers.py:50-137, mxfp_utils.py:37-143); see ct_hip.h for the kinds.
// One lane per group runs this.
enum { QP_INT = 0, QP_FP8 = 1, QP_NVFP4 = 2, QP_MXFP4 = 3, QP_MXFP8 = 4, QP_AMAX = 5 };

template <int XDT>
__device__ __forceinline__ float compute_qparams_float(MinMax m, int kind, float gs) {
    float amax;
    if (m.nan) {
        amax = __builtin_nanf("");
    } else {
        const float mn = m.mn < 0.0f ? m.mn : 0.0f, mx = m.mx > 0.0f ? m.mx : 0.0f;
        const float a = __builtin_fabsf(mn), b = __builtin_fabsf(mx);
        amax = a > b ? a : b;
    }
    if (kind == QP_AMAX) return amax;  // the raw reduction (generate_gparam's input)
    if (kind == QP_FP8) {
        const float eps = XDT == CT_BF16 ? 0.0078125f : (XDT == CT_F16 ? 0.0009765625f : 1.1920928955078125e-07f);
        const float s = round_to<XDT>(amax / 448.0f);
        return s == 0.0f ? eps : s;
    }
    if (kind == QP_NVFP4) {
        const float s2 = gs * round_to<XDT>(amax / 6.0f);           // float32: global * local
        const float s = fp8_round(clamp_nan(s2, -448.0f, 448.0f));  // round_to_quantized_type_dtype(float8_e4m3fn)
        return s == 0.0f ? 0.125f : s;
    }
    // MX: the significand is rounded at a quarter and masked off (round_to_power_2), the exponent goes through uint8
    constexpr int mant = XDT == CT_BF16 ? 7 : (XDT == CT_F16 ? 10 : 23);
    constexpr int expo = XDT == CT_F16 ? 5 : 8;
    uint32_t bits = XDT == CT_BF16 ? f_to_bf16_bits(amax) : (XDT == CT_F16 ? f_to_f16_bits(amax) : f_bits(amax));
    bits = (bits + (1u << (mant - 2))) & (((1u << (expo + 1)) - 1u) << mant);
    if (XDT != CT_F32) bits &= 0xffffu;
    const float p = XDT == CT_BF16 ? bf16_bits_to_f(bits) : (XDT == CT_F16 ? f16_bits_to_f(bits) : bits_f(bits));
    // floor(log2(p)) of a power of two is its exponent; 0 -> -inf, inf -> inf, NaN -> NaN, as torch.log2.  p is never a
    // non-zero subnormal: the mask keeps sign and exponent only, so an exponent field of zero is the value zero
    float l;
    if (p != p) l = p;
    else if (p == 0.0f) l = -__builtin_inff();
    else if (__builtin_isinf(p)) l = __builtin_inff();
    else l = (float)((int)((f_bits(p) >> 23) & 0xffu) - 127);
    const float offset = kind == QP_MXFP4 ? 2.0f : 8.0f;
    const float e = round_to<XDT>(round_to<XDT>(127.0f + l) - offset);
    float ec = clamp_nan(e, 0.0f, 255.0f);
    ec = __builtin_rintf(ec);
    const int code = (ec != ec) ? 0 : (int)ec;
    // 2^(code - 127) as float32 (2^-127 is a float32 subnormal, 2^128 is inf), then to X
    const float two = code == 0 ? 0x1p-127f : (code == 255 ? __builtin_inff() : bits_f((uint32_t)code << 23));
    const float s = round_to<XDT>(two);
    return s == 0.0f ? 1.0f : s;
}

}  // namespace ct
