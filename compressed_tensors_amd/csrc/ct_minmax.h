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

}  // namespace ct
