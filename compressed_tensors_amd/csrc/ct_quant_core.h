// ct_quant_core.h — the pieces of the quantize / dequantize translation units that more than one of them needs: the
// parameter block, the arithmetic cores, scale / zero-point addressing, unit stores and the host-side layout helpers.
// (The any-bit-width pack-group kernels are compiled in two translation units of their own, ct_quant_g32_lo.hip /
// ct_quant_g32_hi.hip: their 64 instantiations took 100 of the 117 s this library needed to build.)
#pragma once
#include "ct_common.h"

namespace ct {

struct QParams {
    const void* x;      // weight (quantize) or int8 codes (dequantize)
    const void* scale;
    const void* zp;     // nullable
    void* out;
    int xdt, sdt, zdt, odt;
    QLayout L;
    float qmin, qmax;
    int vec;            // 16-byte vector path allowed (cols % 8 == 0, pointers aligned)
    int fkind;          // 0: INT codes (rint); 1: FLOAT 8-bit (round to float8_e4m3fn); 2: FLOAT 4-bit (cast_to_fp4)
    const float* gscale;  // nullable: effective scale = fl32(scale / gscale[0]), arithmetic in float32
    int sdt_arith;      // dtype the dequantize arithmetic runs in: sdt, or CT_F32 under a global scale
};

// ------------------------------------------------------------------------------------------
// cores
// ------------------------------------------------------------------------------------------
// reciprocal of a bf16 scale, or 0 when the shortcut x * fl(1/s) is not proven equal to x / s after the
// rounding to bf16 (see w4_quant_word / ct_selftest_bf16_div): the caller then divides
__device__ __forceinline__ float bf16_fast_rcp(float s) {
    const float as = __builtin_fabsf(s);
    return ((as >= 0x1p-64f) && (as <= 0x1p64f)) ? 1.0f / s : 0.0f;
}

// reciprocal of a scale for fp32 arithmetic, or 0 when the caller has to divide.  x * fl(1/s) is NOT the IEEE quotient in fp32, but
// an integer code only depends on which side of a half-integer the (shifted, clamped) quotient falls: quant_core tests that with
// an error bound and divides only for the elements that are too close to call (about 1e-5 of them for int4).
__device__ __forceinline__ float f32_fast_rcp(float s) {
    const float as = __builtin_fabsf(s);
    return ((as >= 0x1p-100f) && (as <= 0x1p100f)) ? __builtin_amdgcn_rcpf(s) : 0.0f;  // v_rcp_f32: 1 ulp
}

// reciprocal of an fp16 scale, or 0 outside the range in which reciprocal + one Newton step is proven to give the fp16 rounding of
// the IEEE quotient (ct_selftest_f16_div: every fp16 x, quotients below 2^-13 excepted — they all end as code / value 0)
__device__ __forceinline__ float f16_newton_rcp(float s) {
    const float as = __builtin_fabsf(s);
    return ((as >= 0x1p-14f) && (as <= 0x1p15f)) ? 1.0f / s : 0.0f;
}

// the quotient x / s before its rounding to T: IEEE divide, or the proven shortcut for T (rs = 0 selects the divide)
template <int TDT>
__device__ __forceinline__ float fast_quotient(float x, float s, float rs) {
    if (rs == 0.0f) return x / s;
    if constexpr (TDT == CT_F16) {
        const float q0 = x * rs;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, s, x), rs, q0);
        // x = +-inf / NaN: the correction would turn inf into NaN; x = -0: it would return +0 ((+0) + (-0)), and q0 is exact there
        float r = (__builtin_isfinite(q0) && x != 0.0f) ? q1 : q0;
        // The shortcut is proven down to 2^-13 (ct_selftest_f16_div); below it the two can differ in the last fp16-subnormal place,
        // which an integer code never sees but a float-typed result can (an underflow to zero changes the sign `+ zero_point`
        // leaves).  Those few elements take the divide: a divergent branch that most waves skip (x == 0 is exact either way).
        if (__builtin_fabsf(r) < 0x1p-13f && x != 0.0f) r = x / s;
        return r;
    } else if constexpr (TDT == CT_F32) {
        return x / s;  // rs only feeds quant_core's half-integer test
    } else {
        return x * rs;
    }
}

template <int TDT>
__device__ __forceinline__ float quant_core(float x, float s, bool has_zp, float zf, float qmin,
                                            float qmax, float rs = 0.0f, int fkind = 0) {
    if constexpr (TDT == CT_F32) {
        // fp32, INT codes: with rs = 1/s to 1 ulp (v_rcp_f32), q0 = x * rs is within 2^-22 |q0| of fl(x / s), and the zero-point add
        // rounds once more, so the value that gets clamped and rounded is within E = 2^-21 (|q0| + |z| + 1) of the reference's.
        // Clamping is 1-Lipschitz: if the clamped value is at least E away from every half-integer, both round to the same integer.
        // The test is written so that a NaN anywhere (x, or an E that overflowed) fails it and takes the divide below; the clamp can
        // then be a plain v_med3_f32.
        if (rs != 0.0f && fkind == 0) {
            const float q0 = x * rs;
            const float a = has_zp ? q0 + zf : q0;
            const float c = __builtin_amdgcn_fmed3f(a, qmin, qmax);
            const float r = __builtin_rintf(c);
            const float tol = __builtin_fmaf(__builtin_fabsf(q0), 0x1p-21f, (__builtin_fabsf(zf) + 1.0f) * 0x1p-21f);
            if (__builtin_fabsf(__builtin_fabsf(c - r) - 0.5f) >= tol) return r;  // (a NaN or infinite x makes tol NaN / inf: false)
        }
    }
    float t = round_to<TDT>(fast_quotient<TDT>(x, s, rs));  // IEEE-correct fp32 divide (or the proven bf16 / fp16 shortcut), RNE to T
    if (has_zp) t = round_to<TDT>(t + zf);
    t = clamp_nan(t, qmin, qmax);
    // INT: v_rndne_f32 (round half to even).  FLOAT 8-bit: tensor.to(float8_e4m3fn) (quant_args.py:463-486); the
    // value is exact in every T
    return fkind == 2 ? fp4_round(t) : fkind ? fp8_round(t) : __builtin_rintf(t);
}

template <int SDT>
__device__ __forceinline__ float dequant_core(float q, bool has_zp, float zf, float s) {
    float d = q;
    if (has_zp) d = round_to<SDT>(d - zf);
    return mul_round_to<SDT>(d, s);
}

// int8 zero point and an int8 code: |q - z| <= 255 is an integer every supported float dtype holds exactly,
// so the reference's rounding of the difference is the identity and is not issued
template <int SDT>
__device__ __forceinline__ float dequant_core_zexact(float q, float zf, float s) {
    return mul_round_to<SDT>(q - zf, s);
}

// scale / zero point of element (row, c)
struct SZ {
    float s, z;
};

template <int XDT>
__device__ __forceinline__ SZ load_sz_q(const QParams& p, int64_t srow, int64_t c) {
    int64_t si = srow + col_group_of(p.L, c);
    SZ r;
    r.s = load_rt(p.scale, p.sdt, si);
    if (p.gscale) r.s = r.s / p.gscale[0];  // scale / global_scale, float32 (forward_helpers.py:535-538)
    r.z = p.zp ? round_to<XDT>(load_rt(p.zp, p.zdt, si)) : 0.0f;  // zp.to(x.dtype)
    return r;
}

template <int SDT>
__device__ __forceinline__ SZ load_sz_dq(const QParams& p, int64_t srow, int64_t c) {
    int64_t si = srow + col_group_of(p.L, c);
    SZ r;
    r.s = load_rt(p.scale, p.sdt, si);  // SDT is the arithmetic dtype; the storage dtype differs under a global scale
    if (p.gscale) r.s = r.s / p.gscale[0];
    r.z = p.zp ? round_to<SDT>(load_rt(p.zp, p.zdt, si)) : 0.0f;  // zp.to(scale.dtype)
    return r;
}

__device__ __forceinline__ bool unit_uniform(const QLayout& L, int64_t c0, int n) {
    if (L.col_group) return false;
    return col_group_of(L, c0) == col_group_of(L, c0 + n - 1);
}

// store n (<= 8) floats as dtype odt starting at flat index i0
__device__ __forceinline__ void store_unit(void* out, int odt, int64_t i0, const float (&v)[8], int n,
                                           bool vec) {
    if (vec && n == 8) {
        switch (odt) {
            case CT_F32: store8<CT_F32>(out, i0, v); return;
            case CT_F16: store8<CT_F16>(out, i0, v); return;
            case CT_BF16: store8<CT_BF16>(out, i0, v); return;
            case CT_I8: {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lo |= ((uint32_t)(int)v[k] & 0xffu) << (8 * k);
                    hi |= ((uint32_t)(int)v[4 + k] & 0xffu) << (8 * k);
                }
                stream_store8(static_cast<int8_t*>(out) + i0, u32x2{lo, hi});
                return;
            }
            case CT_F8E4M3: {
                const uint32_t lo = f2_to_fp8x2(v[0], v[1]) | (f2_to_fp8x2(v[2], v[3]) << 16);
                const uint32_t hi = f2_to_fp8x2(v[4], v[5]) | (f2_to_fp8x2(v[6], v[7]) << 16);
                stream_store8(static_cast<uint8_t*>(out) + i0, u32x2{lo, hi});
                return;
            }
            default: break;
        }
    }
    for (int k = 0; k < n; ++k) store_rt(out, odt, i0 + k, v[k]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
inline int check_layout(int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols) {
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape (%lld, %lld)", (long long)rows, (long long)cols);
    CT_REQUIRE(rdiv >= 1 && cdiv >= 1 && scale_cols >= 1, "rdiv/cdiv/scale_cols must be >= 1");
    return CT_OK;
}

inline QLayout make_layout(int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                           const int32_t* col_group) {
    QLayout L;
    L.rows = rows; L.cols = cols; L.rdiv = rdiv; L.cdiv = cdiv; L.scale_cols = scale_cols;
    L.col_group = col_group;
    L.cdiv_shift = log2_exact(cdiv);
    return L;
}

inline dim3 grid_2d(int64_t rows, int64_t items_per_row) {
    int64_t gx = cdiv64(items_per_row, kBlock);
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    int64_t gy = rows < 1 ? 1 : rows;
    // keep the total around a few thousand workgroups; rows beyond that are grid-strided
    int64_t cap = (8 * kCUs * 4) / gx;
    if (cap < 1) cap = 1;
    if (gy > cap) gy = cap;
    if (gy > 65535) gy = 65535;
    return dim3((unsigned)gx, (unsigned)gy, 1);
}

inline bool zdt_ok(int zdt) {
    return zdt == CT_I8 || zdt == CT_I32 || zdt == CT_F32 || zdt == CT_F16 || zdt == CT_BF16 ||
           zdt == CT_I64 || zdt == CT_U8 || zdt == CT_I16 || zdt == CT_F8E4M3;
}

// valid (xdt, tdt) pairs: T is the promotion of x.dtype with the scale dtype
inline bool xt_ok(int xdt, int tdt) {
    if (!is_float_dt(xdt) || !is_float_dt(tdt)) return false;
    return tdt == xdt || tdt == CT_F32;
}

#define CT_DISPATCH_XT(xdt, tdt, ...)                                                              \
    do {                                                                                           \
        if (xdt == CT_BF16 && tdt == CT_BF16) { constexpr int X = CT_BF16, T = CT_BF16; __VA_ARGS__; } \
        else if (xdt == CT_BF16 && tdt == CT_F32) { constexpr int X = CT_BF16, T = CT_F32; __VA_ARGS__; } \
        else if (xdt == CT_F16 && tdt == CT_F16) { constexpr int X = CT_F16, T = CT_F16; __VA_ARGS__; } \
        else if (xdt == CT_F16 && tdt == CT_F32) { constexpr int X = CT_F16, T = CT_F32; __VA_ARGS__; } \
        else { constexpr int X = CT_F32, T = CT_F32; __VA_ARGS__; }                                \
    } while (0)

#define CT_DISPATCH_BITS(bits, ...)                         \
    switch (bits) {                                         \
        case 1: { constexpr int B = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int B = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int B = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int B = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int B = 5; __VA_ARGS__; } break; \
        case 6: { constexpr int B = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int B = 7; __VA_ARGS__; } break; \
        case 8: { constexpr int B = 8; __VA_ARGS__; } break; \
    }

inline int fill_qparams(QParams& p, const void* x, int xdt, const void* scale, int sdt, const void* zp,
                        int zdt, int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv,
                        int64_t scale_cols, const int32_t* col_group, int bits, void* out, int odt) {
    int rc = check_layout(rows, cols, rdiv, cdiv, scale_cols);
    if (rc) return rc;
    CT_REQUIRE(is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(zp == nullptr || zdt_ok(zdt), "zero-point dtype code %d unsupported", zdt);
    p.x = x; p.scale = scale; p.zp = zp; p.out = out;
    p.xdt = xdt; p.sdt = sdt; p.zdt = zdt; p.odt = odt;
    p.L = make_layout(rows, cols, rdiv, cdiv, scale_cols, col_group);
    p.qmax = (float)((1 << bits) / 2 - 1);
    p.qmin = -(float)((1 << bits) / 2);
    p.vec = (cols % 8 == 0) && aligned16(x) && aligned16(out);
    p.fkind = 0;
    p.gscale = nullptr;
    p.sdt_arith = sdt;
    return CT_OK;
}

// launchers of the any-bit-width kernels (ct_quant_g32.inc): bits 1-4 and 5-8 live in different translation units
int launch_quant_pack_g32_lo(const QParams& p, int xdt, int tdt, int bits, int64_t packed_cols, dim3 grid, ct_stream_t stream);
int launch_quant_pack_g32_hi(const QParams& p, int xdt, int tdt, int bits, int64_t packed_cols, dim3 grid, ct_stream_t stream);
int launch_unpack_dequant_g32_lo(const QParams& p, int sdt, int bits, int64_t words, dim3 grid, ct_stream_t stream);
int launch_unpack_dequant_g32_hi(const QParams& p, int sdt, int bits, int64_t words, dim3 grid, ct_stream_t stream);
// the lean kernels of the widths next to 4 and 8 (ct_quant_wb.hip): `wb_layout_ok` says whether they take the call
bool wb_layout_ok(int dt, int sdt, int other_dt, int bits, int zdt, const void* zp, int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                  const int32_t* col_group, const void* wide, const void* words);
int launch_wb_quant_pack(const void* x, int xdt, const void* scale, const void* zp, int64_t rows, int64_t cols, int64_t cdiv, int bits, int32_t* packed,
                         ct_stream_t stream);
int launch_wb_unpack_dequant(const int32_t* packed, const void* scale, int sdt, const void* zp, int64_t rows, int64_t cols, int64_t cdiv, int bits, void* out,
                             ct_stream_t stream);

}  // namespace ct
