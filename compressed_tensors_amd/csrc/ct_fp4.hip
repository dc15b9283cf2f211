// ct_fp4.hip — FP4 (E2M1) pack-quantized codecs for gfx950: nvfp4-pack-quantized (group 16, fp8-e4m3 group scales
// under a float32 global scale) and mxfp4-pack-quantized (group 32, E8M0 power-of-two scales).  SURVEY.md §8f N4.
//
// Reference: compressors/nvfp4/base.py:68-139 + helpers.py:108-193 (pack / unpack), mxfp4/base.py:27-65,
// mx_utils.py:18-44, quantization/quant_args.py:49-68 + utils/fp4_utils.py:77-98 (cast_to_fp4),
// lifecycle/forward_helpers.py:523-572 (the global-scale arithmetic).
//
// Arithmetic model (T = torch promotion of x.dtype with the effective scale's dtype):
//   compress:   s_eff = global ? fl32(float(scale) / global) : scale;  t = rnd_T(float(x) / float(s_eff));
//               clamp to [-6, 6]; round to the E2M1 grid, ties to the even mantissa (0.25 -> 0, 0.75 -> 1, 1.25 -> 1,
//               1.75 -> 2, 2.5 -> 2, 3.5 -> 4, 5 -> 4: exactly cast_to_fp4's thresholds); nibble = sign << 3 | index,
//               where a NEGATIVE value that rounds to zero keeps its sign bit (0 * -1 = -0.0 upstream) but an input
//               of -0.0 does not (torch.sign(-0.0) == 0); element k of a row sits in bits [4k, 4k+4) of the row's
//               byte stream (low nibble first) — the same unit stream as the int4 path.
//   decompress: v = E2M1 value of the nibble (exact in bf16); NVFP4: s = bf16(fp8 scale) [exact],
//               s_eff = fl32(float(s) / global), y = rnd_bf16(fl32(v * s_eff)); MXFP4: s = 2^(e - 127), y = rnd_bf16(v * s).
// The rounding itself is the hardware's: v_cvt_scalef32_pk_fp4_f32 (two floats -> one byte, RNE, saturating at
// +-6) and v_cvt_scalef32_pk_f32_fp4 back; both are checked against the oracle over every bf16 input on the device
// (tests/test_gpu_parity.py).  -0.0 inputs are folded to +0.0 with one add before the conversion.
#include "ct_common.h"
#include "ct_minmax.h"

namespace ct {

enum { SC_PLAIN = 0, SC_F8E4M3 = 1, SC_E8M0 = 2 };

// four consecutive float pairs -> one 32-bit word of 8 nibbles
__device__ __forceinline__ uint32_t fp4_word(const float (&t)[8]) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[0] + 0.0f, t[1] + 0.0f, 1.0f, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[2] + 0.0f, t[3] + 0.0f, 1.0f, 1);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[4] + 0.0f, t[5] + 0.0f, 1.0f, 2);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[6] + 0.0f, t[7] + 0.0f, 1.0f, 3);
    return w;
}

// code word of 8 values WITHOUT the -0.0 fold: bit 3 is the IEEE sign bit, which is pack_fp4_to_uint8's
// torch.signbit (helpers.py:139-145)
__device__ __forceinline__ uint32_t fp4_word_signbit(const float (&t)[8]) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[0], t[1], 1.0f, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[2], t[3], 1.0f, 1);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[4], t[5], 1.0f, 2);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, t[6], t[7], 1.0f, 3);
    return w;
}

// ---- the quotient t = x / s_eff --------------------------------------------------------------------------------
// The IEEE divide costs ~10 VALU instructions per element (the first form of this kernel: 48.8 us for 8192^2 against
// ~30 us of memory time).  Per unit of 8 elements that share one scale, two exact shortcuts:
//  * s_eff a power of two in [2^-100, 2^100] (every MX scale): x * (1 / s_eff) is the same real number, rounded once.
//    When the reference keeps the quotient in x's own dtype (scale dtype == x dtype) the only visible effect of that
//    extra rounding is a quotient that underflows to zero, which loses its sign: |q| <= half the dtype's smallest
//    subnormal -> +0 (this also folds -0.0 inputs).
//  * any s_eff in [2^-20, 2^10] (NVFP4: fl32(scale / global_scale)): r = rcp(s) refined once (shared by the unit),
//    q0 = x * r, q1 = fma(fma(-s, q0, x), r, q0).  q1 == fl32(x / s) bit for bit whenever |q0| >= 2^-4: shown by
//    ct_selftest_fp4_div over ALL 2^23 mantissas of s x all mantissas of x (the sequence is invariant under powers
//    of two while every intermediate is normal, which the ranges guarantee); below 2^-4 both are < 0.25 and non-zero
//    with the sign of x (s <= 2^10 keeps x * r above the fp32 underflow), which is all the code depends on.
//    +-0 in gives +0 out of the fma chain; |x| is clamped to 8 * s first (code 7 either way; keeps inf out of the fma).
template <int XDT>
__device__ __forceinline__ float fp4_tiny_half() {
    return XDT == CT_BF16 ? 0x1p-134f : 0x1p-25f;
}

__device__ __forceinline__ float fp4_refined_rcp(float s) {
    const float r0 = __builtin_amdgcn_rcpf(s);
    return __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
}

__device__ __forceinline__ float fp4_fast_quotient(float x, float s, float r, float lim) {
    const float xc = __builtin_amdgcn_fmed3f(x, -lim, lim);
    const float q0 = xc * r;
    return __builtin_fmaf(__builtin_fmaf(-s, q0, xc), r, q0);
}

template <int XDT>
__device__ __forceinline__ void fp4_unpack_pair(uint32_t w, float& x0, float& x1) {
    if constexpr (XDT == CT_BF16) { x0 = bits_f(w << 16); x1 = bits_f(w & 0xffff0000u); }
    else { x0 = f16_bits_to_f(w & 0xffffu); x1 = f16_bits_to_f(w >> 16); }
}

// one unit: 8 elements (4 dwords) under one scale -> one word of 8 nibbles
template <int XDT, bool GLOBAL>
__device__ __forceinline__ uint32_t fp4_quant_unit(const uint32_t (&ws)[4], float s, float gs, bool in_dtype) {
    const float s_eff = GLOBAL ? s / gs : s;  // fl32(scale / global_scale)
    const uint32_t sb = f_bits(s_eff);
    float t[8];
    if (!GLOBAL && (sb & 0x807fffffu) == 0 && sb >= (27u << 23) && sb <= (227u << 23)) {
        const float rinv = bits_f(0x7f000000u - sb);
        const float tiny = in_dtype ? fp4_tiny_half<XDT>() : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0, x1;
            fp4_unpack_pair<XDT>(ws[j], x0, x1);
            const float q0 = x0 * rinv, q1 = x1 * rinv;
            t[2 * j] = __builtin_fabsf(q0) > tiny ? q0 : 0.0f;
            t[2 * j + 1] = __builtin_fabsf(q1) > tiny ? q1 : 0.0f;
        }
        return fp4_word_signbit(t);
    }
    if (GLOBAL && s_eff >= 0x1p-20f && s_eff <= 0x1p10f) {
        const float r = fp4_refined_rcp(s_eff), lim = 8.0f * s_eff;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0, x1;
            fp4_unpack_pair<XDT>(ws[j], x0, x1);
            t[2 * j] = fp4_fast_quotient(x0, s_eff, r, lim);
            t[2 * j + 1] = fp4_fast_quotient(x1, s_eff, r, lim);
        }
        return fp4_word_signbit(t);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0, x1;
        fp4_unpack_pair<XDT>(ws[j], x0, x1);
        float q0 = x0 / s_eff, q1 = x1 / s_eff;
        if (in_dtype) { q0 = round_to<XDT>(q0); q1 = round_to<XDT>(q1); }  // the quotient stays in x.dtype
        t[2 * j] = q0; t[2 * j + 1] = q1;
    }
    return fp4_word(t);
}

// lane = 4 consecutive units (32 elements = 64 B in, 16 B out); upg = units per scale group (2: group 16, 4: group 32)
template <int XDT, bool GLOBAL>
__global__ __launch_bounds__(kBlock) void fp4_quant_pack_kernel(const u32x4* __restrict__ in, const void* __restrict__ scale, int sdt,
                                                                const float* __restrict__ global_scale, u32x4* __restrict__ out, int64_t units,
                                                                int upg_shift) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g * 4 >= units) return;
    const int64_t left = units - g * 4;  // < 4 only in the last lane of a tensor whose unit count is not a multiple of 4
    const float gs = GLOBAL ? global_scale[0] : 1.0f;
    const bool in_dtype = !GLOBAL && sdt == XDT;
    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = load_rt(scale, sdt, (i < left ? g * 4 + i : g * 4) >> upg_shift);
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = i < left ? in[g * 4 + i] : u32x4{0, 0, 0, 0};
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        w[i] = fp4_quant_unit<XDT, GLOBAL>(ws, s[i], gs, in_dtype);
    }
    if (left >= 4) {
        stream_store16(out + g, u32x4{w[0], w[1], w[2], w[3]});
    } else {
        uint32_t* o = reinterpret_cast<uint32_t*>(out + g);
        for (int i = 0; i < left; ++i) o[i] = w[i];
    }
}

// float32 weights (round 6; rare — a float32 checkpoint — and declined before): one unit of 8 floats per lane, the quotient is the IEEE float32
// divide (T = float32 whatever the scale's dtype), then cast_to_fp4 by the hardware conversion as above
template <bool GLOBAL>
__global__ __launch_bounds__(kBlock) void fp4_quant_pack_f32_kernel(const u32x4* __restrict__ in, const void* __restrict__ scale, int sdt,
                                                                    const float* __restrict__ global_scale, uint32_t* __restrict__ out, int64_t units, int upg_shift) {
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (u >= units) return;
    const u32x4 a = in[2 * u], b = in[2 * u + 1];
    const float s = load_rt(scale, sdt, u >> upg_shift);
    const float s_eff = GLOBAL ? s / global_scale[0] : s;
    const uint32_t ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float q = bits_f(ws[k]) / s_eff;
        t[k] = q != q ? 0.0f : q;  // cast_to_fp4 of a NaN: no threshold compares true and `x < 0` is false — +0 (the hardware conversion would saturate it to 6)
    }
    out[u] = fp4_word(t);
}

// the common shape, lean: every lane owns 4 full units (tensor size % 32 == 0) and the scale dtype is a template
// parameter, so the lane's one (group 32) or two (group 16) scales are fetched by a single small load BEFORE the 64
// bytes of weights (same idea as w4_quant_pack_lean), and the fast / slow quotient choice is made once per lane.
// tensor.to(float8_e4m3fn) of one float as torch evaluates it (c10/util/Float8_e4m3fn.h, the conversion the reference's `scale.to(scale_dtype)` runs,
// nvfp4/base.py:96-100): round to nearest even, a magnitude that rounds beyond 448 — above the tie at 464 — or a NaN becomes the NaN code 0x7f, sign kept
__device__ __forceinline__ uint32_t f32_to_fp8_as_torch(float v) {
    const float a = __builtin_fabsf(v);
    const uint32_t sign = (f_bits(v) >> 24) & 0x80u;
    uint32_t b;
    if (!(a <= 464.0f)) b = 0x7fu;        // NaN, inf, overflow
    else if (a >= 448.0f) b = 0x7eu;      // [448, 464]: 448 (the tie goes to the even mantissa)
    else b = f2_to_fp8x2(a, 0.0f) & 0xffu;
    return b | sign;
}

// STORED (round 6): the lane also writes its groups' scales in their stored form — NVFP4: float8_e4m3fn bytes; MXFP4: the E8M0 codes of
// compress_mx_scale (mx_utils.py:18-31: 127 + floor(log2(scale)) through int32 to uint8, log2 in the scale's dtype), read from a 65536-entry table
// that the host builds by running that very expression over every 16-bit pattern — so the class call needs no tensor ops beside the launch
template <int XDT, int SDT, int GROUP, bool GLOBAL>
__device__ __forceinline__ void fp4_quant_pack_lean_lane(const u32x4* __restrict__ in, const void* __restrict__ scale, const float* __restrict__ global_scale,
                                                         u32x4* __restrict__ out, int64_t g, uint8_t* __restrict__ stored, const uint8_t* __restrict__ mx_lut) {
    constexpr int NS = 32 / GROUP;  // scales per lane
    float s[2];
    uint32_t sraw = 0;  // the 16-bit patterns of the lane's scales (SDT != F32)
    if constexpr (SDT == CT_F32) {
        if constexpr (NS == 2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 v = __builtin_nontemporal_load(reinterpret_cast<const f2*>(scale) + g);
            s[0] = v.x; s[1] = v.y;
        } else {
            s[0] = s[1] = __builtin_nontemporal_load(static_cast<const float*>(scale) + g);
        }
    } else {
        uint32_t b;
        if constexpr (NS == 2) b = __builtin_nontemporal_load(static_cast<const uint32_t*>(scale) + g);
        else b = __builtin_nontemporal_load(static_cast<const uint16_t*>(scale) + g);
        sraw = b;
        if constexpr (SDT == CT_BF16) { s[0] = bits_f(b << 16); s[1] = NS == 2 ? bits_f(b & 0xffff0000u) : s[0]; }
        else { s[0] = f16_bits_to_f(b & 0xffffu); s[1] = NS == 2 ? f16_bits_to_f(b >> 16) : s[0]; }
    }
    const float gs = GLOBAL ? global_scale[0] : 1.0f;
    if (stored) {  // kernel-uniform
        if constexpr (GROUP == 16) {
            reinterpret_cast<uint16_t*>(stored)[g] = (uint16_t)(f32_to_fp8_as_torch(s[0]) | (f32_to_fp8_as_torch(s[1]) << 8));
        } else if constexpr (SDT != CT_F32) {
            stored[g] = mx_lut[sraw & 0xffffu];
        }
    }
    asm volatile("" ::: "memory");  // keep the small loads ahead of the big ones
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = in[g * 4 + i];
    constexpr bool in_dtype = !GLOBAL && SDT == XDT;
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        w[i] = fp4_quant_unit<XDT, GLOBAL>(ws, s[NS == 2 ? i >> 1 : 0], gs, in_dtype);
    }
    stream_store16(out + g, u32x4{w[0], w[1], w[2], w[3]});
}

template <int XDT, int SDT, int GROUP, bool GLOBAL>
__global__ __launch_bounds__(kBlock) void fp4_quant_pack_lean_kernel(const u32x4* __restrict__ in, const void* __restrict__ scale,
                                                                     const float* __restrict__ global_scale, u32x4* __restrict__ out, int64_t lanes,
                                                                     uint8_t* __restrict__ stored, const uint8_t* __restrict__ mx_lut) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= lanes) return;
    fp4_quant_pack_lean_lane<XDT, SDT, GROUP, GLOBAL>(in, scale, global_scale, out, g, stored, mx_lut);
}

// ------------------------------------------------------------------------------------------------------------------
// Tables of FP4 tensors (round 6): ONE launch per direction for the modules of a checkpoint — the NVFP4 / MXFP4 counterpart of ct_quant_pack_batch /
// ct_q8_quant_batch (ModelCompressor's per-module loop, model_compressor.py:167-169,196-198, over NVFP4PackedCompressor / MXFP4PackedCompressor.compress /
// .decompress, nvfp4/base.py:68-139).  The table is `struct ct_w4_item` with the FP4 reading of its fields (include/ct_hip.h): zp = the tensor's global
// scale (one float32, NVFP4) or NULL (MXFP4), zp_packed = the stored-scale output (compress) resp. the bfloat16 scale output (decompress).  One launch per
// module left a Llama-3-8B-shaped NVFP4 tree at 0.58 of the HBM peak (ramp and tail of 224 launches on tensors of 8-117 MB) and a TinyLlama-shaped one
// bound by 7 us of host work per module.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ const ct_w4_item& fp4_batch_find(const ct_w4_item* __restrict__ items, int n, int64_t block) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= block) lo = mid; else hi = mid - 1;
    }
    return items[lo];
}

template <int XDT, int SDT, int GROUP>
__global__ __launch_bounds__(kBlock) void fp4_quant_pack_batch_kernel(const ct_w4_item* __restrict__ items, int n, const uint8_t* __restrict__ mx_lut) {
    const ct_w4_item& it = fp4_batch_find(items, n, blockIdx.x);
    const int64_t g = ((int64_t)blockIdx.x - it.first_block) * kBlock + threadIdx.x;
    if (g >= (it.units >> 2)) return;
    fp4_quant_pack_lean_lane<XDT, SDT, GROUP, GROUP == 16>(static_cast<const u32x4*>(it.src), it.scale, static_cast<const float*>(it.zp), static_cast<u32x4*>(it.dst), g,
                                                           static_cast<uint8_t*>(it.zp_packed), mx_lut);
}

// ------------------------------------------------------------------------------------------------------------------
// Round-to-nearest MXFP4 in ONE pass (the microscaling format gfx950's matrix cores consume): a lane owns exactly one
// MX group (32 elements = 64 B in flight), so the min-max observer needs no cross-lane step: local amax -> E8M0 scale
// (calculate_qparams' MX branch: compute_qparams_float) -> x / scale -> cast_to_fp4 -> 16 bytes of nibbles + one
// exponent byte.  2 + 0.5 + 1/32 B per element instead of (2 + 2/32) + (2 + 0.5 + 2/32); bit-identical to
// ct_minmax_qparams_float(kind 3) + ct_fp4_quant_pack + compress_mx_scale by construction (same helpers).
template <int XDT>
__global__ __launch_bounds__(kBlock) void rtn_mxfp4_kernel(const u32x4* __restrict__ in, int64_t lanes, u32x4* __restrict__ out, uint8_t* __restrict__ e8m0,
                                                           void* __restrict__ scale_out) {
    const int64_t l = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (l >= lanes) return;
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = in[l * 4 + i];
    uint32_t acc = 0;  // max |x| on the raw bit pairs (ct_minmax.h): the MX schemes are symmetric
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, r[i].x), r[i].y), r[i].z), r[i].w);
    const MinMax m = absmax_finish<XDT>(acc);
    const float s = compute_qparams_float<XDT>(m, QP_MXFP4, 1.0f);
    // compress_mx_scale: 127 + floor(log2(s)); s is a power of two (2^-127, the only subnormal one, is code 0), or inf / NaN
    const uint32_t ef = (f_bits(s) >> 23) & 0xffu;
    e8m0[l] = (uint8_t)((s != s) ? 0u : ef);  // inf: 255 = its exponent field; the int cast of NaN is 0 upstream
    if (scale_out) store1<XDT>(scale_out, l, s);
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        w[i] = fp4_quant_unit<XDT, false>(ws, s, 1.0f, true);
    }
    stream_store16(out + l, u32x4{w[0], w[1], w[2], w[3]});
}

// The NVFP4 counterpart: a lane owns two groups of 16; the global scale (generate_gparam: a tensor-wide amax, i.e. one
// read-only pass before this one) is an input.  Writes the nibbles, the float8_e4m3fn group scales (the stored form) and
// optionally the float32 scales calculate_qparams returns.
template <int XDT>
__global__ __launch_bounds__(kBlock) void rtn_nvfp4_kernel(const u32x4* __restrict__ in, int64_t lanes, const float* __restrict__ global_scale,
                                                           u32x4* __restrict__ out, uint16_t* __restrict__ scale_f8, float* __restrict__ scale_out) {
    const int64_t l = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (l >= lanes) return;
    const float gs = global_scale[0];
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = in[l * 4 + i];
    float s[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 2 * h; i < 2 * h + 2; ++i) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, r[i].x), r[i].y), r[i].z), r[i].w);
        const MinMax m = absmax_finish<XDT>(acc);
        s[h] = compute_qparams_float<XDT>(m, QP_NVFP4, gs);
    }
    scale_f8[l] = (uint16_t)f2_to_fp8x2(s[0], s[1]);  // scale.to(float8_e4m3fn): exact, the values are float8 already
    if (scale_out) { scale_out[2 * l] = s[0]; scale_out[2 * l + 1] = s[1]; }
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        w[i] = fp4_quant_unit<XDT, true>(ws, s[i >> 1], gs, false);
    }
    stream_store16(out + l, u32x4{w[0], w[1], w[2], w[3]});
}

// every mantissa of s in [1, 2) against every mantissa of x in [1, 2) (7 bits bf16, 10 bits fp16; `xbits` of them):
// the fast quotient must equal the IEEE quotient bit for bit.  With x < s the quotient lies in [0.5, 1), else [1, 2).
__global__ __launch_bounds__(kBlock) void selftest_fp4_div_kernel(int xbits, uint32_t m_lo, uint32_t m_hi, unsigned long long* mismatches) {
    unsigned long long local = 0;
    for (uint32_t m = m_lo + blockIdx.x * kBlock + threadIdx.x; m < m_hi; m += gridDim.x * kBlock) {
        const float s = bits_f(0x3f800000u | m);
        const float r = fp4_refined_rcp(s), lim = 8.0f * s;
        for (uint32_t xm = 0; xm < (1u << xbits); ++xm) {
            const float x = bits_f(0x3f800000u | (xm << (23 - xbits)));
            local += f_bits(fp4_fast_quotient(x, s, r, lim)) != f_bits(x / s) ? 1ull : 0ull;
        }
    }
    if (local) atomicAdd(mismatches, local);
}

__device__ __forceinline__ float decode_scale(const void* scale, int kind, int sdt, int64_t si) {
    if (kind == SC_F8E4M3) {
        const uint32_t b = static_cast<const uint8_t*>(scale)[si];
        return __builtin_amdgcn_cvt_f32_fp8((int)b, 0);  // OCP e4m3fn on gfx950
    }
    if (kind == SC_E8M0) {
        const uint32_t e = static_cast<const uint8_t*>(scale)[si];
        return e == 0 ? 0x1p-127f : bits_f(e << 23);  // 2^(e - 127)
    }
    return load_rt(scale, sdt, si);
}

// lane = UNROLL units one block apart (4 B in, 16 B out)
template <int ODT, int UNROLL, bool GLOBAL>
__device__ __forceinline__ void fp4_unpack_dequant_units(const uint32_t* __restrict__ in, const void* __restrict__ scale, int kind, int sdt, float gs, void* __restrict__ out,
                                                         int64_t units, int upg_shift, int64_t base, uint16_t* __restrict__ scale_out) {
    {
        uint32_t word[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u = base + (int64_t)i * kBlock;
            if (u < units) word[i] = in[u];
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u = base + (int64_t)i * kBlock;
            if (u >= units) continue;
            const float s = decode_scale(scale, kind, sdt, u >> upg_shift);
            // the decompressed scale the reference hands back beside the weight (nvfp4/base.py:133-137 `scale.to(bfloat16)`, mx_utils.py:34-44
            // `2.0 ** (code - 127)` as bfloat16): written by the lane that owns the group's first unit
            if (scale_out && (u & (((int64_t)1 << upg_shift) - 1)) == 0) scale_out[u >> upg_shift] = (uint16_t)f_to_bf16_bits(s);
            const float s_eff = GLOBAL ? s / gs : s;
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p;
                switch (j) {  // the byte select is an immediate
                    case 0: p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(word[i], 1.0f, 0); break;
                    case 1: p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(word[i], 1.0f, 1); break;
                    case 2: p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(word[i], 1.0f, 2); break;
                    default: p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(word[i], 1.0f, 3); break;
                }
                v[2 * j] = p.x * s_eff;
                v[2 * j + 1] = p.y * s_eff;
                if constexpr (ODT == CT_F16) { asm("" : "+v"(v[2 * j])); asm("" : "+v"(v[2 * j + 1])); }  // see mul_round_to
            }
            store8<ODT>(out, u * 8, v);  // RNE to the output dtype
        }
    }
}

template <int ODT, int UNROLL, bool GLOBAL>
__global__ __launch_bounds__(kBlock) void fp4_unpack_dequant_kernel(const uint32_t* __restrict__ in, const void* __restrict__ scale, int kind, int sdt,
                                                                    const float* __restrict__ global_scale, void* __restrict__ out, int64_t units,
                                                                    int upg_shift, int64_t stride, uint16_t* __restrict__ scale_out) {
    const float gs = GLOBAL ? global_scale[0] : 1.0f;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; base < units; base += stride)
        fp4_unpack_dequant_units<ODT, UNROLL, GLOBAL>(in, scale, kind, sdt, gs, out, units, upg_shift, base, scale_out);
}

// ---- the stand-alone primitives behind the reference's ImplBackend entry points --------------------------------
__device__ __forceinline__ void fp4_word_values(uint32_t w, float (&v)[8]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p;
    p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 0); v[0] = p.x; v[1] = p.y;
    p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 1); v[2] = p.x; v[3] = p.y;
    p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 2); v[4] = p.x; v[5] = p.y;
    p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(w, 1.0f, 3); v[6] = p.x; v[7] = p.y;
}

// MODE 0: cast_to_fp4 (x -> nearest E2M1 value, same dtype; |x| rounded, times sign(x): -0.0 in gives +0.0 out, a
//         negative that rounds to zero gives -0.0, NaN stays NaN)       fp4_utils.py:77-98
// MODE 1: pack_fp4_to_uint8 (E2M1-valued x -> one nibble each)           nvfp4/helpers.py:108-150
// one lane = one unit of 8 elements; the last, partial unit of a tensor whose element count is not a multiple of
// 8 is handled element-wise by the lane that owns it
template <int XDT, int MODE>
__global__ __launch_bounds__(kBlock) void fp4_prim_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t n) {
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i0 = u * 8;
    if (i0 >= n) return;
    const int live = n - i0 >= 8 ? 8 : (int)(n - i0);
    float v[8];
    if (live == 8) {
        load8<XDT>(x, i0, v);
    } else {
        for (int k = 0; k < 8; ++k) v[k] = k < live ? load_as_f<XDT>(x, i0 + k) : 0.0f;
    }
    if constexpr (MODE == 0) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[k] + 0.0f;
        float q[8];
        fp4_word_values(fp4_word_signbit(t), q);
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = v[k] != v[k] ? v[k] : q[k];
        if (live == 8) {
            store8<XDT>(out, i0, q);
        } else {
            for (int k = 0; k < live; ++k) store1<XDT>(out, i0 + k, q[k]);
        }
    } else {
        const uint32_t w = fp4_word_signbit(v);
        if (live == 8) {
            static_cast<uint32_t*>(out)[u] = w;
        } else {
            for (int k = 0; k < live / 2; ++k) static_cast<uint8_t*>(out)[u * 4 + k] = (uint8_t)(w >> (8 * k));
        }
    }
}

// unpack_fp4_from_uint8 (nvfp4/helpers.py:153-193): n elements from n / 2 bytes, code 8 decodes to -0.0
template <int ODT>
__global__ __launch_bounds__(kBlock) void fp4_unpack_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, int64_t n) {
    const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i0 = u * 8;
    if (i0 >= n) return;
    const int live = n - i0 >= 8 ? 8 : (int)(n - i0);
    uint32_t w = 0;
    if (live == 8) {
        w = reinterpret_cast<const uint32_t*>(in)[u];
    } else {
        for (int k = 0; k < live / 2; ++k) w |= (uint32_t)in[u * 4 + k] << (8 * k);
    }
    float v[8];
    fp4_word_values(w, v);
    if (live == 8) {
        store8<ODT>(out, i0, v);
    } else {
        for (int k = 0; k < live; ++k) store1<ODT>(out, i0 + k, v[k]);
    }
}

template <int ODT, int UNROLL>
__device__ __forceinline__ void fp4_unpack_dequant_nv_units(const uint32_t* __restrict__ in, const uint8_t* __restrict__ scale, const float* s_eff, void* __restrict__ out,
                                                            int64_t units, int64_t base, uint16_t* __restrict__ scale_out);

// nvfp4: there are only 256 possible scale bytes, so s_eff = fl32(float(fp8) / global) is tabulated once per block
// (one IEEE divide per thread) and every unit replaces decode + divide by one LDS read
template <int ODT, int UNROLL>
__global__ __launch_bounds__(kBlock) void fp4_unpack_dequant_nv_kernel(const uint32_t* __restrict__ in, const uint8_t* __restrict__ scale,
                                                                       const float* __restrict__ global_scale, void* __restrict__ out, int64_t units,
                                                                       int64_t stride, uint16_t* __restrict__ scale_out) {
    __shared__ float s_eff[256];
    static_assert(kBlock == 256, "one table entry per thread");
    s_eff[threadIdx.x] = __builtin_amdgcn_cvt_f32_fp8((int)threadIdx.x, 0) / global_scale[0];
    __syncthreads();
    for (int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; base < units; base += stride)
        fp4_unpack_dequant_nv_units<ODT, UNROLL>(in, scale, s_eff, out, units, base, scale_out);
}

template <int ODT, int UNROLL>
__device__ __forceinline__ void fp4_unpack_dequant_nv_units(const uint32_t* __restrict__ in, const uint8_t* __restrict__ scale, const float* s_eff, void* __restrict__ out,
                                                            int64_t units, int64_t base, uint16_t* __restrict__ scale_out) {
    {
        uint32_t word[UNROLL];
        uint32_t sb[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u = base + (int64_t)i * kBlock;
            if (u < units) { word[i] = in[u]; sb[i] = scale[u >> 1]; }
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u = base + (int64_t)i * kBlock;
            if (u >= units) continue;
            const float s = s_eff[sb[i]];
            if (scale_out && (u & 1) == 0) scale_out[u >> 1] = (uint16_t)f_to_bf16_bits(__builtin_amdgcn_cvt_f32_fp8((int)sb[i], 0));  // scale.to(bfloat16): exact
            float v[8];
            fp4_word_values(word[i], v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] *= s;
                if constexpr (ODT == CT_F16) asm("" : "+v"(v[k]));  // see mul_round_to
            }
            store8<ODT>(out, u * 8, v);
        }
    }
}

constexpr int kFp4BatchUnroll = 2;
constexpr int kFp4BatchIter = 2;  // chunks of kBlock * kFp4BatchUnroll units per workgroup (as the W4 table: the table search is paid once per two chunks)

// decompress of a table: NVFP4 (NV = true: float8 scale bytes under the item's global scale, the 256-entry s_eff table per workgroup) or MXFP4 (E8M0 codes)
template <int ODT, bool NV>
__global__ __launch_bounds__(kBlock) void fp4_unpack_dequant_batch_kernel(const ct_w4_item* __restrict__ items, int n) {
    __shared__ float s_eff[256];
    const ct_w4_item& it = fp4_batch_find(items, n, blockIdx.x);
    const int64_t units = it.units;
    const int64_t first = ((int64_t)blockIdx.x - it.first_block) * kBlock * kFp4BatchUnroll * kFp4BatchIter;
    if constexpr (NV) {
        s_eff[threadIdx.x] = __builtin_amdgcn_cvt_f32_fp8((int)threadIdx.x, 0) / static_cast<const float*>(it.zp)[0];
        __syncthreads();
    }
#pragma unroll 1
    for (int c = 0; c < kFp4BatchIter; ++c) {
        const int64_t base = first + (int64_t)c * kBlock * kFp4BatchUnroll + threadIdx.x;
        if (base >= units) break;
        if constexpr (NV) fp4_unpack_dequant_nv_units<ODT, kFp4BatchUnroll>(static_cast<const uint32_t*>(it.src), static_cast<const uint8_t*>(it.scale), s_eff, it.dst, units, base,
                                                                            static_cast<uint16_t*>(it.zp_packed));
        else fp4_unpack_dequant_units<ODT, kFp4BatchUnroll, false>(static_cast<const uint32_t*>(it.src), it.scale, SC_E8M0, -1, 1.0f, it.dst, units, 2, base,
                                                                   static_cast<uint16_t*>(it.zp_packed));
    }
}

// the MX scale tensors by themselves (mx_utils.py:18-44; the MXFP8 codec and upstream's helpers call them): one launch each instead of four tensor ops
__global__ __launch_bounds__(kBlock) void mx_scale_compress_kernel(const uint16_t* __restrict__ scale_bits, int64_t n, const uint8_t* __restrict__ table,
                                                                   uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = table[scale_bits[i]];
}
__global__ __launch_bounds__(kBlock) void mx_scale_decompress_kernel(const uint8_t* __restrict__ codes, int64_t n, uint16_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const uint32_t e = codes[i];
        out[i] = (uint16_t)f_to_bf16_bits(e == 0 ? 0x1p-127f : bits_f(e << 23));  // 2 ** (e - 127) as bfloat16; e = 255: 2^128 = inf
    }
}

// a table of MX scale tensors (MXFP8 modules: ModelCompressor's loop over MXFP8QuantizationCompressor.compress / .decompress, mxfp8/base.py:47-101): items' src /
// dst and `rows` = the element count; COMPRESS: 16-bit scales -> E8M0 codes through the code table, else codes -> bfloat16 powers of two
template <bool COMPRESS>
__global__ __launch_bounds__(kBlock) void mx_scale_batch_kernel(const ct_w4_item* __restrict__ items, int n, const uint8_t* __restrict__ table) {
    // eight scales per lane (16 B of 16-bit scales <-> 8 B of codes): one scale per lane spent a table search per 256 scales — 0.4 ms per direction on an
    // 8B-shaped MXFP8 tree's 109 M scales
    const ct_w4_item& it = fp4_batch_find(items, n, blockIdx.x);
    const int64_t e0 = (((int64_t)blockIdx.x - it.first_block) * kBlock + threadIdx.x) * 8;
    if (e0 >= it.rows) return;
    const bool vec = e0 + 8 <= it.rows && ((reinterpret_cast<uintptr_t>(it.src) | reinterpret_cast<uintptr_t>(it.dst)) & 15u) == 0;
    if constexpr (COMPRESS) {
        const uint16_t* src = static_cast<const uint16_t*>(it.src);
        uint8_t* dst = static_cast<uint8_t*>(it.dst);
        if (vec) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + e0);
            const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                lo |= ((uint32_t)table[ws[k] & 0xffffu] | ((uint32_t)table[ws[k] >> 16] << 8)) << (16 * k);
                hi |= ((uint32_t)table[ws[2 + k] & 0xffffu] | ((uint32_t)table[ws[2 + k] >> 16] << 8)) << (16 * k);
            }
            *reinterpret_cast<u32x2*>(dst + e0) = u32x2{lo, hi};
        } else {
            for (int64_t i = e0; i < e0 + 8 && i < it.rows; ++i) dst[i] = table[src[i]];
        }
    } else {
        const uint8_t* src = static_cast<const uint8_t*>(it.src);
        uint16_t* dst = static_cast<uint16_t*>(it.dst);
        auto decode = [](uint32_t e) { return (uint32_t)f_to_bf16_bits(e == 0 ? 0x1p-127f : bits_f(e << 23)) & 0xffffu; };  // 2 ** (e - 127) as bfloat16
        if (vec) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(src + e0);
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                w[k] = decode((v.x >> (16 * k)) & 0xffu) | (decode((v.x >> (16 * k + 8)) & 0xffu) << 16);
                w[2 + k] = decode((v.y >> (16 * k)) & 0xffu) | (decode((v.y >> (16 * k + 8)) & 0xffu) << 16);
            }
            *reinterpret_cast<u32x4*>(dst + e0) = u32x4{w[0], w[1], w[2], w[3]};
        } else {
            for (int64_t i = e0; i < e0 + 8 && i < it.rows; ++i) dst[i] = (uint16_t)decode(src[i]);
        }
    }
}

}  // namespace ct

using namespace ct;

extern "C" {

static int fp4_quant_pack_impl(const void* x, int xdt, const void* scale, int sdt, const float* global_scale, int64_t rows, int64_t cols, int64_t group,
                              uint8_t* packed, uint8_t* stored, const uint8_t* mx_lut, ct_stream_t stream) {
    CT_REQUIRE(is_float_dt(xdt), "FP4 compression expects float weights, got dtype %d", xdt);
    CT_REQUIRE(is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    CT_REQUIRE(group == 16 || group == 32, "FP4 group size must be 16 (nvfp4) or 32 (mxfp4), got %lld", (long long)group);
    CT_REQUIRE(cols % 2 == 0, "tensor must have an even number of columns for nvfp4 compression");
    CT_REQUIRE(cols % group == 0, "columns (%lld) must be a multiple of the group size %lld", (long long)cols, (long long)group);
    CT_REQUIRE(aligned16(x) && aligned16(packed), "buffers must be 16-byte aligned");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t units = rows * (cols / 8);  // rows are contiguous and cols % 8 == 0: one flat unit stream
    if (xdt == CT_F32) {
        if (stored) CT_UNSUPPORTED("ct_fp4_quant_pack_stored: float32 weights take ct_fp4_quant_pack and the host's scale conversion");
        CT_REQUIRE(cdiv64(units, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
        dim3 g32((unsigned)cdiv64(units, kBlock));
        if (global_scale) hipLaunchKernelGGL((fp4_quant_pack_f32_kernel<true>), g32, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), scale, sdt, global_scale,
                                             reinterpret_cast<uint32_t*>(packed), units, group == 16 ? 1 : 2);
        else hipLaunchKernelGGL((fp4_quant_pack_f32_kernel<false>), g32, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), scale, sdt, global_scale,
                                reinterpret_cast<uint32_t*>(packed), units, group == 16 ? 1 : 2);
        CT_LAUNCH_CHECK("ct_fp4_quant_pack[f32]");
    }
    const int64_t lanes = cdiv64(units, 4);
    const int shift = group == 16 ? 1 : 2;
    CT_REQUIRE(cdiv64(lanes, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
    dim3 grid((unsigned)cdiv64(lanes, kBlock));
    const bool lean = units % 4 == 0 && (reinterpret_cast<uintptr_t>(scale) & 7u) == 0;
    if (stored) {  // the stored scales ride the lean kernel only; NVFP4 any float scale, MXFP4 16-bit scales through the caller's table
        if (!lean || (group == 32 && (sdt == CT_F32 || mx_lut == nullptr)) || (group == 16 && (reinterpret_cast<uintptr_t>(stored) & 1u)))
            CT_UNSUPPORTED("ct_fp4_quant_pack_stored: this layout takes ct_fp4_quant_pack and the host's scale conversion");
    }
    if (lean) {
        dim3 lg((unsigned)cdiv64(lanes, kBlock));
#define CT_FP4L(XD, SD, G, GL) hipLaunchKernelGGL((fp4_quant_pack_lean_kernel<XD, SD, G, GL>), lg, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), scale, \
                                                  global_scale, reinterpret_cast<u32x4*>(packed), lanes, stored, mx_lut)
#define CT_FP4L_G(XD, SD) do { if (group == 16) { if (global_scale) CT_FP4L(XD, SD, 16, true); else CT_FP4L(XD, SD, 16, false); } \
                               else { if (global_scale) CT_FP4L(XD, SD, 32, true); else CT_FP4L(XD, SD, 32, false); } } while (0)
#define CT_FP4L_S(XD) do { if (sdt == CT_F32) CT_FP4L_G(XD, CT_F32); else if (sdt == CT_BF16) CT_FP4L_G(XD, CT_BF16); else CT_FP4L_G(XD, CT_F16); } while (0)
        if (xdt == CT_BF16) CT_FP4L_S(CT_BF16); else CT_FP4L_S(CT_F16);
#undef CT_FP4L_S
#undef CT_FP4L_G
#undef CT_FP4L
        CT_LAUNCH_CHECK("ct_fp4_quant_pack[lean]");
    }
#define CT_FP4Q(DT, GL) hipLaunchKernelGGL((fp4_quant_pack_kernel<DT, GL>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), scale, sdt, \
                                          global_scale, reinterpret_cast<u32x4*>(packed), units, shift)
    if (xdt == CT_BF16) { if (global_scale) CT_FP4Q(CT_BF16, true); else CT_FP4Q(CT_BF16, false); }
    else { if (global_scale) CT_FP4Q(CT_F16, true); else CT_FP4Q(CT_F16, false); }
#undef CT_FP4Q
    CT_LAUNCH_CHECK("ct_fp4_quant_pack");
}

int ct_fp4_quant_pack(const void* x, int xdt, const void* scale, int sdt, const float* global_scale, int64_t rows, int64_t cols, int64_t group,
                      uint8_t* packed, ct_stream_t stream) {
    return fp4_quant_pack_impl(x, xdt, scale, sdt, global_scale, rows, cols, group, packed, nullptr, nullptr, stream);
}

int ct_fp4_quant_pack_stored(const void* x, int xdt, const void* scale, int sdt, const float* global_scale, int64_t rows, int64_t cols, int64_t group,
                             uint8_t* packed, uint8_t* scale_stored, const uint8_t* mx_code_table, ct_stream_t stream) {
    CT_REQUIRE(scale_stored != nullptr, "ct_fp4_quant_pack_stored needs the stored-scale output");
    return fp4_quant_pack_impl(x, xdt, scale, sdt, global_scale, rows, cols, group, packed, scale_stored, mx_code_table, stream);
}

static int fp4_unpack_dequant_impl(const uint8_t* packed, int64_t rows, int64_t cols, const void* scale, int scale_kind, int sdt, const float* global_scale,
                                   int64_t group, void* out, int odt, uint16_t* scale_out, ct_stream_t stream) {
    CT_REQUIRE(odt == CT_BF16 || odt == CT_F16, "FP4 decompression writes 16-bit floats, got dtype %d", odt);
    CT_REQUIRE(scale_kind >= SC_PLAIN && scale_kind <= SC_E8M0, "bad scale kind %d", scale_kind);
    CT_REQUIRE(scale_kind != SC_PLAIN || is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    CT_REQUIRE(group == 16 || group == 32, "FP4 group size must be 16 (nvfp4) or 32 (mxfp4), got %lld", (long long)group);
    CT_REQUIRE(cols % group == 0, "columns (%lld) must be a multiple of the group size %lld", (long long)cols, (long long)group);
    CT_REQUIRE(aligned16(out) && (reinterpret_cast<uintptr_t>(packed) & 3u) == 0, "misaligned buffers");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t units = rows * (cols / 8);
    const int shift = group == 16 ? 1 : 2;
    constexpr int U = 2;
    CT_REQUIRE(cdiv64(units, (int64_t)kBlock * U) < ((int64_t)1 << 31), "tensor too large for one launch");
    dim3 grid((unsigned)cdiv64(units, (int64_t)kBlock * U));
    const int64_t stride = (int64_t)1 << 40;  // one trip; the loop form schedules better (see ct_quant.hip)
    if (scale_kind == SC_F8E4M3 && global_scale && group == 16) {
        if (odt == CT_BF16) hipLaunchKernelGGL((fp4_unpack_dequant_nv_kernel<CT_BF16, U>), grid, dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const uint32_t*>(packed),
                                               static_cast<const uint8_t*>(scale), global_scale, out, units, stride, scale_out);
        else hipLaunchKernelGGL((fp4_unpack_dequant_nv_kernel<CT_F16, U>), grid, dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const uint32_t*>(packed),
                                static_cast<const uint8_t*>(scale), global_scale, out, units, stride, scale_out);
        CT_LAUNCH_CHECK("ct_fp4_unpack_dequant[nvfp4]");
    }
#define CT_FP4D(DT, GL) hipLaunchKernelGGL((fp4_unpack_dequant_kernel<DT, U, GL>), grid, dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const uint32_t*>(packed), \
                                          scale, scale_kind, sdt, global_scale, out, units, shift, stride, scale_out)
    if (odt == CT_BF16) { if (global_scale) CT_FP4D(CT_BF16, true); else CT_FP4D(CT_BF16, false); }
    else { if (global_scale) CT_FP4D(CT_F16, true); else CT_FP4D(CT_F16, false); }
#undef CT_FP4D
    CT_LAUNCH_CHECK("ct_fp4_unpack_dequant");
}

int ct_fp4_unpack_dequant(const uint8_t* packed, int64_t rows, int64_t cols, const void* scale, int scale_kind, int sdt, const float* global_scale,
                          int64_t group, void* out, int odt, ct_stream_t stream) {
    return fp4_unpack_dequant_impl(packed, rows, cols, scale, scale_kind, sdt, global_scale, group, out, odt, nullptr, stream);
}

int ct_fp4_unpack_dequant_scale(const uint8_t* packed, int64_t rows, int64_t cols, const void* scale, int scale_kind, int sdt, const float* global_scale,
                                int64_t group, void* out, int odt, void* scale_bf16_out, ct_stream_t stream) {
    CT_REQUIRE(scale_bf16_out != nullptr && (reinterpret_cast<uintptr_t>(scale_bf16_out) & 1u) == 0, "ct_fp4_unpack_dequant_scale needs the (2-byte aligned) scale output");
    return fp4_unpack_dequant_impl(packed, rows, cols, scale, scale_kind, sdt, global_scale, group, out, odt, static_cast<uint16_t*>(scale_bf16_out), stream);
}

int64_t ct_fp4_batch_plan(ct_w4_item* items, int n, int direction) {
    if (!items || n < 0 || (direction != 0 && direction != 1)) {
        set_error("ct_fp4_batch_plan: bad arguments");
        return -1;
    }
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        ct_w4_item& it = items[i];
        const int64_t g = it.group;
        // compress: the lean kernel's layout (every lane four full units, its scales one aligned small load); decompress: 4-byte aligned words, 16-byte aligned output
        bool ok = it.rows > 0 && it.cols > 0 && (g == 16 || g == 32) && it.cols % g == 0 && (it.rows * it.cols) % 32 == 0 && it.src && it.scale && it.dst && it.zp_packed &&
                  (it.zp != nullptr) == (g == 16);
        if (ok && direction == 0)
            ok = aligned16(it.src) && aligned16(it.dst) && (reinterpret_cast<uintptr_t>(it.scale) & 7u) == 0 && (reinterpret_cast<uintptr_t>(it.zp_packed) & 1u) == 0;
        if (ok && direction == 1)
            ok = (reinterpret_cast<uintptr_t>(it.src) & 3u) == 0 && aligned16(it.dst) && (reinterpret_cast<uintptr_t>(it.zp_packed) & 1u) == 0;
        if (!ok) {
            set_error("ct_fp4_batch_plan: item %d (rows %lld, cols %lld, group %lld) is not eligible for the batched FP4 path (needs group 16 with a global scale or group 32 "
                      "without one, cols %% group == 0, rows * cols %% 32 == 0, the scale outputs, aligned buffers)", i, (long long)it.rows, (long long)it.cols, (long long)g);
            return -1;
        }
        it.units = it.rows * (it.cols / 8);
        it.upg = (int32_t)(g / 8);
        it.upg_shift = g == 16 ? 1 : 2;
        it.first_block = blocks;
        it.main_blocks = direction == 0 ? cdiv64(it.units / 4, kBlock) : cdiv64(it.units, (int64_t)kBlock * kFp4BatchUnroll * kFp4BatchIter);
        it.g_magic = 0;
        it.g_shift = 0;
        blocks += it.main_blocks;
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_fp4_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    return blocks;
}

int ct_fp4_quant_pack_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int xdt, int sdt, int group, const uint8_t* mx_code_table, ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "batched FP4 path: 16-bit weights only, got dtype %d", xdt);
    CT_REQUIRE(is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(group == 16 || group == 32, "FP4 group size must be 16 (nvfp4) or 32 (mxfp4), got %d", group);
    CT_REQUIRE(group == 16 || (sdt != CT_F32 && mx_code_table != nullptr), "MXFP4 tables take 16-bit scales and the E8M0 code table");
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    const dim3 grid((unsigned)total_blocks);
#define CT_FP4B(XD, SD, G) hipLaunchKernelGGL((fp4_quant_pack_batch_kernel<XD, SD, G>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, mx_code_table)
#define CT_FP4B_S(XD) do { if (group == 16) { if (sdt == CT_F32) CT_FP4B(XD, CT_F32, 16); else if (sdt == CT_BF16) CT_FP4B(XD, CT_BF16, 16); else CT_FP4B(XD, CT_F16, 16); } \
                           else { if (sdt == CT_BF16) CT_FP4B(XD, CT_BF16, 32); else CT_FP4B(XD, CT_F16, 32); } } while (0)
    if (xdt == CT_BF16) CT_FP4B_S(CT_BF16); else CT_FP4B_S(CT_F16);
#undef CT_FP4B_S
#undef CT_FP4B
    CT_LAUNCH_CHECK("ct_fp4_quant_pack_batch");
}

int ct_fp4_unpack_dequant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int group, int odt, ct_stream_t stream) {
    CT_REQUIRE(odt == CT_BF16 || odt == CT_F16, "FP4 decompression writes 16-bit floats, got dtype %d", odt);
    CT_REQUIRE(group == 16 || group == 32, "FP4 group size must be 16 (nvfp4) or 32 (mxfp4), got %d", group);
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    const dim3 grid((unsigned)total_blocks);
    if (odt == CT_BF16) {
        if (group == 16) hipLaunchKernelGGL((fp4_unpack_dequant_batch_kernel<CT_BF16, true>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n);
        else hipLaunchKernelGGL((fp4_unpack_dequant_batch_kernel<CT_BF16, false>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n);
    } else {
        if (group == 16) hipLaunchKernelGGL((fp4_unpack_dequant_batch_kernel<CT_F16, true>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n);
        else hipLaunchKernelGGL((fp4_unpack_dequant_batch_kernel<CT_F16, false>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n);
    }
    CT_LAUNCH_CHECK("ct_fp4_unpack_dequant_batch");
}

int64_t ct_mx_scale_batch_plan(ct_w4_item* items, int n) {
    if (!items || n < 0) {
        set_error("ct_mx_scale_batch_plan: bad arguments");
        return -1;
    }
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        ct_w4_item& it = items[i];
        if (!(it.rows > 0 && it.src && it.dst && (reinterpret_cast<uintptr_t>(it.src) & 1u) == 0 && (reinterpret_cast<uintptr_t>(it.dst) & 1u) == 0)) {
            set_error("ct_mx_scale_batch_plan: item %d needs src, dst (2-byte aligned) and rows = its element count > 0", i);
            return -1;
        }
        it.first_block = blocks;
        blocks += cdiv64(cdiv64(it.rows, 8), kBlock);  // eight scales per lane
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_mx_scale_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    return blocks;
}

int ct_mx_scale_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int direction, int sdt, const uint8_t* code_table, ct_stream_t stream) {
    CT_REQUIRE(direction == 0 || direction == 1, "direction must be 0 (compress) or 1 (decompress)");
    CT_REQUIRE(direction == 1 || ((sdt == CT_BF16 || sdt == CT_F16) && code_table != nullptr), "compress takes 16-bit scales and their code table");
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    if (direction == 0) hipLaunchKernelGGL((mx_scale_batch_kernel<true>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n, code_table);
    else hipLaunchKernelGGL((mx_scale_batch_kernel<false>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n, code_table);
    CT_LAUNCH_CHECK("ct_mx_scale_batch");
}

int ct_mx_scale_compress(const void* scale, int sdt, int64_t n, const uint8_t* code_table, uint8_t* codes_out, ct_stream_t stream) {
    CT_REQUIRE(sdt == CT_BF16 || sdt == CT_F16, "ct_mx_scale_compress takes 16-bit scales (the table has one entry per 16-bit pattern), got dtype %d", sdt);
    CT_REQUIRE(n >= 0 && code_table != nullptr && (reinterpret_cast<uintptr_t>(scale) & 1u) == 0, "bad arguments");
    if (n == 0) return CT_OK;
    CT_REQUIRE(cdiv64(n, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
    hipLaunchKernelGGL(mx_scale_compress_kernel, dim3((unsigned)cdiv64(n, kBlock)), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(scale), n, code_table, codes_out);
    CT_LAUNCH_CHECK("ct_mx_scale_compress");
}

int ct_mx_scale_decompress(const uint8_t* codes, int64_t n, void* scale_bf16_out, ct_stream_t stream) {
    CT_REQUIRE(n >= 0 && (reinterpret_cast<uintptr_t>(scale_bf16_out) & 1u) == 0, "bad arguments");
    if (n == 0) return CT_OK;
    CT_REQUIRE(cdiv64(n, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
    hipLaunchKernelGGL(mx_scale_decompress_kernel, dim3((unsigned)cdiv64(n, kBlock)), dim3(kBlock), 0, as_stream(stream), codes, n, static_cast<uint16_t*>(scale_bf16_out));
    CT_LAUNCH_CHECK("ct_mx_scale_decompress");
}

static int fp4_prim(const void* x, int xdt, void* out, int64_t n, int mode, ct_stream_t stream, const char* what) {
    CT_REQUIRE(is_float_dt(xdt), "%s: dtype code %d is not a float type", what, xdt);
    CT_REQUIRE(n >= 0, "%s: negative size", what);
    CT_REQUIRE(aligned16(x) && aligned16(out), "%s: buffers must be 16-byte aligned", what);
    if (n == 0) return CT_OK;
    const int64_t units = cdiv64(n, 8);
    CT_REQUIRE(cdiv64(units, kBlock) < ((int64_t)1 << 31), "%s: tensor too large for one launch", what);
    dim3 grid((unsigned)cdiv64(units, kBlock));
#define CT_FP4P(DT) do { if (mode == 0) hipLaunchKernelGGL((fp4_prim_kernel<DT, 0>), grid, dim3(kBlock), 0, as_stream(stream), x, out, n); \
                         else hipLaunchKernelGGL((fp4_prim_kernel<DT, 1>), grid, dim3(kBlock), 0, as_stream(stream), x, out, n); } while (0)
    switch (xdt) {
        case CT_BF16: CT_FP4P(CT_BF16); break;
        case CT_F16: CT_FP4P(CT_F16); break;
        default: CT_FP4P(CT_F32); break;
    }
#undef CT_FP4P
    CT_LAUNCH_CHECK(what);
}

int ct_fp4_cast(const void* x, int xdt, void* out, int64_t n, ct_stream_t stream) { return fp4_prim(x, xdt, out, n, 0, stream, "ct_fp4_cast"); }

int ct_fp4_pack(const void* x, int xdt, uint8_t* packed, int64_t n, ct_stream_t stream) {
    CT_REQUIRE(n % 2 == 0, "tensor must have an even number of columns for nvfp4 compression");
    return fp4_prim(x, xdt, packed, n, 1, stream, "ct_fp4_pack");
}

int ct_fp4_unpack(const uint8_t* packed, int64_t n, void* out, int odt, ct_stream_t stream) {
    CT_REQUIRE(is_float_dt(odt), "ct_fp4_unpack: dtype code %d is not a float type", odt);
    CT_REQUIRE(n >= 0 && n % 2 == 0, "ct_fp4_unpack: element count must be even and non-negative");
    CT_REQUIRE(aligned16(packed) && aligned16(out), "ct_fp4_unpack: buffers must be 16-byte aligned");
    if (n == 0) return CT_OK;
    const int64_t units = cdiv64(n, 8);
    CT_REQUIRE(cdiv64(units, kBlock) < ((int64_t)1 << 31), "ct_fp4_unpack: tensor too large for one launch");
    dim3 grid((unsigned)cdiv64(units, kBlock));
    switch (odt) {
        case CT_BF16: hipLaunchKernelGGL((fp4_unpack_kernel<CT_BF16>), grid, dim3(kBlock), 0, as_stream(stream), packed, out, n); break;
        case CT_F16: hipLaunchKernelGGL((fp4_unpack_kernel<CT_F16>), grid, dim3(kBlock), 0, as_stream(stream), packed, out, n); break;
        default: hipLaunchKernelGGL((fp4_unpack_kernel<CT_F32>), grid, dim3(kBlock), 0, as_stream(stream), packed, out, n); break;
    }
    CT_LAUNCH_CHECK("ct_fp4_unpack");
}

int ct_selftest_fp4_div(int xdt, uint32_t m_lo, uint32_t m_hi, unsigned long long* mismatches, ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "ct_selftest_fp4_div: x dtype must be bf16 or fp16");
    CT_REQUIRE(m_lo <= m_hi && m_hi <= (1u << 23), "ct_selftest_fp4_div: mantissa range must lie in [0, 2^23]");
    hipError_t e = hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), as_stream(stream));
    if (e != hipSuccess) return hip_check(e, "ct_selftest_fp4_div memset");
    if (m_lo == m_hi) return CT_OK;
    const int64_t n = m_hi - m_lo;
    int64_t g = cdiv64(n, kBlock);
    if (g > kCUs * 64) g = kCUs * 64;
    hipLaunchKernelGGL(selftest_fp4_div_kernel, dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), xdt == CT_BF16 ? 7 : 10, m_lo, m_hi, mismatches);
    CT_LAUNCH_CHECK("ct_selftest_fp4_div");
}

int ct_rtn_mxfp4_quant_pack(const void* x, int xdt, int64_t rows, int64_t cols, uint8_t* packed, uint8_t* scale_e8m0, void* scale_out, ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "the one-pass MXFP4 compress takes 16-bit float weights, got dtype %d", xdt);
    CT_REQUIRE(rows >= 0 && cols >= 0 && cols % 32 == 0, "columns (%lld) must be a multiple of the MX group size 32", (long long)cols);
    CT_REQUIRE(aligned16(x) && aligned16(packed) && scale_e8m0 != nullptr, "buffers must be 16-byte aligned and the scale output present");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t lanes = rows * (cols / 32);
    CT_REQUIRE(cdiv64(lanes, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
    dim3 grid((unsigned)cdiv64(lanes, kBlock));
    if (xdt == CT_BF16) hipLaunchKernelGGL((rtn_mxfp4_kernel<CT_BF16>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), lanes,
                                           reinterpret_cast<u32x4*>(packed), scale_e8m0, scale_out);
    else hipLaunchKernelGGL((rtn_mxfp4_kernel<CT_F16>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), lanes,
                            reinterpret_cast<u32x4*>(packed), scale_e8m0, scale_out);
    CT_LAUNCH_CHECK("ct_rtn_mxfp4_quant_pack");
}

int ct_rtn_nvfp4_quant_pack(const void* x, int xdt, int64_t rows, int64_t cols, const float* global_scale, uint8_t* packed, uint8_t* scale_f8,
                            float* scale_out, ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "the one-pass NVFP4 compress takes 16-bit float weights, got dtype %d", xdt);
    CT_REQUIRE(rows >= 0 && cols >= 0 && cols % 32 == 0, "columns (%lld) must be a multiple of 32 (two groups of 16 per lane)", (long long)cols);
    CT_REQUIRE(global_scale != nullptr && scale_f8 != nullptr, "the global scale and the scale output are required");
    CT_REQUIRE(aligned16(x) && aligned16(packed) && (reinterpret_cast<uintptr_t>(scale_f8) & 1u) == 0, "misaligned buffers");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t lanes = rows * (cols / 32);
    CT_REQUIRE(cdiv64(lanes, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
    dim3 grid((unsigned)cdiv64(lanes, kBlock));
    if (xdt == CT_BF16) hipLaunchKernelGGL((rtn_nvfp4_kernel<CT_BF16>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), lanes, global_scale,
                                           reinterpret_cast<u32x4*>(packed), reinterpret_cast<uint16_t*>(scale_f8), scale_out);
    else hipLaunchKernelGGL((rtn_nvfp4_kernel<CT_F16>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), lanes, global_scale,
                            reinterpret_cast<u32x4*>(packed), reinterpret_cast<uint16_t*>(scale_f8), scale_out);
    CT_LAUNCH_CHECK("ct_rtn_nvfp4_quant_pack");
}

}  // extern "C"
