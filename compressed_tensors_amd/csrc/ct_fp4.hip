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

// lane = 4 consecutive units (32 elements = 64 B in, 16 B out); upg = units per scale group (2: group 16, 4: group 32)
template <int XDT, bool GLOBAL>
__global__ __launch_bounds__(kBlock) void fp4_quant_pack_kernel(const u32x4* __restrict__ in, const void* __restrict__ scale, int sdt,
                                                                const float* __restrict__ global_scale, u32x4* __restrict__ out, int64_t units,
                                                                int upg_shift) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g * 4 >= units) return;
    const int64_t left = units - g * 4;  // < 4 only in the last lane of a tensor whose unit count is not a multiple of 4
    const float gs = GLOBAL ? global_scale[0] : 1.0f;
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = i < left ? in[g * 4 + i] : u32x4{0, 0, 0, 0};
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t si = (i < left ? g * 4 + i : g * 4) >> upg_shift;
        const float s = load_rt(scale, sdt, si);
        const float s_eff = GLOBAL ? s / gs : s;  // fl32(scale / global_scale)
        const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        float t[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x0, x1;
            if constexpr (XDT == CT_BF16) { x0 = bits_f(ws[j] << 16); x1 = bits_f(ws[j] & 0xffff0000u); }
            else { x0 = f16_bits_to_f(ws[j] & 0xffffu); x1 = f16_bits_to_f(ws[j] >> 16); }
            float q0 = x0 / s_eff, q1 = x1 / s_eff;
            if (!GLOBAL && sdt == XDT) { q0 = round_to<XDT>(q0); q1 = round_to<XDT>(q1); }  // the quotient stays in x.dtype
            t[2 * j] = q0; t[2 * j + 1] = q1;
        }
        w[i] = fp4_word(t);
    }
    if (left >= 4) {
        stream_store16(out + g, u32x4{w[0], w[1], w[2], w[3]});
    } else {
        uint32_t* o = reinterpret_cast<uint32_t*>(out + g);
        for (int i = 0; i < left; ++i) o[i] = w[i];
    }
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
__global__ __launch_bounds__(kBlock) void fp4_unpack_dequant_kernel(const uint32_t* __restrict__ in, const void* __restrict__ scale, int kind, int sdt,
                                                                    const float* __restrict__ global_scale, void* __restrict__ out, int64_t units,
                                                                    int upg_shift, int64_t stride) {
    const float gs = GLOBAL ? global_scale[0] : 1.0f;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; base < units; base += stride) {
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
            }
            store8<ODT>(out, u * 8, v);  // RNE to the output dtype
        }
    }
}

}  // namespace ct

using namespace ct;

extern "C" {

int ct_fp4_quant_pack(const void* x, int xdt, const void* scale, int sdt, const float* global_scale, int64_t rows, int64_t cols, int64_t group,
                      uint8_t* packed, ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "FP4 compression expects 16-bit float weights, got dtype %d", xdt);
    CT_REQUIRE(is_float_dt(sdt), "scale dtype code %d is not a float type", sdt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    CT_REQUIRE(group == 16 || group == 32, "FP4 group size must be 16 (nvfp4) or 32 (mxfp4), got %lld", (long long)group);
    CT_REQUIRE(cols % 2 == 0, "tensor must have an even number of columns for nvfp4 compression");
    CT_REQUIRE(cols % group == 0, "columns (%lld) must be a multiple of the group size %lld", (long long)cols, (long long)group);
    CT_REQUIRE(aligned16(x) && aligned16(packed), "buffers must be 16-byte aligned");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t units = rows * (cols / 8);  // rows are contiguous and cols % 8 == 0: one flat unit stream
    const int64_t lanes = cdiv64(units, 4);
    const int shift = group == 16 ? 1 : 2;
    CT_REQUIRE(cdiv64(lanes, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
    dim3 grid((unsigned)cdiv64(lanes, kBlock));
#define CT_FP4Q(DT, GL) hipLaunchKernelGGL((fp4_quant_pack_kernel<DT, GL>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), scale, sdt, \
                                          global_scale, reinterpret_cast<u32x4*>(packed), units, shift)
    if (xdt == CT_BF16) { if (global_scale) CT_FP4Q(CT_BF16, true); else CT_FP4Q(CT_BF16, false); }
    else { if (global_scale) CT_FP4Q(CT_F16, true); else CT_FP4Q(CT_F16, false); }
#undef CT_FP4Q
    CT_LAUNCH_CHECK("ct_fp4_quant_pack");
}

int ct_fp4_unpack_dequant(const uint8_t* packed, int64_t rows, int64_t cols, const void* scale, int scale_kind, int sdt, const float* global_scale,
                          int64_t group, void* out, int odt, ct_stream_t stream) {
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
#define CT_FP4D(DT, GL) hipLaunchKernelGGL((fp4_unpack_dequant_kernel<DT, U, GL>), grid, dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const uint32_t*>(packed), \
                                          scale, scale_kind, sdt, global_scale, out, units, shift, stride)
    if (odt == CT_BF16) { if (global_scale) CT_FP4D(CT_BF16, true); else CT_FP4D(CT_BF16, false); }
    else { if (global_scale) CT_FP4D(CT_F16, true); else CT_FP4D(CT_F16, false); }
#undef CT_FP4D
    CT_LAUNCH_CHECK("ct_fp4_unpack_dequant");
}

}  // extern "C"
