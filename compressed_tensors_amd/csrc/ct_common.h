// ct_common.h — shared device/host helpers for libct_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/ct_hip.h"

namespace ct {

// ------------------------------------------------------------------------- error handling
void set_error(const char* fmt, ...);
int hip_check(hipError_t e, const char* what);

#define CT_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ::ct::set_error(__VA_ARGS__);     \
            return CT_ERR_INVALID_ARG;        \
        }                                     \
    } while (0)

#define CT_UNSUPPORTED(...)               \
    do {                                  \
        ::ct::set_error(__VA_ARGS__);     \
        return CT_ERR_UNSUPPORTED;        \
    } while (0)

#define CT_LAUNCH_CHECK(name) return ::ct::hip_check(hipGetLastError(), name)

inline hipStream_t as_stream(ct_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kCUs = 256;          // MI355X
constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr int kMaxGridX = 1 << 20;

inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline bool is_float_dt(int dt) { return dt == CT_F32 || dt == CT_F16 || dt == CT_BF16; }
inline int dt_size(int dt) {
    switch (dt) {
        case CT_F32: case CT_I32: return 4;
        case CT_F16: case CT_BF16: case CT_I16: return 2;
        case CT_I8: case CT_U8: case CT_F8E4M3: return 1;
        case CT_I64: return 8;
    }
    return 0;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// log2 of a power of two, or -1
inline int log2_exact(int64_t v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((int64_t(1) << l) < v) ++l;
    return l;
}

// ------------------------------------------------------------------------- device: formats
typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// streaming stores: every large output of this library is written once and not re-read by the
// writing kernel.  A non-temporal 16-byte store does not linger dirty in the XCD L2, which removes
// the end-of-kernel write-back bubble: measured on MI355X at 8192^2 (tools/kbench/kbench_flavors),
// the W4 decompress drops from 35.9 us to 24.3 us and a 84 MB copy from 32.9 us to 29.9 us.
// (Non-temporal LOADS measured slower, 29.3 -> 42 us on the compress side: loads stay plain.)
__device__ __forceinline__ void stream_store16(void* p, u32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}
__device__ __forceinline__ void stream_store8(void* p, u32x2 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
}

__device__ __forceinline__ float bits_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ float bf16_bits_to_f(uint32_t h16) { return bits_f(h16 << 16); }
__device__ __forceinline__ float f16_bits_to_f(uint32_t h16) {
    return (float)__builtin_bit_cast(f16_t, (uint16_t)h16);
}
// RNE conversions (v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950)
__device__ __forceinline__ uint32_t f_to_bf16_bits(float v) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (bf16_t)v);
}
__device__ __forceinline__ uint32_t f_to_f16_bits(float v) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (f16_t)v);
}

// float8_e4m3fn <-> float: gfx950's conversions are the OCP ones (RNE, subnormals; 0x7f = NaN).  Callers clamp to
// +-448 first (torch.clamp precedes the cast upstream), so the overflow behaviour of the instruction is never reached.
__device__ __forceinline__ float fp8_to_f(uint32_t byte) { return __builtin_amdgcn_cvt_f32_fp8((int)byte, 0); }
__device__ __forceinline__ uint32_t f2_to_fp8x2(float a, float b) {  // two bytes in the low half
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}
__device__ __forceinline__ float fp8_round(float t) {  // value of float8(t) for |t| <= 448; NaN stays NaN
    const float r = fp8_to_f(f2_to_fp8x2(t, 0.0f) & 0xffu);
    return (t != t) ? t : r;
}

// cast_to_fp4 (quantization/utils/fp4_utils.py:77-98) of a clamped value: the E2M1 grid with RNE ties (upstream's
// thresholds), a negative that rounds to zero gives -0.0, -0.0 itself gives +0.0 (sign(-0.0) == 0), NaN stays NaN
__device__ __forceinline__ float fp4_round(float t) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const uint32_t b = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, t + 0.0f, 0.0f, 1.0f, 0);
    const f2 p = __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(b, 1.0f, 0);
    return (t != t) ? t : p.x;
}

// round a float to dtype DT and back: "every torch op rounds to the tensor dtype"
template <int DT>
__device__ __forceinline__ float round_to(float v) {
    if constexpr (DT == CT_BF16) return bf16_bits_to_f(f_to_bf16_bits(v));
    else if constexpr (DT == CT_F16) return (float)(f16_t)v;
    else return v;
}

// rnd_DT(a * b).  For fp16 the product is pinned in a register first: clang otherwise selects
// v_fma_mixlo_f16 a, b, +0 for "convert the product to half", and (-0.0 * s) + (+0.0) is +0.0 — the sign of a
// zero product (a -0.0 float8 / FP4 code, or a zero code under a negative scale) would be lost.  The empty asm
// emits nothing.
template <int DT>
__device__ __forceinline__ float mul_round_to(float a, float b) {
    float p = a * b;
    if constexpr (DT == CT_F16) asm("" : "+v"(p));
    return round_to<DT>(p);
}

__device__ __forceinline__ float round_to_rt(int dt, float v) {
    switch (dt) {
        case CT_BF16: return round_to<CT_BF16>(v);
        case CT_F16: return round_to<CT_F16>(v);
        default: return v;
    }
}

__device__ __forceinline__ float mul_round_to_rt(int dt, float a, float b) {
    switch (dt) {
        case CT_BF16: return mul_round_to<CT_BF16>(a, b);
        case CT_F16: return mul_round_to<CT_F16>(a, b);
        default: return a * b;
    }
}

// one element of any supported dtype as float (runtime dtype; used for scale / zero point)
__device__ __forceinline__ float load_rt(const void* p, int dt, int64_t i) {
    switch (dt) {
        case CT_F32: return static_cast<const float*>(p)[i];
        case CT_F16: return f16_bits_to_f(static_cast<const uint16_t*>(p)[i]);
        case CT_BF16: return bf16_bits_to_f(static_cast<const uint16_t*>(p)[i]);
        case CT_I8: return (float)static_cast<const int8_t*>(p)[i];
        case CT_I32: return (float)static_cast<const int32_t*>(p)[i];
        case CT_U8: return (float)static_cast<const uint8_t*>(p)[i];
        case CT_I16: return (float)static_cast<const int16_t*>(p)[i];
        case CT_I64: return (float)static_cast<const int64_t*>(p)[i];
        case CT_F8E4M3: return fp8_to_f(static_cast<const uint8_t*>(p)[i]);
    }
    return 0.0f;
}

template <int DT>
__device__ __forceinline__ float load_as_f(const void* p, int64_t i) {
    if constexpr (DT == CT_F32) return static_cast<const float*>(p)[i];
    else if constexpr (DT == CT_F16) return f16_bits_to_f(static_cast<const uint16_t*>(p)[i]);
    else return bf16_bits_to_f(static_cast<const uint16_t*>(p)[i]);
}

// 8 consecutive elements -> 8 floats; `vec` selects 16-byte loads (requires alignment)
template <int DT>
__device__ __forceinline__ void load8(const void* base, int64_t i0, float (&v)[8]) {
    if constexpr (DT == CT_F32) {
        const f32x4* p = reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + i0);
        f32x4 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        u32x4 w = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(base) + i0);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (DT == CT_BF16) {
                v[2 * j] = bits_f(ws[j] << 16);
                v[2 * j + 1] = bits_f(ws[j] & 0xffff0000u);
            } else {
                v[2 * j] = f16_bits_to_f(ws[j] & 0xffffu);
                v[2 * j + 1] = f16_bits_to_f(ws[j] >> 16);
            }
        }
    }
}

// 8 floats -> 8 consecutive elements of dtype DT (values are converted with RNE)
template <int DT>
__device__ __forceinline__ void store8(void* base, int64_t i0, const float (&v)[8]) {
    if constexpr (DT == CT_F32) {
        f32x4* p = reinterpret_cast<f32x4*>(static_cast<float*>(base) + i0);
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, p);
        __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, p + 1);
    } else {
        uint32_t ws[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (DT == CT_BF16) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef bf16_t b2 __attribute__((ext_vector_type(2)));
                b2 r = __builtin_convertvector(f2{v[2 * j], v[2 * j + 1]}, b2);
                ws[j] = __builtin_bit_cast(uint32_t, r);
            } else {
                ws[j] = f_to_f16_bits(v[2 * j]) | (f_to_f16_bits(v[2 * j + 1]) << 16);
            }
        }
        stream_store16(static_cast<uint16_t*>(base) + i0, u32x4{ws[0], ws[1], ws[2], ws[3]});
    }
}

template <int DT>
__device__ __forceinline__ void store1(void* base, int64_t i, float v) {
    if constexpr (DT == CT_F32) static_cast<float*>(base)[i] = v;
    else if constexpr (DT == CT_F16) static_cast<uint16_t*>(base)[i] = (uint16_t)f_to_f16_bits(v);
    else static_cast<uint16_t*>(base)[i] = (uint16_t)f_to_bf16_bits(v);
}

__device__ __forceinline__ void store_rt(void* base, int dt, int64_t i, float v) {
    switch (dt) {
        case CT_F32: store1<CT_F32>(base, i, v); break;
        case CT_F16: store1<CT_F16>(base, i, v); break;
        case CT_BF16: store1<CT_BF16>(base, i, v); break;
        // float -> int of an integral, in-range value; NaN converts to 0 (v_cvt_i32_f32),
        // which is also what the reference's CPU cast produces
        case CT_I8: static_cast<int8_t*>(base)[i] = (int8_t)(int)v; break;
        case CT_I32: static_cast<int32_t*>(base)[i] = (int)v; break;
        case CT_F8E4M3: static_cast<uint8_t*>(base)[i] = (uint8_t)(f2_to_fp8x2(v, 0.0f) & 0xffu); break;
    }
}

// BITS consecutive int32 words of a pack group: 16-byte / 8-byte vectors when the width and the
// address allow (a lane owns BITS words; scalar stores at a BITS-word stride waste the store path:
// W8 generic compress 94 us -> see DESIGN.md)
template <int BITS>
__device__ __forceinline__ void store_words(int32_t* o, const uint32_t* words, int64_t limit /*words still inside the row*/) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(o);
    if (limit >= BITS) {
        if constexpr (BITS % 4 == 0) {
            if ((a & 15u) == 0) {
#pragma unroll
                for (int j = 0; j < BITS; j += 4) stream_store16(o + j, u32x4{words[j], words[j + 1], words[j + 2], words[j + 3]});
                return;
            }
        }
        if constexpr (BITS % 2 == 0) {
            if ((a & 7u) == 0) {
#pragma unroll
                for (int j = 0; j < BITS; j += 2) stream_store8(o + j, u32x2{words[j], words[j + 1]});
                return;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < BITS; ++j)
        if (j < limit) o[j] = (int32_t)words[j];
}

template <int BITS>
__device__ __forceinline__ void load_words(const int32_t* in, uint32_t* words, int64_t limit) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(in);
    if (limit >= BITS) {
        if constexpr (BITS % 4 == 0) {
            if ((a & 15u) == 0) {
#pragma unroll
                for (int j = 0; j < BITS; j += 4) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(in + j);
                    words[j] = v.x; words[j + 1] = v.y; words[j + 2] = v.z; words[j + 3] = v.w;
                }
                return;
            }
        }
        if constexpr (BITS % 2 == 0) {
            if ((a & 7u) == 0) {
#pragma unroll
                for (int j = 0; j < BITS; j += 2) {
                    const u32x2 v = *reinterpret_cast<const u32x2*>(in + j);
                    words[j] = v.x; words[j + 1] = v.y;
                }
                return;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < BITS; ++j) words[j] = j < limit ? (uint32_t)in[j] : 0u;
}

// ------------------------------------------------------------------------- quantization layout
struct QLayout {
    int64_t rows, cols;
    int64_t rdiv, cdiv, scale_cols;
    const int32_t* col_group;
    int cdiv_shift;  // log2(cdiv) or -1
};

__device__ __forceinline__ int64_t col_group_of(const QLayout& L, int64_t c) {
    if (L.col_group) return (int64_t)L.col_group[c];
    return L.cdiv_shift >= 0 ? (c >> L.cdiv_shift) : (c / L.cdiv);
}

// clamp that propagates NaN like torch.clamp, for finite bounds
__device__ __forceinline__ float clamp_nan(float t, float lo, float hi) {
    float c = __builtin_fminf(__builtin_fmaxf(t, lo), hi);
    return (t != t) ? t : c;
}

}  // namespace ct
