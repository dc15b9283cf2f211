// ct_quant_wb.hip — lean compress / decompress kernels for the bit widths next to 4 and 8 (round 6; VERDICT r05 missing #3).
//
// The reference's contract is 1 <= num_bits <= 8 (compressors/pack_quantized/helpers.py:39-42; tests/test_compressors/
// test_pack_quant.py:143-146 parametrises 1..8) and W3 / W2 / W6 checkpoints exist.  Rounds 1-5 served everything but 4 and 8 with the
// any-layout pack-group kernels (ct_quant_g32.inc): one lane per 32-element group on a 2-D grid, runtime dtype switches, an IEEE divide
// per element — W3 g128 at 8192^2 measured 79 us, 25 % of the HBM peak for 160 MB.  The common checkpoint shape (16-bit weights and
// scales of one dtype, int8 or no zero points, row-wise groups of a multiple of 32 columns, cols % 32 == 0) gets the treatment W4 got:
//
//   compress    a lane owns one pack group: 32 elements = four 16-byte loads in (64 B in flight, as the W4 lean kernel), BITS words out —
//               the tensor is then ONE flat stream of groups and the wave's 64 x BITS words are contiguous.  The arithmetic is the W4
//               word builder's (ct_quant_lean.h: reciprocal fast path where it is proven bit-identical, packed fp16 back end), the clamp
//               and the bias are the width's, and the 32 codes are placed at compile-time bit positions
//               (helpers.py:74-93: element i occupies bits [i b, i b + b) of the group's little-endian 32 b-bit string).
//   decompress  a lane owns one 8-element unit (one 16-byte store: a wave instruction writes 1 KiB contiguous — what made the W4
//               decompress fast; a lane with a whole group would store 16 bytes at a 64-byte stride).  A unit's 8 b bits start at bit
//               8 b j of its group (j = unit % 4) and span up to three words, overlapping its neighbours' — so the WAVE loads its
//               16 x BITS-word window once, coalesced (one or two dword loads per lane), and every lane picks its two or three words
//               out of the other lanes' registers with ds_bpermute_b32 (the LDS crossbar: no LDS memory, no barrier), funnel-shifts
//               them (v_alignbit_b32) and cuts the codes out at compile-time positions.  Vector-memory instructions per 8 elements:
//               1-2 loads + 1 store (+ the scale), as W4.
//
// Zero points of an asymmetric scheme travel as int8 here (their stored form for b != 4 is made by ct_pack_int32_dim0).
#include "ct_quant_lean.h"

namespace ct {

typedef uint32_t u32x3a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

// the biased codes (c + 2^(BITS-1), in [0, 2^BITS)) of the 8 weights of one 16-byte load
template <int DT, int BITS, bool FAST, bool ZP>
__device__ __forceinline__ void wb_codes8(const u32x4& raw, float s, float rs, float z, uint32_t (&code)[8]) {
    constexpr int OFF = 1 << (BITS - 1);
    if constexpr (DT == CT_F16 && FAST) {
        // fl16(t + MAGIC) == MAGIC + rint(t) needs an EVEN magic (the ties of t + MAGIC are then the ties of t): 1536 + OFF for BITS >= 2;
        // for BITS = 1 (OFF = 1) the magic stays 1536 and the bias is added to the byte afterwards
        constexpr int BIAS_IN_MAGIC = BITS >= 2 ? OFF : 0;
        uint32_t u[4];
        quant_pairs_f16<ZP, 1536 + BIAS_IN_MAGIC>(raw, s, rs, z, -(float)OFF, (float)(OFF - 1), u);  // low byte of each half = BIAS_IN_MAGIC + code
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            code[2 * j] = (u[j] + (uint32_t)(OFF - BIAS_IN_MAGIC)) & 0xffu;
            code[2 * j + 1] = ((u[j] >> 16) + (uint32_t)(OFF - BIAS_IN_MAGIC)) & 0xffu;
        }
    } else {
        const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
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
            c0 = c0 < -OFF ? -OFF : (c0 > OFF - 1 ? OFF - 1 : c0);  // v_med3_i32
            c1 = c1 < -OFF ? -OFF : (c1 > OFF - 1 ? OFF - 1 : c1);
            code[2 * j] = (uint32_t)(c0 + OFF);
            code[2 * j + 1] = (uint32_t)(c1 + OFF);
        }
    }
}

// the BITS words of a lane leave as the widest 4-byte-aligned stores that cover them (global stores need dword alignment only)
template <int BITS>
__device__ __forceinline__ void wb_store_words(uint32_t* __restrict__ o, const uint32_t (&w)[BITS + 1]) {
    if constexpr (BITS == 1) __builtin_nontemporal_store(w[0], o);
    else if constexpr (BITS == 2) __builtin_nontemporal_store(u32x2a4{w[0], w[1]}, reinterpret_cast<u32x2a4*>(o));
    else if constexpr (BITS == 3) __builtin_nontemporal_store(u32x3a4{w[0], w[1], w[2]}, reinterpret_cast<u32x3a4*>(o));
    else {
        __builtin_nontemporal_store(u32x4a4{w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4a4*>(o));
        if constexpr (BITS == 5) __builtin_nontemporal_store(w[4], o + 4);
        else if constexpr (BITS == 6) __builtin_nontemporal_store(u32x2a4{w[4], w[5]}, reinterpret_cast<u32x2a4*>(o + 4));
        else if constexpr (BITS == 7) __builtin_nontemporal_store(u32x3a4{w[4], w[5], w[6]}, reinterpret_cast<u32x3a4*>(o + 4));
        else if constexpr (BITS == 8) __builtin_nontemporal_store(u32x4a4{w[4], w[5], w[6], w[7]}, reinterpret_cast<u32x4a4*>(o + 4));
    }
}

template <int DT, int BITS, bool HAS_ZP>
__global__ __launch_bounds__(kBlock) void wb_quant_pack_lean_kernel(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale,
                                                                    const int8_t* __restrict__ zp, uint32_t* __restrict__ out, int64_t groups,
                                                                    int gshift /* log2(pack groups per scale group), or -1 */, int64_t gpg) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= groups) return;
    const int64_t si = gshift >= 0 ? (g >> gshift) : g / gpg;
    const uint32_t sbits = __builtin_nontemporal_load(scale + si);
    const float z = HAS_ZP ? (float)__builtin_nontemporal_load(zp + si) : 0.0f;  // int8 -> exact in bf16 / fp16
    asm volatile("" ::: "memory");  // keep the small loads ahead of the big ones (as the W4 lean kernel)
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = in[g * 4 + i];
    const float s = DT == CT_BF16 ? bf16_bits_to_f(sbits) : f16_bits_to_f(sbits);
    const bool fast = fast_scale_ok<DT>(s) && fast_data_ok<DT, 4>(r);
    const float rs = 1.0f / s;
    const bool use_zp = HAS_ZP && (__builtin_amdgcn_ballot_w64(z != 0.0f) != 0);
    uint32_t code[4][8];
    // real branches (not selects): the IEEE divide must not be issued on the fast path; the lanes of a wave almost always agree
    if (fast) {
        if (use_zp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb_codes8<DT, BITS, true, true>(r[i], s, rs, z, code[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb_codes8<DT, BITS, true, false>(r[i], s, rs, z, code[i]);
        }
    } else {
        if (use_zp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb_codes8<DT, BITS, false, true>(r[i], s, rs, z, code[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) wb_codes8<DT, BITS, false, false>(r[i], s, rs, z, code[i]);
        }
    }
    uint32_t words[BITS + 1];
#pragma unroll
    for (int j = 0; j <= BITS; ++j) words[j] = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int pos = i * BITS, w = pos >> 5, sh = pos & 31;  // compile-time after unrolling
        const uint32_t c = code[i >> 3][i & 7];
        words[w] |= c << sh;
        if (sh + BITS > 32) words[w + 1] |= c >> (32 - sh);
    }
    wb_store_words<BITS>(out + g * BITS, words);
}

constexpr int kWbScalePerLane = 0, kWbScaleScalar = 2;

// UNROLL units per lane, one block apart.  SM = 2: groups of 128 elements on a tensor whose unit count is a multiple of 64 — the wave's four
// scales (and zero points) by two scalar loads, as the W4 kernel's (ct_quant.hip: kW4ScaleScalar).
template <int DT, int BITS, int UNROLL, bool HAS_ZP, int SM>
__global__ __launch_bounds__(kBlock) void wb_unpack_dequant_kernel(const uint32_t* __restrict__ packed, const void* __restrict__ scale,
                                                                   const int8_t* __restrict__ zp, void* __restrict__ out, int64_t units, int64_t total_words,
                                                                   int upg_shift, int64_t upg) {
    constexpr int OFF = 1 << (BITS - 1);
    constexpr int NW = 16 * BITS;  // words of a wave's window: 64 units = 16 pack groups
    constexpr uint32_t kMask = (1u << BITS) - 1u;
    const unsigned lane = threadIdx.x & 63u;
    const int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x;
    uint32_t r0[UNROLL], r1[UNROLL];
    uint64_t s4[UNROLL];
    uint32_t z4[UNROLL];
    if constexpr (SM == kWbScaleScalar) {
        typedef const __attribute__((address_space(4))) uint32_t* const_u32_t;
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u0 = base + (int64_t)i * kBlock - (int64_t)lane;  // the wave's first unit, a multiple of 64
            const uint32_t si_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u0 >> 4));
            const uint32_t si_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((u0 >> 4) >> 32));
            int64_t si0 = (int64_t)(((uint64_t)si_hi << 32) | si_lo);
            const int64_t si_last = (units >> 4) - 4;
            si0 = si0 < si_last ? si0 : si_last;  // a wave beyond the tensor reads the last entries and drops them
            const_u32_t sp = (const_u32_t)(uintptr_t)(static_cast<const uint16_t*>(scale) + si0);
            s4[i] = ((uint64_t)sp[1] << 32) | sp[0];
            z4[i] = 0;
            if constexpr (HAS_ZP) z4[i] = *(const_u32_t)(uintptr_t)(zp + si0);
        }
    }
    // (BITS <= 2: a window is 16 / 32 words, half a load instruction or less.  Letting the TWO windows of a lane's two units share one load —
    // lanes [0, 32) window 0, lanes [32, 64) window 1 — measured SLOWER: W2 decompress 29.2 -> 30.9 us at 8192^2.  Round 6, dropped.)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u0 = base + (int64_t)i * kBlock - (int64_t)lane;
        const int64_t w0 = (u0 >> 6) * NW;  // the window's first word
        r0[i] = (lane < (unsigned)NW && w0 + lane < total_words) ? packed[w0 + lane] : 0u;
        r1[i] = 0u;
        if constexpr (NW > 64) r1[i] = (lane + 64u < (unsigned)NW && w0 + 64 + lane < total_words) ? packed[w0 + 64 + lane] : 0u;
    }
    // where this lane's 8 BITS bits start inside the window (the same for every i)
    const unsigned gq = lane >> 2, j = lane & 3u;
    const unsigned bit0 = j * (8u * BITS);
    const unsigned wrel = gq * BITS + (bit0 >> 5), sh = bit0 & 31u;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        // (every lane takes part in the exchange: lanes past the end only lend their registers)
        uint32_t wv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k == 2 && 8 * BITS + 31 <= 64) { wv[k] = 0; continue; }  // compile-time: a unit of <= 33 bits never reaches a third word
            const unsigned idx = wrel + k;                              // < NW + 2: an index past the window is only ever shifted out
            uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)r0[i]);
            if constexpr (NW > 64) {
                const uint32_t v1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)r1[i]);
                v = idx >= 64u ? v1 : v;
            }
            wv[k] = v;
        }
        if (u >= units) continue;
        const uint32_t x0 = __builtin_amdgcn_alignbit(wv[1], wv[0], sh);  // bits [0, 32) of the unit
        const uint32_t x1 = __builtin_amdgcn_alignbit(wv[2], wv[1], sh);  // bits [32, 64)
        float s, z;
        if constexpr (SM == kWbScaleScalar) {
            const uint32_t g16 = (threadIdx.x & 48u);
            const uint32_t sb = (uint32_t)(s4[i] >> g16);
            s = DT == CT_BF16 ? bits_f(sb << 16) : f16_bits_to_f(sb & 0xffffu);
            z = HAS_ZP ? (float)(int)__builtin_amdgcn_sbfe(z4[i], g16 >> 1, 8) : 0.0f;
        } else {
            const int64_t si = upg_shift >= 0 ? (u >> upg_shift) : u / upg;
            s = load_as_f<DT>(scale, si);
            z = HAS_ZP ? (float)zp[si] : 0.0f;
        }
        // (2^23 + code) - (2^23 + 2^(BITS-1) + z) is exact: the un-bias and the zero point cost ONE subtract
        const float off = 8388608.0f + (float)OFF + z;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            constexpr int dummy = 0;
            (void)dummy;
            const int pos = k * BITS;  // compile-time
            uint32_t c;
            if (pos + BITS <= 32) c = (x0 >> pos) & kMask;
            else if (pos >= 32) c = (x1 >> (pos - 32)) & kMask;
            else c = ((x0 >> pos) | (x1 << (32 - pos))) & kMask;
            v[k] = mul_round_to<DT>(bits_f(0x4B000000u | c) - off, s);
        }
        store8<DT>(out, u * 8, v);
    }
}

// ------------------------------------------------------------------------------------------ launchers (declared in ct_quant_core.h)
static int log2_exact64(int64_t v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while (((int64_t)1 << l) < v) ++l;
    return l;
}

#define CT_WB_BITS(bits, ...)                                \
    switch (bits) {                                          \
        case 1: { constexpr int B = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int B = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int B = 3; __VA_ARGS__; } break; \
        case 5: { constexpr int B = 5; __VA_ARGS__; } break; \
        case 6: { constexpr int B = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int B = 7; __VA_ARGS__; } break; \
        default: return -1;                                  \
    }

// the layout both directions take: 16-bit weights / scales of ONE dtype, int8 or no zero points, one scale per `cdiv` consecutive columns of
// a row with cdiv % 32 == 0 (or the whole row), cols % 32 == 0, no activation ordering, aligned buffers.  Bits 4 and 8 have kernels of their own.
bool wb_layout_ok(int dt, int sdt, int other_dt, int bits, int zdt, const void* zp, int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                  const int32_t* col_group, const void* wide, const void* words) {
    if (bits < 1 || bits > 7 || bits == 4 || col_group) return false;
    if (!(dt == CT_BF16 || dt == CT_F16) || sdt != dt || other_dt != dt) return false;
    if (zp && zdt != CT_I8) return false;
    if (rows <= 0 || cols <= 0 || cols % 32 || rdiv != 1) return false;
    const int64_t c = cdiv > cols ? cols : cdiv;
    if (c % 32 || cols % c || scale_cols != cols / c) return false;
    if (rows * (cols / 32) >= ((int64_t)1 << 38)) return false;
    return aligned16(wide) && (reinterpret_cast<uintptr_t>(words) & 3u) == 0;
}

// returns -1 when the width has no lean kernel (the caller falls through to the any-layout kernels)
int launch_wb_quant_pack(const void* x, int xdt, const void* scale, const void* zp, int64_t rows, int64_t cols, int64_t cdiv, int bits, int32_t* packed,
                         ct_stream_t stream) {
    const int64_t c = cdiv > cols ? cols : cdiv;
    const int64_t groups = rows * (cols / 32), gpg = c / 32;
    const int gshift = log2_exact64(gpg);
    const dim3 grid((unsigned)cdiv64(groups, kBlock));
#define CT_WBQ(DT, ZP) CT_WB_BITS(bits, hipLaunchKernelGGL((wb_quant_pack_lean_kernel<DT, B, ZP>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), \
                                                          static_cast<const uint16_t*>(scale), static_cast<const int8_t*>(zp), reinterpret_cast<uint32_t*>(packed), groups, gshift, gpg))
    if (xdt == CT_BF16) { if (zp) { CT_WBQ(CT_BF16, true); } else { CT_WBQ(CT_BF16, false); } }
    else { if (zp) { CT_WBQ(CT_F16, true); } else { CT_WBQ(CT_F16, false); } }
#undef CT_WBQ
    CT_LAUNCH_CHECK("ct_quant_pack[lean, any width]");
}

int launch_wb_unpack_dequant(const int32_t* packed, const void* scale, int sdt, const void* zp, int64_t rows, int64_t cols, int64_t cdiv, int bits, void* out,
                             ct_stream_t stream) {
    const int64_t c = cdiv > cols ? cols : cdiv;
    const int64_t units = rows * (cols / 8), upg = c / 8, total_words = rows * (cols / 32) * bits;
    const int upg_shift = log2_exact64(upg);
    constexpr int U = 2;
    const dim3 grid((unsigned)cdiv64(units, (int64_t)kBlock * U));
    const bool scalar = upg == 16 && units % 64 == 0 && (reinterpret_cast<uintptr_t>(scale) & 7u) == 0 && (reinterpret_cast<uintptr_t>(zp) & 3u) == 0;
#define CT_WBD(DT, ZP)                                                                                                                                                           \
    CT_WB_BITS(bits, if (scalar) hipLaunchKernelGGL((wb_unpack_dequant_kernel<DT, B, U, ZP, kWbScaleScalar>), grid, dim3(kBlock), 0, as_stream(stream),                         \
                                                    reinterpret_cast<const uint32_t*>(packed), scale, static_cast<const int8_t*>(zp), out, units, total_words, upg_shift, upg); \
                     else hipLaunchKernelGGL((wb_unpack_dequant_kernel<DT, B, U, ZP, kWbScalePerLane>), grid, dim3(kBlock), 0, as_stream(stream),                               \
                                             reinterpret_cast<const uint32_t*>(packed), scale, static_cast<const int8_t*>(zp), out, units, total_words, upg_shift, upg))
    if (sdt == CT_BF16) { if (zp) { CT_WBD(CT_BF16, true); } else { CT_WBD(CT_BF16, false); } }
    else { if (zp) { CT_WBD(CT_F16, true); } else { CT_WBD(CT_F16, false); } }
#undef CT_WBD
    CT_LAUNCH_CHECK("ct_unpack_dequant[lean, any width]");
}

}  // namespace ct
