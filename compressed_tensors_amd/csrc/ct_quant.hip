// ct_quant.hip — quantize / dequantize / fake_quantize and the fused compress (quantize+pack)
// and decompress (unpack+dequantize) kernels for gfx950.
//
// Arithmetic model (bit-exact with the reference's eager CPU op sequence; SURVEY.md §8a R5/R6,
// reference quantization/lifecycle/forward_helpers.py:523-572, quant_args.py:460-496):
//   quantize:   t = rnd_T(x / s); [t = rnd_T(t + rnd_X(zp))]; t = clamp(t, qmin, qmax);
//               t = rint(t) (half-even); cast
//   dequantize: d = S(q); [d = rnd_S(d - rnd_S(zp))]; y = rnd_S(d * s); cast
// T = torch result dtype of x / scale, S = scale dtype, X = x dtype.  All of this is
// bandwidth-bound streaming work: 16-byte coalesced global accesses, one 8-element unit per
// lane, no LDS needed for the data itself (the 8-code nibble group of a lane is exactly one
// 32-bit word); scales are broadcast loads served by L1/L2.
#include "ct_quant_core.h"
#include "ct_quant_lean.h"
#include "ct_minmax.h"

#include <cstdlib>

namespace ct {

// ------------------------------------------------------------------------------------------
// quantize / fake_quantize: one 8-element unit per lane per iteration
// ------------------------------------------------------------------------------------------
enum { MODE_Q = 0, MODE_FQ = 1 };

template <int XDT, int TDT, int MODE>
__global__ __launch_bounds__(kBlock) void quant_units_kernel(QParams p) {
    const int64_t cols = p.L.cols;
    const int64_t upr = (cols + 7) >> 3;
    const bool has_zp = p.zp != nullptr;
    for (int64_t row = blockIdx.y; row < p.L.rows; row += gridDim.y) {
        const int64_t srow = (row / p.L.rdiv) * p.L.scale_cols;
        for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < upr;
             u += (int64_t)gridDim.x * kBlock) {
            const int64_t c0 = u << 3;
            const int n = (int)((cols - c0) < 8 ? (cols - c0) : 8);
            const int64_t i0 = row * cols + c0;
            float v[8];
            if (p.vec && n == 8) {
                load8<XDT>(p.x, i0, v);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = k < n ? load_as_f<XDT>(p.x, i0 + k) : 0.0f;
            }
            const bool uni = unit_uniform(p.L, c0, n);
            SZ sz = load_sz_q<XDT>(p, srow, c0);
            // one reciprocal per unit instead of eight divides when x, T and the scale are all bf16, or all fp16 (+ a Newton step and
            // the exact fix-up of the sub-2^-13 quotients: fast_quotient)
            const bool can_rcp = !p.gscale && ((XDT == CT_BF16 && TDT == CT_BF16 && p.sdt == CT_BF16) || (XDT == CT_F16 && TDT == CT_F16 && p.sdt == CT_F16) ||
                                               TDT == CT_F32);
            float rs = (can_rcp && uni) ? (TDT == CT_BF16 ? bf16_fast_rcp(sz.s) : (TDT == CT_F16 ? f16_newton_rcp(sz.s) : f32_fast_rcp(sz.s))) : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < n) {
                    if (!uni && k > 0) sz = load_sz_q<XDT>(p, srow, c0 + k);
                    float t = quant_core<TDT>(v[k], sz.s, has_zp, sz.z, p.qmin, p.qmax, rs, p.fkind);
                    if constexpr (MODE == MODE_FQ) {
                        // dequantize in S = scale dtype (forward_helpers.py:207-215)
                        float zs = has_zp ? round_to_rt(p.sdt_arith, load_rt(p.zp, p.zdt, srow + col_group_of(p.L, c0 + k))) : 0.0f;
                        float d = round_to_rt(p.sdt_arith, t);
                        if (has_zp) d = round_to_rt(p.sdt_arith, d - zs);
                        t = mul_round_to_rt(p.sdt_arith, d, sz.s);
                    }
                    v[k] = t;
                }
            }
            store_unit(p.out, p.odt, i0, v, n, p.vec);
        }
    }
}

// ------------------------------------------------------------------------------------------
// dequantize: int8 (or any supported dtype) codes -> float
// ------------------------------------------------------------------------------------------
template <int SDT>
__global__ __launch_bounds__(kBlock) void dequant_units_kernel(QParams p) {
    const int64_t cols = p.L.cols;
    const int64_t upr = (cols + 7) >> 3;
    const bool has_zp = p.zp != nullptr;
    for (int64_t row = blockIdx.y; row < p.L.rows; row += gridDim.y) {
        const int64_t srow = (row / p.L.rdiv) * p.L.scale_cols;
        for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < upr;
             u += (int64_t)gridDim.x * kBlock) {
            const int64_t c0 = u << 3;
            const int n = (int)((cols - c0) < 8 ? (cols - c0) : 8);
            const int64_t i0 = row * cols + c0;
            float v[8];
            if (p.vec && n == 8 && p.xdt == CT_I8) {
                u32x2 w = *reinterpret_cast<const u32x2*>(static_cast<const int8_t*>(p.x) + i0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = (float)(int8_t)(w.x >> (8 * k));
                    v[4 + k] = (float)(int8_t)(w.y >> (8 * k));
                }
            } else if (p.vec && n == 8 && p.xdt == CT_F8E4M3) {
                u32x2 w = *reinterpret_cast<const u32x2*>(static_cast<const uint8_t*>(p.x) + i0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {  // float8 -> S is exact for every S
                    v[k] = fp8_to_f((w.x >> (8 * k)) & 0xffu);
                    v[4 + k] = fp8_to_f((w.y >> (8 * k)) & 0xffu);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    v[k] = k < n ? round_to<SDT>(load_rt(p.x, p.xdt, i0 + k)) : 0.0f;  // x_q.to(S)
            }
            const bool uni = unit_uniform(p.L, c0, n);
            SZ sz = load_sz_dq<SDT>(p, srow, c0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < n) {
                    if (!uni && k > 0) sz = load_sz_dq<SDT>(p, srow, c0 + k);
                    v[k] = dequant_core<SDT>(v[k], has_zp, sz.z, sz.s);
                }
            }
            store_unit(p.out, p.odt, i0, v, n, p.vec);
        }
    }
}

// ------------------------------------------------------------------------------------------
// HOT PATH — W4A16: 4-bit, 16-bit weights and scales of the same dtype.
//
// The tensor is one flat stream of 8-element units: 16 B of bf16 <-> one packed int32 word (a
// lane's 8 nibbles ARE word u of the row-major packed tensor, so no cross-lane exchange is
// needed).  Access shapes were chosen by measurement on MI355X (tools/kbench, profiles/):
//   compress:   a lane owns 4 CONSECUTIVE units = 64 contiguous bytes in (4 x global_load_dwordx4)
//               and ONE 16-byte store out.  Strided 16-byte loads cost nothing, 4-byte stores do:
//               29.8-30.9 us vs 43 us for the unit-per-lane layout at 8192^2 (the 16B->4B copy
//               ceiling on the same box is 29.6 us).
//   decompress: a lane owns UNROLL units spaced one block apart: 4-byte loads (256 B contiguous
//               per wave instruction) and 16-byte stores (1 KiB contiguous per wave instruction);
//               strided stores are what hurts in this direction (49 us), UNROLL=2 is the optimum
//               (30.1 us; 4B->16B copy ceiling 31.0 us).
// VALU budget (compress is co-bound): 11 VALU ops per element with a zero point, 9 without —
// packed multiplies, v_cvt_pk_bf16_f32 for the torch-style rounding to bf16, and the hardware
// float->int conversion (saturating, NaN -> 0, exactly the reference's int8 cast) + v_med3_i32.
// ------------------------------------------------------------------------------------------
struct W4Params {
    const void* x;         // weight (compress) / packed words (decompress)
    const void* scale;
    const void* zp;        // nullable
    void* out;
    int zdt;
    int64_t units;         // rows * cols / 8
    int64_t upr;           // units per row = cols / 8
    int64_t rdiv;          // rows per scale row
    int64_t scale_cols;
    int upg_shift;         // log2(units per scale group) = log2(cdiv / 8), or -1
    int64_t upg;           // units per group (cdiv / 8)
    int flat_scale;        // 1: scale index == unit / upg (rdiv == 1 and cdiv | cols)
    // round 6, decompress with the zero points in their STORED form (ZPACKED instantiations only): int32 (ceil(rows / 8), scale_cols) words,
    // nibble r % 8 of word (r / 8, g) = zp[r][g] + 8; n / scale_cols == (n * g_magic) >> g_shift for n < 2^31
    const uint32_t* zpk = nullptr;
    uint32_t g_magic = 0;
    int g_shift = 0;
    // block strategy (scale rows of `rdiv` weight rows: FP8 / int8 block 128 x 128): the row, the scale row and the column group of a unit as
    // 32-bit multiply-highs, n / d == (n * magic) >> shift for n, d < 2^31 (make_w4) — the 64-bit divisions below cost the quantize kernels
    // 3-4 us of 39 at 8192^2 (two per scale index; profiles/r06_shape_sweep_floats.txt)
    int nf_fast = 0;
    uint32_t upr_magic = 0, rdiv_magic = 0, upg_magic = 0;
    int upr_shift = 0, rdiv_shift = 0, upg_mshift = 0;
};

__device__ __forceinline__ int64_t w4_scale_index(const W4Params& p, int64_t u) {
    if (p.flat_scale) return p.upg_shift >= 0 ? (u >> p.upg_shift) : (u / p.upg);
    if (p.nf_fast) {
        const uint32_t n = (uint32_t)u;
        const uint32_t row = (uint32_t)(((uint64_t)n * p.upr_magic) >> p.upr_shift);
        const uint32_t cu = n - row * (uint32_t)p.upr;
        const uint32_t rb = (uint32_t)(((uint64_t)row * p.rdiv_magic) >> p.rdiv_shift);
        const uint32_t cg = p.upg_shift >= 0 ? (cu >> p.upg_shift) : (uint32_t)(((uint64_t)cu * p.upg_magic) >> p.upg_mshift);
        return (int64_t)rb * p.scale_cols + (int64_t)cg;
    }
    const int64_t row = u / p.upr, cu = u - row * p.upr;
    return (row / p.rdiv) * p.scale_cols + (p.upg_shift >= 0 ? (cu >> p.upg_shift) : (cu / p.upg));
}

// Q = 4 consecutive units per lane; SHARED: the 4 units share one scale (cdiv % 32 == 0)
template <int DT, bool HAS_ZP, bool SHARED>
__device__ __forceinline__ void w4_quant_pack_group(const W4Params& p, int64_t g) {
    constexpr int Q = 4;
    const u32x4* in = static_cast<const u32x4*>(p.x);
    u32x4* out = static_cast<u32x4*>(p.out);
    u32x4 r[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) r[i] = in[g * Q + i];
    uint32_t w[Q];
    float s = 0.0f, z = 0.0f, rs = 0.0f;
    bool fast = false, use_zp = false;
    const bool data_ok = fast_data_ok<DT, Q>(r);
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        if (i == 0 || !SHARED) {
            const int64_t si = w4_scale_index(p, g * Q + i);
            s = load_as_f<DT>(p.scale, si);
            z = HAS_ZP ? round_to<DT>(load_rt(p.zp, p.zdt, si)) : 0.0f;  // zp.to(x.dtype)
            fast = fast_scale_ok<DT>(s) && data_ok;
            rs = 1.0f / s;
            // adding an all-zero zero point is the identity (t is already rounded): skip the
            // add + second rounding when every lane of the wave has z == 0 (symmetric schemes)
            use_zp = HAS_ZP && (__builtin_amdgcn_ballot_w64(z != 0.0f) != 0);
        }
        // real branches (not selects): the IEEE divide is 11 VALU ops per element and must not
        // be issued on the fast path; lanes of a wave almost always agree
        if (fast) {
            w[i] = use_zp ? w4_quant_word<DT, true, true>(r[i], s, rs, z) : w4_quant_word<DT, true, false>(r[i], s, rs, z);
        } else {
            w[i] = use_zp ? w4_quant_word<DT, false, true>(r[i], s, rs, z) : w4_quant_word<DT, false, false>(r[i], s, rs, z);
        }
    }
    stream_store16(out + g, u32x4{w[0], w[1], w[2], w[3]});
}

template <int DT, bool HAS_ZP, bool SHARED>
__global__ __launch_bounds__(kBlock) void w4_quant_pack_kernel(W4Params p) {
    const int64_t groups = p.units / 4;  // host guarantees units % 4 == 0
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock)
        w4_quant_pack_group<DT, HAS_ZP, SHARED>(p, g);
}


constexpr int kGidxMaxGroups = 1024;
// ---- activation ordering, round 3: R rows per workgroup.  What round 2's kernel above actually paid for (bench leg `bf16_8192_actorder`,
// HBM-cold: 43 / 45 us for a 29 us job) was not the gathers: (1) every workgroup first fetched its row's scales, waited, filled the LDS
// table, hit a barrier and only THEN issued its data loads — two dependent HBM latencies per 512 units of work; (2) the group table is
// int32: 32 bytes of table per 4-byte packed word / 16-byte weight unit, re-fetched from the L2 for every row (268 MB of L2 reads for
// a 168 MB job); (3) the compress side stored 4 bytes per lane.  Here a lane keeps the group numbers of ITS columns in registers (as
// 16-bit LDS byte offsets, two per register) and reuses them for R rows; all R rows' data loads are issued before anything is
// waited for; the R rows' scale entries go to LDS in one pass (one barrier per workgroup); (3) turned out NOT to be the lever: a lane
// with 4 units / one 16-byte store needs 140 VGPRs and measured 41.5 us, one unit per lane 30.6 (see the constants below).  LDS entry: the reciprocal (compress) or the scale (decompress), with the zero point
// beside it in one 8-byte entry when the scheme has one — ONE ds_read per element.
// Round 4, measured and dropped: requesting this thread's scale entry FIRST (inline asm, converted behind vmcnt(<loads issued after it>))
// and making the group-number / data loads unconditional so that the rows are consumed behind vmcnt(3 .. 0) — what bought marlin-24
// 10 % — measured 32.8 / 32.5 us here against 31.2 / 31.8: at 8 waves per SIMD the exposed latency was already
// covered by other waves, and the clamped indices plus the early conversions only added instructions.
constexpr int kGidxRows = 4;
// rows per workgroup R and units per lane UL, measured at 8192^2 bf16 on the product's own template (tools/kbench/kbench_prod.hip `gidx`,
// profiles/r03_kbench_prod.txt, r03_kbench_gidx_small_tables.txt).  With the tables sized for 1024 groups per row LDS capped the
// kernel at 3-5 workgroups per CU: compress (R, UL) = (4, 4) 41.5 us — 140 VGPRs —, (4, 2) 32.9, (4, 1) 34.4, (2, 2) 35.2, (8, 2) 48.4.
// With tables for <= 128 groups (6-8 waves per SIMD): compress (4, 1) 30.6, (8, 1) 33.8, (4, 2) 34.5, (2, 2) 35.2, (4, 4) 41.3;
// decompress (8, 1) 30.3, (4, 1) 30.5, (8, 2) 31.9, (4, 2) 32.2.  One unit per lane on both sides: occupancy beats store width here.
constexpr int kGidxCompressUnits = 1;    // units per lane on the compress side (one 4-byte word per lane and row)
constexpr int kGidxDecompressUnits = 1;  // units per lane on the decompress side (one 16-byte store per lane and row)
constexpr int kGidxSmallGroups = 128;  // tables sized for rows of up to 128 groups (8192 columns of group 64) keep LDS out of the occupancy limit
template <int DT, bool HAS_ZP, bool COMPRESS, int R = kGidxRows, int UL = (COMPRESS ? kGidxCompressUnits : kGidxDecompressUnits), int MAXG = kGidxMaxGroups>
__global__ __launch_bounds__(kBlock) void w4_gidx_rows_kernel(W4Params p, const int32_t* __restrict__ col_group, int chunks_per_row, int64_t rows) {
    static_assert(!COMPRESS || UL == 1 || UL == 2 || UL == 4, "the compress side stores UL consecutive words per lane");
    constexpr int ESZ = HAS_ZP ? 8 : 4;                   // bytes per LDS entry
    constexpr int kRowBytes = MAXG * ESZ;                 // compile-time row stride: the row index is an instruction offset
    __shared__ __attribute__((aligned(16))) unsigned char s_tab[R * kRowBytes];
    __shared__ float s_slow[COMPRESS ? R * MAXG : 1];     // the scales themselves, read only by lanes that must divide
    const int64_t rb = blockIdx.x / (unsigned)chunks_per_row;
    const int chunk = (int)(blockIdx.x - (unsigned)rb * (unsigned)chunks_per_row);
    const int64_t row0 = rb * R;
    // this lane's units: compress UL consecutive ones, decompress UL one block apart (16-byte stores, 1 KiB per wave instruction)
    int64_t cu[UL];
    bool live[UL];
#pragma unroll
    for (int i = 0; i < UL; ++i) {
        cu[i] = COMPRESS ? ((int64_t)chunk * kBlock + threadIdx.x) * UL + i : (int64_t)chunk * (UL * kBlock) + (int64_t)i * kBlock + threadIdx.x;
        live[i] = cu[i] < p.upr;
    }
    // 1. group numbers of this lane's columns (the same for every row)
    u32x4 gv[UL][2];
#pragma unroll
    for (int i = 0; i < UL; ++i) {
        if (live[i]) {
            gv[i][0] = reinterpret_cast<const u32x4*>(col_group)[2 * cu[i]];
            gv[i][1] = reinterpret_cast<const u32x4*>(col_group)[2 * cu[i] + 1];
        } else {
            gv[i][0] = gv[i][1] = u32x4{0, 0, 0, 0};
        }
    }
    // 2. every row's data loads, before anything is waited for
    u32x4 wv[COMPRESS ? R : 1][COMPRESS ? UL : 1];
    uint32_t pw[COMPRESS ? 1 : R][COMPRESS ? 1 : UL];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r;
#pragma unroll
        for (int i = 0; i < UL; ++i) {
            if (row < rows && live[i]) {
                if constexpr (COMPRESS) wv[r][i] = static_cast<const u32x4*>(p.x)[row * p.upr + cu[i]];
                else pw[r][i] = static_cast<const uint32_t*>(p.x)[row * p.upr + cu[i]];
            }
        }
    }
    // 3. the R rows' entries -> LDS.  Round 6: every load of the table is issued before the first conversion, with compile-time trip counts (row by
    // row, MAXG / 256 entries per lane and row) — the earlier `for (e = tid; e < R * scale_cols; e += 256)` loop ran its iterations one memory
    // latency after the other and divided by scale_cols in each: 4 dependent round trips at 224 groups per row (28672 columns of group 128)
    constexpr int KG = MAXG > kBlock ? MAXG / kBlock : 1;
    bool nz = false;  // some zero point of these rows is not zero
    // (the zero point's dtype is decided ONCE, outside the loads: a dtype switch around every load ends in a merge point where the wait-count pass
    // drains all loads — 4 x R dependent round trips instead of one; measured 32.9 -> 45.1 us at 8192^2 with the switch inside)
    auto stage_table = [&](auto load_zp) {
        float sv[R][KG], zv[HAS_ZP ? R : 1][HAS_ZP ? KG : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const int g = (int)threadIdx.x + k * kBlock;
                const bool ok = g < (int)p.scale_cols && row0 + r < rows;
                const int64_t si = ok ? (row0 + r) * p.scale_cols + g : 0;  // (clamped: the loads are unconditional)
                sv[r][k] = load_as_f<DT>(p.scale, si);
                if constexpr (HAS_ZP) zv[r][k] = load_zp(si);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int k = 0; k < KG; ++k) {
                const int g = (int)threadIdx.x + k * kBlock;
                if (g < (int)p.scale_cols && row0 + r < rows) {
                    const float sc = sv[r][k];
                    const float v = COMPRESS ? (DT == CT_BF16 ? bf16_fast_rcp(sc) : f16_newton_rcp(sc)) : sc;
                    if constexpr (HAS_ZP) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const float z = round_to<DT>(zv[r][k]);
                        nz |= z != 0.0f;
                        *reinterpret_cast<f2*>(s_tab + r * kRowBytes + g * 8) = f2{v, z};
                    } else {
                        *reinterpret_cast<float*>(s_tab + r * kRowBytes + g * 4) = v;
                    }
                    if constexpr (COMPRESS) s_slow[r * MAXG + g] = sc;
                }
            }
        }
    };
    if (HAS_ZP && p.zdt == CT_I8) stage_table([&](int64_t si) { return (float)static_cast<const int8_t*>(p.zp)[si]; });
    else stage_table([&](int64_t si) { return HAS_ZP ? load_rt(p.zp, p.zdt, si) : 0.0f; });
    // LDS byte offsets of the groups, two per register
    uint32_t go[UL][4];
#pragma unroll
    for (int i = 0; i < UL; ++i) {
        const uint32_t gs[8] = {gv[i][0].x, gv[i][0].y, gv[i][0].z, gv[i][0].w, gv[i][1].x, gv[i][1].y, gv[i][1].z, gv[i][1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) go[i][j] = (gs[2 * j] * ESZ) | ((gs[2 * j + 1] * ESZ) << 16);
    }
    // symmetric schemes hand over an all-zero zero point: adding it is the identity (t is already rounded), so the add and the second
    // rounding are skipped when no zero point of the workgroup's rows is set — the wave-uniform skip of the flat kernels, per workgroup
    const bool use_zp = HAS_ZP && (COMPRESS ? __syncthreads_or(nz) != 0 : true);
    if (!(HAS_ZP && COMPRESS)) __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r;
        if (row >= rows) break;
        const unsigned char* tab = s_tab + r * kRowBytes;
        if constexpr (COMPRESS) {
            uint32_t words[UL];
#pragma unroll
            for (int i = 0; i < UL; ++i) {
                const uint32_t ws[4] = {wv[r][i].x, wv[r][i].y, wv[r][i].z, wv[r][i].w};
                uint32_t word = 0x88888888u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x0, x1;
                    unpack2<DT>(ws[j], x0, x1);
                    const uint32_t oa = go[i][j] & 0xffffu, ob = go[i][j] >> 16;
                    float ra, rb2, za = 0.0f, zb = 0.0f;
                    if constexpr (HAS_ZP) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const f2 ea = *reinterpret_cast<const f2*>(tab + oa), eb = *reinterpret_cast<const f2*>(tab + ob);
                        ra = ea.x; za = ea.y; rb2 = eb.x; zb = eb.y;
                    } else {
                        ra = *reinterpret_cast<const float*>(tab + oa); rb2 = *reinterpret_cast<const float*>(tab + ob);
                    }
                    int c0, c1;
                    if (DT == CT_BF16 && ra != 0.0f && rb2 != 0.0f) {
                        // both scales inside the proven range: w4_quant_word's arithmetic with one reciprocal per element
                        float t0 = x0 * ra, t1 = x1 * rb2;
                        round2<DT>(t0, t1);
                        if (use_zp) {
                            t0 += za; t1 += zb;
                            round2<DT>(t0, t1);
                        }
                        c0 = cvt_i32_hw(__builtin_rintf(t0)); c1 = cvt_i32_hw(__builtin_rintf(t1));
                        c0 = c0 < -8 ? -8 : (c0 > 7 ? 7 : c0);
                        c1 = c1 < -8 ? -8 : (c1 > 7 ? 7 : c1);
                    } else {
                        const float sa = s_slow[r * MAXG + oa / ESZ], sb = s_slow[r * MAXG + ob / ESZ];
                        c0 = cvt_i32_hw(quant_core<DT>(x0, sa, HAS_ZP, za, -8.0f, 7.0f, ra));  // NaN -> code 0
                        c1 = cvt_i32_hw(quant_core<DT>(x1, sb, HAS_ZP, zb, -8.0f, 7.0f, rb2));
                    }
                    word += (uint32_t)c0 << (8 * j);       // codes in [-8, 7] on top of the 0x88888888 bias: no carries
                    word += (uint32_t)c1 << (8 * j + 4);
                }
                words[i] = word;
            }
            if (live[UL - 1]) {  // every unit of the lane is inside the row: one wide store
                uint32_t* dst = static_cast<uint32_t*>(p.out) + row * p.upr + cu[0];
                if constexpr (UL == 4) stream_store16(dst, u32x4{words[0], words[1], words[2], words[3]});
                else if constexpr (UL == 2) stream_store8(dst, u32x2{words[0], words[1]});
                else __builtin_nontemporal_store(words[0], dst);
            } else if constexpr (UL > 1) {  // the row ends inside this lane's units (upr % UL != 0): word by word, live units only
#pragma unroll
                for (int i = 0; i < UL; ++i)
                    if (live[i]) __builtin_nontemporal_store(words[i], static_cast<uint32_t*>(p.out) + row * p.upr + cu[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < UL; ++i) {
                if (!live[i]) continue;
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t o = (k & 1) ? (go[i][k >> 1] >> 16) : (go[i][k >> 1] & 0xffffu);
                    const float q = (float)((int)((pw[r][i] >> (4 * k)) & 15u) - 8);
                    if constexpr (HAS_ZP) {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const f2 e = *reinterpret_cast<const f2*>(tab + o);
                        v[k] = dequant_core<DT>(q, true, e.y, e.x);
                    } else {
                        v[k] = dequant_core<DT>(q, false, 0.0f, *reinterpret_cast<const float*>(tab + o));
                    }
                }
                store8<DT>(p.out, (row * p.upr + cu[i]) * 8, v);
            }
        }
    }
}

// ---- fp32 weights (the reference's own unit tests feed them; fp32 checkpoints exist): a lane takes one unit = 8 floats = two 16-byte
// loads and produces one packed word; two units per lane, a block apart, all four loads issued first.  The any-width kernels
// (ct_quant_g32.inc) give a lane 32 elements = 128 bytes at a 128-byte lane stride, which is fine for 16-bit weights going through
// the flat kernels above anyway but left fp32 at 140 us (compress) and 688 us (decompress: 32 scalar stores per lane) for
// 304 MB at 8192^2.  The quotient: quant_core's reciprocal + half-integer test (fp32 has no proven shortcut; the test is exact).
template <int XDT, bool HAS_ZP>
__global__ __launch_bounds__(kBlock) void w4_quant_pack_f32_kernel(W4Params p, int sdt) {
    // XDT = CT_F32, or a 16-bit weight divided by a float32 scale (torch promotes the quotient to float32: T = float32)
    constexpr int U = 2;
    constexpr int V = XDT == CT_F32 ? 2 : 1;  // 16-byte vectors per unit
    const u32x4* in = static_cast<const u32x4*>(p.x);
    uint32_t* out = static_cast<uint32_t*>(p.out);
    const int64_t base = (int64_t)blockIdx.x * (U * kBlock) + threadIdx.x;
    u32x4 a[U][V];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u < p.units) {
#pragma unroll
            for (int v = 0; v < V; ++v) a[i][v] = in[V * u + v];
        }
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u >= p.units) continue;
        const int64_t si = w4_scale_index(p, u);
        const float s = load_rt(p.scale, sdt, si);
        const float z = HAS_ZP ? load_rt(p.zp, p.zdt, si) : 0.0f;  // zp.to(float32): exact
        const float rs = f32_fast_rcp(s);
        float xs[8];
        if constexpr (XDT == CT_F32) {
            const uint32_t ws[8] = {a[i][0].x, a[i][0].y, a[i][0].z, a[i][0].w, a[i][1].x, a[i][1].y, a[i][1].z, a[i][1].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) xs[k] = bits_f(ws[k]);
        } else {
            const uint32_t ws[4] = {a[i][0].x, a[i][0].y, a[i][0].z, a[i][0].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) unpack2<XDT>(ws[k], xs[2 * k], xs[2 * k + 1]);
        }
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float t = quant_core<CT_F32>(xs[k], s, HAS_ZP, z, -8.0f, 7.0f, rs);
            word |= (uint32_t)((cvt_i32_hw(t) + 8) & 15) << (4 * k);  // NaN -> code 0
        }
        __builtin_nontemporal_store(word, out + u);
    }
}

// decompress to fp32 (scales and output fp32).  A lane takes HALF a word (4 codes -> one 16-byte store), so that a store instruction
// covers 1 KB without gaps: with a whole word per lane the two stores of a lane interleaved at 32 bytes and every 32-byte sector
// was written twice, half each time (110-135 us at 8192^2 instead of ~55).
template <bool HAS_ZP>
__global__ __launch_bounds__(kBlock) void w4_unpack_dequant_f32_kernel(W4Params p) {
    constexpr int U = 4;
    const uint32_t* in = static_cast<const uint32_t*>(p.x);
    u32x4* out = static_cast<u32x4*>(p.out);
    const int64_t halves = 2 * p.units;
    const int64_t base = (int64_t)blockIdx.x * (U * kBlock) + threadIdx.x;
    uint32_t w[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t h = base + (int64_t)i * kBlock;
        w[i] = h < halves ? in[h >> 1] : 0u;
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t h = base + (int64_t)i * kBlock;
        if (h >= halves) continue;
        const int64_t si = w4_scale_index(p, h >> 1);
        const float s = static_cast<const float*>(p.scale)[si];
        const float z = HAS_ZP ? load_rt(p.zp, p.zdt, si) : 0.0f;
        const uint32_t c4 = (w[i] >> (16 * (int)(h & 1))) & 0xffffu;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float q = (float)((int)((c4 >> (4 * k)) & 15u) - 8);
            v[k] = dequant_core<CT_F32>(q, HAS_ZP, z, s);
        }
        stream_store16(out + h, u32x4{f_bits(v[0]), f_bits(v[1]), f_bits(v[2]), f_bits(v[3])});
    }
}

// fp32 quantize (-> int8 codes) / dequantize (int8 -> fp32) / fake_quantize, INT kinds: a lane takes FOUR consecutive elements — 16 bytes
// of floats and 4 bytes of codes — so that every load and store instruction of a wave is contiguous (the unit kernels give a lane 8
// floats = two 16-byte accesses interleaved with its neighbour's: 88 / 68 / 123 us at 8192^2 for 335 / 335 / 537 MB).  Four such
// quads per lane, a block apart, all loads first.
enum { F32_Q = 0, F32_DQ = 1, F32_FQ = 2 };
template <int MODE, bool HAS_ZP, bool FP8 = false /* F32_DQ only: the codes are float8_e4m3fn values */>
__global__ __launch_bounds__(kBlock) void f32_quads_kernel(W4Params p, int sdt, float qmin, float qmax, uint32_t xoff /* 0x80808080: codes stored + 128 (8-bit packed words) */) {
    constexpr int U = MODE == F32_DQ ? 8 : 4;  // dequantize loads only 4 bytes per quad: more of them in flight
    const int64_t quads = 2 * p.units;
    const int64_t base = (int64_t)blockIdx.x * (U * kBlock) + threadIdx.x;
    u32x4 a[U];
    uint32_t c[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t h = base + (int64_t)i * kBlock;
        if (h < quads) {
            if constexpr (MODE == F32_DQ) c[i] = static_cast<const uint32_t*>(p.x)[h];
            else a[i] = static_cast<const u32x4*>(p.x)[h];
        }
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t h = base + (int64_t)i * kBlock;
        if (h >= quads) continue;
        const int64_t si = w4_scale_index(p, h >> 1);
        const float s = load_rt(p.scale, sdt, si);
        const float z = HAS_ZP ? load_rt(p.zp, p.zdt, si) : 0.0f;  // zp.to(float32): exact
        float v[4];
        if constexpr (MODE == F32_DQ && FP8) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 q01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)c[i], false), q23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)c[i], true);  // x_q.to(float32): exact
            const float q[4] = {q01.x, q01.y, q23.x, q23.y};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = dequant_core<CT_F32>(q[k], HAS_ZP, z, s);
            stream_store16(static_cast<u32x4*>(p.out) + h, u32x4{f_bits(v[0]), f_bits(v[1]), f_bits(v[2]), f_bits(v[3])});
        } else if constexpr (MODE == F32_DQ) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = dequant_core<CT_F32>((float)(int)(int8_t)((c[i] ^ xoff) >> (8 * k)), HAS_ZP, z, s);
            stream_store16(static_cast<u32x4*>(p.out) + h, u32x4{f_bits(v[0]), f_bits(v[1]), f_bits(v[2]), f_bits(v[3])});
        } else {
            const float rs = f32_fast_rcp(s);
            const uint32_t ws[4] = {a[i].x, a[i].y, a[i].z, a[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = quant_core<CT_F32>(bits_f(ws[k]), s, HAS_ZP, z, qmin, qmax, rs);
            if constexpr (MODE == F32_Q) {
                uint32_t word = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) word |= (uint32_t)(cvt_i32_hw(v[k]) & 255) << (8 * k);  // NaN -> 0
                __builtin_nontemporal_store(word ^ xoff, static_cast<uint32_t*>(p.out) + h);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = dequant_core<CT_F32>(v[k], HAS_ZP, z, s);
                stream_store16(static_cast<u32x4*>(p.out) + h, u32x4{f_bits(v[0]), f_bits(v[1]), f_bits(v[2]), f_bits(v[3])});
            }
        }
    }
}

// fp32 -> int8 codes: a lane takes a whole unit (8 floats, two 16-byte loads, one 8-byte store) — the scale work is paid once per 8
// elements and the store is twice as wide as in the quads form (68 -> see DESIGN 5.2 us at 8192^2)
// FP8 = true: float32 -> float8_e4m3fn (FLOAT 8-bit, quant_args.py:463-486): the IEEE float32 quotient, the zero-point add, the clamp to +-448 and the
// hardware's round-to-nearest-even conversion (exact on a clamped value; NaN stays NaN) — the any-layout kernel ran these shapes at 94-96 us for 8192^2
template <bool HAS_ZP, bool FP8 = false>
__global__ __launch_bounds__(kBlock) void f32_quant_units_kernel(W4Params p, int sdt, float qmin, float qmax) {
    constexpr int U = 2;
    const u32x4* in = static_cast<const u32x4*>(p.x);
    u32x2* out = static_cast<u32x2*>(p.out);
    const int64_t base = (int64_t)blockIdx.x * (U * kBlock) + threadIdx.x;
    u32x4 a[U][2];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u < p.units) {
            a[i][0] = in[2 * u];
            a[i][1] = in[2 * u + 1];
        }
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u >= p.units) continue;
        const int64_t si = w4_scale_index(p, u);
        const float s = load_rt(p.scale, sdt, si);
        const float z = HAS_ZP ? load_rt(p.zp, p.zdt, si) : 0.0f;
        const uint32_t ws[8] = {a[i][0].x, a[i][0].y, a[i][0].z, a[i][0].w, a[i][1].x, a[i][1].y, a[i][1].z, a[i][1].w};
        if constexpr (FP8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                t[k] = bits_f(ws[k]) / s;
                if (HAS_ZP) t[k] += z;  // kept for z == 0: a -0 quotient becomes +0, as upstream's `+=`
                t[k] = clamp_nan(t[k], -448.0f, 448.0f);
            }
            int c0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false), c1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], 0, false);
            c0 = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], c0, true);
            c1 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], c1, true);
            stream_store8(out + u, u32x2{(uint32_t)c0, (uint32_t)c1});
            continue;
        }
        const float rs = f32_fast_rcp(s);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo |= (uint32_t)(cvt_i32_hw(quant_core<CT_F32>(bits_f(ws[k]), s, HAS_ZP, z, qmin, qmax, rs)) & 255) << (8 * k);  // NaN -> 0
            hi |= (uint32_t)(cvt_i32_hw(quant_core<CT_F32>(bits_f(ws[4 + k]), s, HAS_ZP, z, qmin, qmax, rs)) & 255) << (8 * k);
        }
        stream_store8(out + u, u32x2{lo, hi});
    }
}

// lean compress body for the common layout (flat scale index = lane >> gshift, one scale per lane, int8
// zero point): no grid-stride loop, no generic index arithmetic, and the scale / zero point are loaded
// BEFORE the 64 bytes of weights so that the reciprocal is ready when they land.  30.3 -> 29.4 us at 8192^2.
// (Round 3, for the 4096^2 launches: F = 2 / 3 / 4 consecutive chunks per workgroup with every chunk's loads requested up
// front measured 9.9 / 11.7 / 12.1 us against 8.8 us for this one-chunk form, and 31.0 / 32.7 / 34.2 against 29.4 us at 8192^2;
// block sizes 128 ... 1024 and the traffic-only stand-in all sit at 8.4-8.5 us: the exact grid of small workgroups stays.)
template <int DT, bool HAS_ZP>
__device__ __forceinline__ void w4_quant_pack_lean(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale,
                                                   const int8_t* __restrict__ zp, u32x4* __restrict__ out, int64_t g, int gshift) {
    const int64_t si = g >> gshift;
    const uint32_t sbits = __builtin_nontemporal_load(scale + si);
    const float z = HAS_ZP ? (float)__builtin_nontemporal_load(zp + si) : 0.0f;  // int8 -> exact in bf16 / fp16
    asm volatile("" ::: "memory");  // keep the small loads ahead of the big ones
    u32x4 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = in[g * 4 + i];
    const float s = DT == CT_BF16 ? bf16_bits_to_f(sbits) : f16_bits_to_f(sbits);
    const bool fast = fast_scale_ok<DT>(s) && fast_data_ok<DT, 4>(r);
    const float rs = 1.0f / s;
    const bool use_zp = HAS_ZP && (__builtin_amdgcn_ballot_w64(z != 0.0f) != 0);
    uint32_t w[4];
    if (fast) {
        if (use_zp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, true, true>(r[i], s, rs, z);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, true, false>(r[i], s, rs, z);
        }
    } else {
        if (use_zp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, false, true>(r[i], s, rs, z);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, false, false>(r[i], s, rs, z);
        }
    }
    stream_store16(out + g, u32x4{w[0], w[1], w[2], w[3]});
}

template <int DT, bool HAS_ZP>
__global__ __launch_bounds__(kBlock) void w4_quant_pack_lean_kernel(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale,
                                                                    const int8_t* __restrict__ zp, u32x4* __restrict__ out, int64_t groups,
                                                                    int gshift /* log2(lanes per scale group) */) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g < groups) w4_quant_pack_lean<DT, HAS_ZP>(in, scale, zp, out, g, gshift);
}

// ------------------------------------------------------------------------------------------
// Round-to-nearest compress in ONE pass ("minmax -> scale -> quantize -> pack"): the min-max
// observer + calculate_qparams (ct_qparams.hip) and the lean W4 compress above share their lane layout exactly — a
// lane owns 4 consecutive units (32 elements, 64 B in flight) and a group of 32 * LPG elements is LPG adjacent
// lanes — so the weight is read once: local min / max, DPP reduction inside the group, every lane of the group
// evaluates the (identical) scale / zero point, quantizes its own 32 elements against it and stores one 16-byte
// word vector; lane 0 of the group stores the scale and the zero point.  169 MB instead of 136 + 169 MB of
// traffic for 8192^2 g128, one launch instead of two.  Results are bit-identical to ct_minmax_qparams followed by
// ct_quant_pack by construction (same helpers), and tested against that composition.
// ------------------------------------------------------------------------------------------
template <int DT, bool SYM>
__global__ __launch_bounds__(kBlock) void rtn_w4_kernel(const u32x4* __restrict__ in, int64_t lanes, int lpg, u32x4* __restrict__ out,
                                                        void* __restrict__ scale_out, int8_t* __restrict__ zp_out) {
    constexpr int symmetric = SYM ? 1 : 0;
    const int64_t l = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // lanes is a multiple of lpg, kBlock too: groups never straddle blocks
    const bool live = l < lanes;
    u32x4 r[4];
    MinMax m;
    m.mn = __builtin_inff(); m.mx = -__builtin_inff(); m.nan = 0;
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = in[l * 4 + i];
    }
    if constexpr (SYM) {
        // symmetric: max |x| on the raw bit pairs (ct_minmax.h): 1 VALU per element for the observer part
        uint32_t acc = 0;
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, r[i].x), r[i].y), r[i].z), r[i].w);
        }
        m = absmax_finish<DT>(absmax_group_reduce(acc, lpg));
    } else {
        // asymmetric (round 6): min and max on the raw pairs as order-preserving int16 keys (ct_minmax.h) — the float form (unpack, NaN test, fmin,
        // fmax: 5 VALU per element) made this kernel 37 us at 8192^2 against 31 for the symmetric one
        MinMaxKey k = mmk_init();
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) k = mmk_acc(mmk_acc(mmk_acc(mmk_acc(k, r[i].x), r[i].y), r[i].z), r[i].w);
        }
        m = mmk_finish<DT>(mmk_group_reduce(k, lpg));
    }
    if (!live) return;
    float s, z;
    compute_qparams<DT>(m, 4, symmetric, s, z);
    if ((threadIdx.x & (lpg - 1)) == 0) {
        store1<DT>(scale_out, l / lpg, s);
        zp_out[l / lpg] = (int8_t)(int)z;
    }
    const bool fast = fast_scale_ok<DT>(s) && fast_data_ok<DT, 4>(r);
    const float rs = 1.0f / s;
    const bool use_zp = !symmetric && (__builtin_amdgcn_ballot_w64(z != 0.0f) != 0);
    uint32_t w[4];
    if (fast) {
        if (use_zp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, true, true>(r[i], s, rs, z);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, true, false>(r[i], s, rs, z);
        }
    } else {
        if (use_zp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, false, true>(r[i], s, rs, z);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = w4_quant_word<DT, false, false>(r[i], s, rs, z);
        }
    }
    stream_store16(out + l, u32x4{w[0], w[1], w[2], w[3]});
}

// value of lane 0 of each 16-lane row in every lane of the row (DPP row_newbcast:0)
__device__ __forceinline__ uint32_t row_leader_value(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150, 0xf, 0xf, false);
}

// UNROLL units per lane, one block apart, starting at `base`.
// ROWLEAD (flat scale index, >= 16 units per scale group, units % 16 == 0, int8 zero points): the 16 lanes of a DPP row
// hold 16 consecutive units of ONE group, so only the row's first lane loads the scale (and the zero point) and a
// v_mov_b32_dpp row_newbcast hands it to the other 15.  The kernel issues one vector-memory instruction per 4-byte
// word, 2-byte scale and 1-byte zero point — three per 8 elements in the asymmetric case, and that instruction rate,
// not the byte rate, is what made asymmetric decompress slower than symmetric (37 vs 29 us at 8192^2).  (One lane per WAVE fetching
// the wave's four scales / zero points and a scalar broadcast measured slower again: 36.2 vs 34.4 us.)
// SM = 2 (round 5; groups of exactly 128 elements = 16 units, units % 64 == 0, scale 8-byte and zero point 4-byte aligned): a wave's 64
// consecutive units are FOUR consecutive groups, so their four 16-bit scales are one aligned 8-byte run and their four int8 zero points one
// aligned dword at a wave-uniform address — two SCALAR loads (s_load_dwordx2 / s_load_dword through the constant address space: the scalar
// data cache, no vector-memory instruction at all), a lane picks its group's entry with one shift (v_lshrrev_b64 by 16 x (lane >> 4)) resp.
// one v_bfe_i32.  Vector-memory instructions per 8 elements: 2 (word in, 16 bytes out) instead of 4 (asymmetric) / 3 (symmetric).
constexpr int kW4ScalePerLane = 0, kW4ScaleRowLead = 1, kW4ScaleScalar = 2;
// ZPACKED (round 6; SM = 2 only, and a wave never straddles a row: units per row % 64 == 0): the zero points are read from their STORED form —
// the wave's row r and first group come from ONE scalar multiply-high (p.g_magic), its four groups' words (row r / 8) are one scalar
// 16-byte load, nibble r % 8 of each is cut out and re-based on the SCALAR unit, and the four bytes take the place of the int8 dword the
// unpacked form would have delivered: not one vector instruction more than the int8 path, and no unpack launch in front of this one.
template <int DT, int UNROLL, bool HAS_ZP, int SM = kW4ScalePerLane, bool ZPACKED = false>
__device__ __forceinline__ void w4_unpack_dequant_units(const W4Params& p, int64_t base) {
    static_assert(!ZPACKED || (HAS_ZP && SM == kW4ScaleScalar), "packed zero points ride the scalar-load form");
    constexpr bool ROWLEAD = SM == kW4ScaleRowLead;
    const uint32_t* in = static_cast<const uint32_t*>(p.x);
    uint32_t word[UNROLL];
    uint64_t s4[UNROLL];  // SM = 2: the wave's four scales / zero points, requested AHEAD of the words (scalar loads are counted by lgkmcnt,
    uint32_t z4[UNROLL];  // the words by vmcnt: neither waits for the other)
    if constexpr (SM == kW4ScaleScalar) {
        typedef const __attribute__((address_space(4))) uint32_t* const_u32_t;
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u0 = base + (int64_t)i * kBlock - (int64_t)(threadIdx.x & 63);  // the wave's first unit: a multiple of 64 (units % 64 == 0: in range or not as a wave)
            const uint32_t si_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u0 >> 4));
            const uint32_t si_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((u0 >> 4) >> 32));
            int64_t si0 = (int64_t)(((uint64_t)si_hi << 32) | si_lo);  // wave-uniform, a multiple of 4
            const int64_t si_last = (p.units >> 4) - 4;                 // unconditional (a branch would put a wait between these loads and the
            si0 = si0 < si_last ? si0 : si_last;                        // words'): a wave beyond the tensor reads the last entries and drops them
            const_u32_t sp = (const_u32_t)(uintptr_t)(static_cast<const uint16_t*>(p.scale) + si0);
            s4[i] = ((uint64_t)sp[1] << 32) | sp[0];
            z4[i] = 0;
            if constexpr (HAS_ZP && ZPACKED) {
                const uint32_t n = (uint32_t)si0;                                            // < 2^27 (the entry refuses units >= 2^31)
                const uint32_t row = (uint32_t)(((uint64_t)n * p.g_magic) >> p.g_shift);     // wave-uniform: scalar multiplies
                const uint32_t g0 = n - row * (uint32_t)p.scale_cols;                        // the wave's first group, a multiple of 4
                const_u32_t zw = (const_u32_t)(uintptr_t)(p.zpk + (size_t)(row >> 3) * (size_t)p.scale_cols + g0);
                const uint32_t sh = (row & 7u) * 4u;
                // (nibble - 8) as a byte each, in the order the int8 dword would have had
                z4[i] = ((((zw[0] >> sh) & 15u) - 8u) & 0xffu) | (((((zw[1] >> sh) & 15u) - 8u) & 0xffu) << 8) | (((((zw[2] >> sh) & 15u) - 8u) & 0xffu) << 16) |
                        (((((zw[3] >> sh) & 15u) - 8u) & 0xffu) << 24);
            } else if constexpr (HAS_ZP) {
                z4[i] = *(const_u32_t)(uintptr_t)(static_cast<const int8_t*>(p.zp) + si0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u < p.units) word[i] = in[u];
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u >= p.units) continue;
        const int64_t si = SM == kW4ScaleScalar ? 0 : w4_scale_index(p, u);
        float s, z;
        if constexpr (SM == kW4ScaleScalar) {
            const uint32_t g16 = (threadIdx.x & 48u);  // 16 x (lane >> 4)
            const uint32_t sb = (uint32_t)(s4[i] >> g16);
            s = DT == CT_BF16 ? bits_f(sb << 16) : f16_bits_to_f(sb & 0xffffu);
            z = HAS_ZP ? (float)(int)__builtin_amdgcn_sbfe(z4[i], g16 >> 1, 8) : 0.0f;  // (the builtin returns unsigned: without the cast -3 became 4294967293.0)
        } else if constexpr (ROWLEAD) {
            uint32_t sb = 0, zb = 0;
            if ((threadIdx.x & 15) == 0) {
                sb = static_cast<const uint16_t*>(p.scale)[si];
                if (HAS_ZP) zb = (uint32_t)(int)static_cast<const int8_t*>(p.zp)[si];
            }
            sb = row_leader_value(sb);
            s = DT == CT_BF16 ? bf16_bits_to_f(sb) : f16_bits_to_f(sb);
            z = HAS_ZP ? (float)(int)row_leader_value(zb) : 0.0f;
        } else {
            s = load_as_f<DT>(p.scale, si);
            z = HAS_ZP ? round_to<DT>(load_rt(p.zp, p.zdt, si)) : 0.0f;  // zp.to(scale.dtype)
        }
        float v[8];
        if (HAS_ZP && (SM != kW4ScalePerLane || p.zdt == CT_I8)) {  // wave-uniform
            // (2^23 + nibble) - (2^23 + 8 + z) is exact: the un-bias and the zero point cost ONE subtract, as in the symmetric case
            const float off = 8388616.0f + z;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = mul_round_to<DT>(bits_f(0x4B000000u | ((word[i] >> (4 * k)) & 0xfu)) - off, s);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // code - 8 as a float without v_cvt_f32_i32: 0x4B000000 | nibble is 2^23 + nibble exactly
                const float q = bits_f(0x4B000000u | ((word[i] >> (4 * k)) & 0xfu)) - 8388616.0f;
                v[k] = dequant_core<DT>(q, HAS_ZP, z, s);
            }
        }
        store8<DT>(p.out, u * 8, v);
    }
}

template <int DT, int UNROLL, bool HAS_ZP, int SM = kW4ScalePerLane>
__global__ __launch_bounds__(kBlock) void w4_unpack_dequant_kernel(W4Params p) {
    const int64_t stride = (int64_t)gridDim.x * kBlock * UNROLL;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; base < p.units; base += stride)
        w4_unpack_dequant_units<DT, UNROLL, HAS_ZP, SM>(p, base);
}

// ------------------------------------------------------------------------------------------
// batched W4A16: ONE launch over a table of tensors (a whole checkpoint shard).  A TinyLlama-1.1B
// shard is 154 modules, some as small as 1 MB: launched one by one the job is bound by the launch
// path (~5 us per module from the host, ~1.5 us of kernel boundary on the device), not by HBM.
// Every workgroup finds its tensor by a binary search over the table's running block count
// (wave-uniform scalar loads), then runs the same per-lane body as the single-tensor kernels.
// ------------------------------------------------------------------------------------------
constexpr int kBatchUnroll = 2;
constexpr int kBatchIter = 4;  // chunks per workgroup in the batched 8-bit decompress
// chunks per workgroup in the batched W4 decompress.  Round 5, with the scalar-load scale fetch (tools/time_batch.py, TinyLlama's 154 modules /
// 1 x 8192^2 / 16 x 2048^2 / 1024 x 256^2): 1 chunk 442 / 28.9 / 35.4 / 51.4 us, 2 chunks **402** / 30.3 / 32.9 / 42.2, 4 chunks (rounds 2-4)
// 426 / 33.6 / 33.9 / 40.2, 8 chunks 449 / 34.2 / 35.7 / 41.1: the table search (a chain of ~8 dependent scalar loads for 154 items) wants to be
// paid rarely, the chunks of a workgroup run one after the other and want to be few
constexpr int kW4BatchIter = 2;

__device__ __forceinline__ const ct_w4_item& batch_find(const ct_w4_item* __restrict__ items, int n, int64_t block) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= block) lo = mid; else hi = mid - 1;
    }
    return items[lo];
}

// groups per row of an item (a shift for the power-of-two group sizes; the division runs for the others only)
__device__ __forceinline__ int64_t item_groups_per_row(const ct_w4_item& it) {
    return it.upg_shift >= 0 ? ((it.cols >> 3) >> it.upg_shift) : (it.upg > 0 ? (it.cols >> 3) / it.upg : 1);
}

__device__ __forceinline__ W4Params batch_params(const ct_w4_item& it) {
    W4Params p;
    p.x = it.src; p.scale = it.scale; p.zp = it.zp; p.out = it.dst; p.zdt = CT_I8;
    p.units = it.units; p.upr = it.cols >> 3; p.rdiv = 1; p.scale_cols = it.zp_packed ? item_groups_per_row(it) : 0;
    p.upg_shift = it.upg_shift; p.upg = it.upg; p.flat_scale = 1;
    p.zpk = static_cast<const uint32_t*>(it.zp_packed); p.g_magic = it.g_magic; p.g_shift = it.g_shift;
    return p;
}

// the zero-point tail of an item (round 6): the workgroups behind the item's `main_blocks`, one lane per STORED word (r8, g) =
// sum_k ((zp[8 r8 + k][g] + 8) & 15) << 4k — pack_to_int32(zp, 4, packed_dim=0): rows beyond the matrix contribute 0 nibbles (the pad
// is applied AFTER the offset, helpers.py:65-67).  PACK: int8 -> words (compress); else words -> int8 (decompress writes back what
// unpack_from_int32(..., packed_dim=0) would have produced, base.py:147-153).  Consecutive lanes = consecutive groups of one row octet.
template <bool PACK>
__device__ __forceinline__ void w4_zp_tail(const ct_w4_item& it, int64_t tail_block) {
    const int64_t G = item_groups_per_row(it);
    const int64_t total = ((it.rows + 7) >> 3) * G;
    const int64_t wi = tail_block * kBlock + threadIdx.x;
    if (wi >= total) return;
    const int64_t r8 = wi < ((int64_t)1 << 31) ? (int64_t)(((uint64_t)(uint32_t)wi * it.g_magic) >> it.g_shift) : wi / G;
    const int64_t g = wi - r8 * G;
    if (PACK) {
        const int8_t* zp = static_cast<const int8_t*>(it.zp);
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t r = r8 * 8 + k;
            if (r < it.rows) word |= ((uint32_t)((int)zp[r * G + g] + 8) & 15u) << (4 * k);
        }
        static_cast<uint32_t*>(it.zp_packed)[wi] = word;
    } else {
        const uint32_t word = static_cast<const uint32_t*>(it.zp_packed)[wi];
        int8_t* zp = static_cast<int8_t*>(const_cast<void*>(it.zp));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t r = r8 * 8 + k;
            if (r < it.rows) zp[r * G + g] = (int8_t)((int)((word >> (4 * k)) & 15u) - 8);
        }
    }
}

template <int DT>
__device__ __forceinline__ void w4_quant_pack_item(const ct_w4_item& it, int64_t local_block) {
    if (local_block >= it.main_blocks) {  // workgroup-uniform
        w4_zp_tail<true>(it, local_block - it.main_blocks);
        return;
    }
    const W4Params p = batch_params(it);
    const int64_t g = local_block * kBlock + threadIdx.x;
    if (g >= p.units / 4) return;
    if (it.upg_shift >= 2) {  // power-of-two group of at least 32 columns: the lean body
        const int gshift = it.upg_shift - 2;
        if (p.zp) w4_quant_pack_lean<DT, true>(static_cast<const u32x4*>(p.x), static_cast<const uint16_t*>(p.scale), static_cast<const int8_t*>(p.zp), static_cast<u32x4*>(p.out), g, gshift);
        else w4_quant_pack_lean<DT, false>(static_cast<const u32x4*>(p.x), static_cast<const uint16_t*>(p.scale), nullptr, static_cast<u32x4*>(p.out), g, gshift);
        return;
    }
    if (p.zp) w4_quant_pack_group<DT, true, true>(p, g);
    else w4_quant_pack_group<DT, false, true>(p, g);
}

template <int DT>
__global__ __launch_bounds__(kBlock) void w4_quant_pack_batch_kernel(const ct_w4_item* __restrict__ items, int n) {
    const ct_w4_item& it = batch_find(items, n, blockIdx.x);
    w4_quant_pack_item<DT>(it, (int64_t)blockIdx.x - it.first_block);
}

// the same body on ONE item handed over by value (ct_quant_pack_w4_zp): no table in device memory
template <int DT>
__global__ __launch_bounds__(kBlock) void w4_quant_pack_one_kernel(const ct_w4_item it) {
    w4_quant_pack_item<DT>(it, (int64_t)blockIdx.x);
}

// Written as a runtime-stride loop because hipcc schedules this body markedly better inside one (31.5 us
// vs 37.1 us for one 8192^2 item with a single trip; the single-tensor kernel has the same shape) —
// measured, not understood; tools/time_batch.py reproduces it.
template <int DT>
__device__ __forceinline__ void w4_unpack_dequant_item(const ct_w4_item& it, int64_t local_block, int64_t stride) {
    if (local_block >= it.main_blocks) {  // workgroup-uniform: only items with zp_packed AND zp have such blocks
        w4_zp_tail<false>(it, local_block - it.main_blocks);
        return;
    }
    const W4Params p = batch_params(it);
    // a workgroup owns kW4BatchIter consecutive chunks of kBlock * kBatchUnroll units (the table search is
    // paid once per 1024 units); `stride` == chunk size, `limit` ends the walk after kW4BatchIter chunks
    const int64_t first = local_block * kBlock * kBatchUnroll * kW4BatchIter;
    const int64_t limit = (first + stride * kW4BatchIter < p.units) ? first + stride * kW4BatchIter : p.units;
    if (p.zpk) {  // zero points in their stored form (the plan admitted the item: groups of 128, whole waves per row, aligned tables)
        for (int64_t b = first + threadIdx.x; b < limit; b += stride) w4_unpack_dequant_units<DT, kBatchUnroll, true, kW4ScaleScalar, true>(p, b);
        return;
    }
    // round 5: an item with groups of 128 (upg_shift == 4), units % 64 == 0 and aligned scale / zero-point tables takes the SCALAR-load form
    // (a batch is always a launch of many residency rounds); the branch is workgroup-uniform
    const bool scalar = it.upg_shift == 4 && (p.units & 63) == 0 && (reinterpret_cast<uintptr_t>(p.scale) & 7u) == 0 && (reinterpret_cast<uintptr_t>(p.zp) & 3u) == 0;
    if (scalar) {
        if (p.zp) {
            for (int64_t b = first + threadIdx.x; b < limit; b += stride) w4_unpack_dequant_units<DT, kBatchUnroll, true, kW4ScaleScalar>(p, b);
        } else {
            for (int64_t b = first + threadIdx.x; b < limit; b += stride) w4_unpack_dequant_units<DT, kBatchUnroll, false, kW4ScaleScalar>(p, b);
        }
    } else if (p.zp) {
        for (int64_t b = first + threadIdx.x; b < limit; b += stride) w4_unpack_dequant_units<DT, kBatchUnroll, true>(p, b);
    } else {
        for (int64_t b = first + threadIdx.x; b < limit; b += stride) w4_unpack_dequant_units<DT, kBatchUnroll, false>(p, b);
    }
}

template <int DT>
__global__ __launch_bounds__(kBlock) void w4_unpack_dequant_batch_kernel(const ct_w4_item* __restrict__ items, int n, int64_t stride) {
    const ct_w4_item& it = batch_find(items, n, blockIdx.x);
    w4_unpack_dequant_item<DT>(it, (int64_t)blockIdx.x - it.first_block, stride);
}

template <int DT>
__global__ __launch_bounds__(kBlock) void w4_unpack_dequant_one_kernel(const ct_w4_item it, int64_t stride) {
    w4_unpack_dequant_item<DT>(it, (int64_t)blockIdx.x, stride);
}


// ------------------------------------------------------------------------------------------
// 8-bit storage fast paths (same stream-of-units shape as W4): int8 codes of
// Naive/IntQuantizationCompressor (OFF = 0, any num_bits <= 8 clamped to [qmin, qmax]) and the 8-bit
// pack-quantized words (OFF = 128: four (code + 128) bytes per int32, helpers.py:53-96 at b = 8).
//   quantize:   lane = 2 consecutive units (32 B in, one 16 B store out)
//   dequantize: lane = UNROLL units one block apart (8 B in, 16 B out each; 1 KiB per wave store)
// The generic kernels pay an IEEE divide per element and runtime dtype switches: 24.8 / 16.9 us for
// the 4096^2 per-tensor case of BASELINE config 1 against a ~9 us traffic floor.
// ------------------------------------------------------------------------------------------
template <int DT, bool FAST, bool ZP>
__device__ __forceinline__ void q8_quant_words(const u32x4& raw, float s, float rs, float z, int qmin, int qmax, uint32_t& lo, uint32_t& hi) {
    if constexpr (DT == CT_F16 && FAST) {  // packed fp16 pipeline; 1536: the low byte of each half is the two's-complement code
        uint32_t u[4];
        quant_pairs_f16<ZP, 1536>(raw, s, rs, z, (float)qmin, (float)qmax, u);
        lo = __builtin_amdgcn_perm(u[1], u[0], 0x06040200u) ^ 0x80808080u;  // the callers expect code + 128 per byte
        hi = __builtin_amdgcn_perm(u[3], u[2], 0x06040200u) ^ 0x80808080u;
        return;
    }
    const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t acc[2] = {0x80808080u, 0x80808080u};  // sum of (128 + code) << 8k never carries between bytes
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
        c0 = c0 < qmin ? qmin : (c0 > qmax ? qmax : c0);  // v_med3_i32
        c1 = c1 < qmin ? qmin : (c1 > qmax ? qmax : c1);
        acc[j >> 1] += (uint32_t)c0 << (16 * (j & 1));
        acc[j >> 1] += (uint32_t)c1 << (16 * (j & 1) + 8);
    }
    lo = acc[0]; hi = acc[1];
}

// lane g = units 2g, 2g + 1 (32 B in, one 16 B store out)
template <int DT, bool HAS_ZP, bool SHARED, int OFF>
__device__ __forceinline__ void q8_quant_lane(const W4Params& p, int64_t g, int qmin, int qmax) {
    constexpr int Q = 2;
    constexpr uint32_t kFlip = OFF == 0 ? 0x80808080u : 0u;  // (code + 128) ^ 0x80 == two's-complement code
    const u32x4* in = static_cast<const u32x4*>(p.x);
    u32x4* out = static_cast<u32x4*>(p.out);
    u32x4 r[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) r[i] = in[g * Q + i];
    uint32_t w[2 * Q];
    float s = 0.0f, z = 0.0f, rs = 0.0f;
    bool fast = false, use_zp = false;
    const bool data_ok = fast_data_ok<DT, Q>(r);
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        if (i == 0 || !SHARED) {
            const int64_t si = w4_scale_index(p, g * Q + i);
            s = load_as_f<DT>(p.scale, si);
            z = HAS_ZP ? round_to<DT>(load_rt(p.zp, p.zdt, si)) : 0.0f;
            fast = fast_scale_ok<DT>(s) && data_ok;
            rs = 1.0f / s;
            use_zp = HAS_ZP && (__builtin_amdgcn_ballot_w64(z != 0.0f) != 0);
        }
        if (fast) {
            if (use_zp) q8_quant_words<DT, true, true>(r[i], s, rs, z, qmin, qmax, w[2 * i], w[2 * i + 1]);
            else q8_quant_words<DT, true, false>(r[i], s, rs, z, qmin, qmax, w[2 * i], w[2 * i + 1]);
        } else {
            if (use_zp) q8_quant_words<DT, false, true>(r[i], s, rs, z, qmin, qmax, w[2 * i], w[2 * i + 1]);
            else q8_quant_words<DT, false, false>(r[i], s, rs, z, qmin, qmax, w[2 * i], w[2 * i + 1]);
        }
    }
    stream_store16(out + g, u32x4{w[0] ^ kFlip, w[1] ^ kFlip, w[2] ^ kFlip, w[3] ^ kFlip});
}

template <int DT, bool HAS_ZP, bool SHARED, int OFF>
__global__ __launch_bounds__(kBlock) void q8_quant_kernel(W4Params p, int qmin, int qmax) {
    const int64_t groups = p.units / 2;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock)
        q8_quant_lane<DT, HAS_ZP, SHARED, OFF>(p, g, qmin, qmax);
}

// UNROLL units per lane, one block apart, starting at `base` (8 B in, 16 B out each)
template <int DT, int UNROLL, bool HAS_ZP, int OFF>
__device__ __forceinline__ void q8_dequant_units(const W4Params& p, int64_t base) {
    constexpr uint32_t kFlip = OFF == 0 ? 0x80808080u : 0u;
    const u32x2* in = static_cast<const u32x2*>(p.x);
    u32x2 word[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u < p.units) word[i] = in[u];
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u >= p.units) continue;
        const int64_t si = w4_scale_index(p, u);
        const float s = load_as_f<DT>(p.scale, si);
        const float z = HAS_ZP ? round_to<DT>(load_rt(p.zp, p.zdt, si)) : 0.0f;
        const uint32_t ws[2] = {word[i].x ^ kFlip, word[i].y ^ kFlip};
        float v[8];
        if (HAS_ZP && p.zdt == CT_I8) {  // wave-uniform
            const float off = 128.0f + z;  // exact: the un-bias and the zero point in one subtract
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = mul_round_to<DT>((float)((ws[k >> 2] >> (8 * (k & 3))) & 0xffu) - off, s);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float q = (float)((ws[k >> 2] >> (8 * (k & 3))) & 0xffu) - 128.0f;  // v_cvt_f32_ubyteN
                v[k] = dequant_core<DT>(q, HAS_ZP, z, s);
            }
        }
        store8<DT>(p.out, u * 8, v);
    }
}

template <int DT, int UNROLL, bool HAS_ZP, int OFF>
__global__ __launch_bounds__(kBlock) void q8_dequant_kernel(W4Params p) {
    const int64_t stride = (int64_t)gridDim.x * kBlock * UNROLL;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; base < p.units; base += stride)
        q8_dequant_units<DT, UNROLL, HAS_ZP, OFF>(p, base);
}


// ------------------------------------------------------------------------------------------
// FLOAT 8-bit (float-quantized / mxfp8-quantized weights): same flat unit stream as the int8 kernels.
//   quantize:   lane = 2 units (32 B in, 16 B out): t = rnd_T(x / s) [+ zp], clamp to +-448, v_cvt_pk_fp8_f32
//   dequantize: lane = UNROLL units one block apart (8 B in, 16 B out each): v_cvt_pk_f32_fp8, (q - z) * s
// ------------------------------------------------------------------------------------------
// fp16 weights -> float8 (round 6): everything after the fp32 quotient on fp16 PAIRS, as quant_pairs_f16 does for the integer codes — reciprocal + one
// Newton step as two v_pk_fma_f32, v_cvt_pk_f16_f32 is the rounding to T, the zero-point add is v_pk_add_f16, the clamp v_pk_max_f16 / v_pk_min_f16.  The
// scalar form below (fast_quotient's selects and its sub-2^-13 branch, four roundings and two NaN-propagating clamps per element: ~20 VALU per element) held
// the FP8 quantize of fp16 weights at 51-57 us for 8192^2 against 35 for bf16 (profiles/r06_shape_sweep_floats.txt).  Preconditions, tested per 16-byte unit
// on the raw pairs (f8_f16_unit_ok): every element finite (inf / NaN: the Newton correction would turn inf into NaN), and zero or at least 2^-13 |s| in
// magnitude — below that the shortcut's quotient may differ from the IEEE one in the last fp16-subnormal place (ct_selftest_f16_div), i.e. be -0 where the
// divide gives -2^-24, and `+ zero_point` turns the first into +0 and leaves the second negative.  A -0 weight needs no care HERE (the Newton step returns +0
// for it, and -0 + z == +0 + z); WITHOUT a zero point the sign of a zero result is the weight's, so a unit that holds a -0 takes the scalar form too.
template <bool ZP>
__device__ __forceinline__ bool f8_f16_unit_ok(const u32x4& raw, float s) {
    // |x| >= thr  <=>  bits(|x|) >= bits(thr) for non-negative fp16; thr = 2^-13 |s| rounded to fp16 and moved one place up
    const uint32_t thr = f_to_f16_bits(__builtin_fabsf(s) * 0x1p-13f) + 1u;  // >= 1; |s| <= 2^15 on the fast path: no overflow
    const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
    u16x2_t mx = {0, 0}, mn = {0xffff, 0xffff}, nz = {0xffff, 0xffff};
    const u16x2_t one = {1, 1};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u16x2_t xa = __builtin_bit_cast(u16x2_t, ws[j] & 0x7fff7fffu);
        mx = __builtin_elementwise_max(mx, xa);
        mn = __builtin_elementwise_min(mn, xa - one);  // wraps: a zero becomes 0xffff and passes
        if (!ZP) nz = __builtin_elementwise_min(nz, __builtin_bit_cast(u16x2_t, ws[j] ^ 0x80008000u));  // 0 for a -0
    }
    const uint32_t hi16 = mx.x > mx.y ? mx.x : mx.y, lo16 = mn.x < mn.y ? mn.x : mn.y;
    return hi16 <= 0x7bffu && lo16 >= thr - 1u && (ZP || (nz.x != 0 && nz.y != 0));
}

template <bool ZP>
__device__ __forceinline__ void f8_quant_words_f16(const u32x4& raw, float s, float rs, float z, uint32_t& lo, uint32_t& hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 rs2 = {rs, rs}, s2 = {s, s};
    const qh2_t lo2 = {(_Float16)-448.0f, (_Float16)-448.0f}, hi2 = {(_Float16)448.0f, (_Float16)448.0f}, z2 = {(_Float16)z, (_Float16)z};
    const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
    int acc[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f2 x = __builtin_convertvector(__builtin_bit_cast(qh2_t, ws[j]), f2);
        f2 t = x * rs2;
        t = __builtin_elementwise_fma(__builtin_elementwise_fma(-t, s2, x), rs2, t);
        qh2_t t16 = __builtin_convertvector(t, qh2_t);
        if (ZP) t16 = t16 + z2;
        t16 = __builtin_elementwise_min(__builtin_elementwise_max(t16, lo2), hi2);
        const f2 c = __builtin_convertvector(t16, f2);
        if (j & 1) acc[j >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(c.x, c.y, acc[j >> 1], true);
        else acc[j >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(c.x, c.y, acc[j >> 1], false);
    }
    lo = (uint32_t)acc[0]; hi = (uint32_t)acc[1];
}

template <int DT, bool FAST, bool ZP>
__device__ __forceinline__ void f8_quant_words(const u32x4& raw, float s, float rs, float z, uint32_t& lo, uint32_t& hi) {
    if constexpr (DT == CT_F16 && FAST) {
        if (f8_f16_unit_ok<ZP>(raw, s)) {  // lanes almost always agree
            f8_quant_words_f16<ZP>(raw, s, rs, z, lo, hi);
            return;
        }
    }
    const uint32_t ws[4] = {raw.x, raw.y, raw.z, raw.w};
    int acc[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x0, x1;
        unpack2<DT>(ws[j], x0, x1);
        float t0 = fast_quotient<DT>(x0, s, FAST ? rs : 0.0f), t1 = fast_quotient<DT>(x1, s, FAST ? rs : 0.0f);
        round2<DT>(t0, t1);
        if (ZP) {
            t0 += z; t1 += z;
            round2<DT>(t0, t1);
        }
        t0 = clamp_nan(t0, -448.0f, 448.0f);
        t1 = clamp_nan(t1, -448.0f, 448.0f);
        if (j & 1) acc[j >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(t0, t1, acc[j >> 1], true);
        else acc[j >> 1] = __builtin_amdgcn_cvt_pk_fp8_f32(t0, t1, acc[j >> 1], false);
    }
    lo = (uint32_t)acc[0]; hi = (uint32_t)acc[1];
}

template <int DT, bool HAS_ZP, bool SHARED>
__device__ __forceinline__ void f8_quant_lane(const W4Params& p, int64_t g) {
    constexpr int Q = 2;
    const u32x4* in = static_cast<const u32x4*>(p.x);
    u32x4* out = static_cast<u32x4*>(p.out);
    u32x4 r[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) r[i] = in[g * Q + i];
    uint32_t w[2 * Q];
    float s = 0.0f, z = 0.0f, rs = 0.0f;
    bool fast = false;
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        if (i == 0 || !SHARED) {
            const int64_t si = w4_scale_index(p, g * Q + i);
            s = load_as_f<DT>(p.scale, si);
            z = HAS_ZP ? round_to<DT>(load_rt(p.zp, p.zdt, si)) : 0.0f;
            fast = fast_scale_ok<DT>(s);  // fp16: reciprocal + Newton step, exact fix-up below 2^-13 (fast_quotient)
            rs = 1.0f / s;
        }
        // the zero-point add is kept even for z == 0: it turns a -0.0 quotient into +0.0, as upstream's `+=`
        if (fast) f8_quant_words<DT, true, HAS_ZP>(r[i], s, rs, z, w[2 * i], w[2 * i + 1]);
        else f8_quant_words<DT, false, HAS_ZP>(r[i], s, rs, z, w[2 * i], w[2 * i + 1]);
    }
    stream_store16(out + g, u32x4{w[0], w[1], w[2], w[3]});
}

template <int DT, bool HAS_ZP, bool SHARED>
__global__ __launch_bounds__(kBlock) void f8_quant_kernel(W4Params p) {
    const int64_t groups = p.units / 2;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) f8_quant_lane<DT, HAS_ZP, SHARED>(p, g);
}

template <int DT, int UNROLL, bool HAS_ZP>
__device__ __forceinline__ void f8_dequant_units(const W4Params& p, int64_t base) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const u32x2* in = static_cast<const u32x2*>(p.x);
    u32x2 word[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u < p.units) word[i] = in[u];
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u >= p.units) continue;
        const int64_t si = w4_scale_index(p, u);
        const float s = load_as_f<DT>(p.scale, si);
        const float z = HAS_ZP ? round_to<DT>(load_rt(p.zp, p.zdt, si)) : 0.0f;
        const f2 q01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)word[i].x, false), q23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)word[i].x, true);
        const f2 q45 = __builtin_amdgcn_cvt_pk_f32_fp8((int)word[i].y, false), q67 = __builtin_amdgcn_cvt_pk_f32_fp8((int)word[i].y, true);
        const float q[8] = {q01.x, q01.y, q23.x, q23.y, q45.x, q45.y, q67.x, q67.y};
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = dequant_core<DT>(q[k], HAS_ZP, z, s);
        store8<DT>(p.out, u * 8, v);
    }
}

template <int DT, int UNROLL, bool HAS_ZP>
__global__ __launch_bounds__(kBlock) void f8_dequant_kernel(W4Params p) {
    const int64_t stride = (int64_t)gridDim.x * kBlock * UNROLL;
    for (int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; base < p.units; base += stride) f8_dequant_units<DT, UNROLL, HAS_ZP>(p, base);
}

// ------------------------------------------------------------------------------------------
// batched 8-bit codecs: ONE launch over a table of tensors (the W8A8 / FP8 counterpart of the W4 batch above; the
// per-module loop of ModelCompressor, model_compressor.py:167-169,196-198, with Naive / Int / FloatQuantizationCompressor,
// naive_quantized/base.py:48-126).  Same table type; `group` = number of consecutive elements of the row-major stream that
// share one scale: cols (channel), a divisor of cols (group) or rows * cols (tensor).  Zero points: int8 or none.
// ------------------------------------------------------------------------------------------
// the 8-bit tables' reading of an item: as batch_params, plus the block strategy (group < 0 in the caller's table: scale[r / bh][c / bw], the FP8-block
// checkpoints' layout, forward.py:198-216).  ct_q8_batch_plan leaves log2(bh) + 1 in `main_blocks` (0: not a block item; the 8-bit tables have no
// zero-point tail) and the multiply-high for n / (units per row) in g_magic / g_shift; bh and bw are powers of two, so the other two quotients are shifts.
__device__ __forceinline__ W4Params q8_batch_params(const ct_w4_item& it) {
    W4Params p = batch_params(it);
    if (it.main_blocks > 0) {
        const int L = (int)it.main_blocks - 1;
        p.flat_scale = 0;
        p.nf_fast = 1;
        p.upr_magic = it.g_magic; p.upr_shift = it.g_shift;
        p.rdiv = (int64_t)1 << L; p.rdiv_magic = 0x80000000u; p.rdiv_shift = 31 + L;  // n >> L as a multiply-high
        p.scale_cols = p.upr >> p.upg_shift;
    }
    return p;
}

template <int DT, bool FP8, int OFF = 0 /* 128: the codes are stored + 128, four to an int32 word (pack-quantized 8-bit, pack_to_int32) */>
__global__ __launch_bounds__(kBlock) void q8_quant_batch_kernel(const ct_w4_item* __restrict__ items, int n, int qmin, int qmax, int zdt) {
    const ct_w4_item& it = batch_find(items, n, blockIdx.x);
    W4Params p = q8_batch_params(it);
    p.zdt = zdt;  // int8, or the float8 zero points a calibrated FLOAT scheme carries (round 6)
    const int64_t g = ((int64_t)blockIdx.x - it.first_block) * kBlock + threadIdx.x;
    if (g >= p.units / 2) return;
    if constexpr (FP8) {
        if (p.zp) f8_quant_lane<DT, true, true>(p, g);
        else f8_quant_lane<DT, false, true>(p, g);
    } else {
        if (p.zp) q8_quant_lane<DT, true, true, OFF>(p, g, qmin, qmax);
        else q8_quant_lane<DT, false, true, OFF>(p, g, qmin, qmax);
    }
}

template <int DT, bool FP8, int OFF = 0>
__global__ __launch_bounds__(kBlock) void q8_dequant_batch_kernel(const ct_w4_item* __restrict__ items, int n, int64_t stride, int zdt) {
    const ct_w4_item& it = batch_find(items, n, blockIdx.x);
    W4Params p = q8_batch_params(it);
    p.zdt = zdt;
    const int64_t first = ((int64_t)blockIdx.x - it.first_block) * kBlock * kBatchUnroll * kBatchIter;
    const int64_t limit = (first + stride * kBatchIter < p.units) ? first + stride * kBatchIter : p.units;
    // (a runtime-stride loop, like the W4 batch: hipcc schedules the body better inside one)
    if (p.zp) {
        for (int64_t b = first + threadIdx.x; b < limit; b += stride) {
            if constexpr (FP8) f8_dequant_units<DT, kBatchUnroll, true>(p, b);
            else q8_dequant_units<DT, kBatchUnroll, true, OFF>(p, b);
        }
    } else {
        for (int64_t b = first + threadIdx.x; b < limit; b += stride) {
            if constexpr (FP8) f8_dequant_units<DT, kBatchUnroll, false>(p, b);
            else q8_dequant_units<DT, kBatchUnroll, false, OFF>(p, b);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Channel-wise 8-bit round-to-nearest in ONE pass (W8A8 int8 / FP8 weights: the per-output-channel min-max observer +
// calculate_qparams + quantize): one workgroup per row holds the row in registers (MAXU 16-byte units per thread,
// one block apart: every wave load reads 1 KiB contiguous), reduces min / max through DPP + LDS, every thread
// evaluates the identical scale and quantizes its own units.  2 + 1 B per element instead of 2 + (2 + 1).
// ------------------------------------------------------------------------------------------
template <int DT, int MAXU, bool FP8>
__global__ __launch_bounds__(kBlock) void rtn_channel8_kernel(const u32x4* __restrict__ in, int64_t rows, int upr /* units per row */, int symmetric,
                                                              u32x2* __restrict__ out, void* __restrict__ scale_out, int8_t* __restrict__ zp_out) {
    __shared__ float s_mn[kBlock / 64], s_mx[kBlock / 64];
    __shared__ int s_nan[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const u32x4* rin = in + row * upr;
        u32x4 r[MAXU];
        MinMax m;
        m.mn = __builtin_inff(); m.mx = -__builtin_inff(); m.nan = 0;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = i * kBlock + tid;
            if (u < upr) r[i] = rin[u];
        }
        if (FP8 || symmetric) {  // block-uniform: max |x| on the raw bit pairs (ct_minmax.h)
            uint32_t acc = 0;
#pragma unroll
            for (int i = 0; i < MAXU; ++i) {
                const int u = i * kBlock + tid;
                if (u < upr) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, r[i].x), r[i].y), r[i].z), r[i].w);
            }
            acc = absmax_group_reduce(acc, 64);
            if (lane == 0) s_mn[wave] = __builtin_bit_cast(float, acc);
            __syncthreads();
            uint32_t all = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) all = absmax_acc(all, __builtin_bit_cast(uint32_t, s_mn[w]));
            m = absmax_finish<DT>(all);
        } else {
#pragma unroll
            for (int i = 0; i < MAXU; ++i) {
                const int u = i * kBlock + tid;
                if (u < upr) {
                    const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float a, b;
                        unpack2<DT>(ws[j], a, b);
                        m.nan |= (a != a) | (b != b);
                        m.mn = __builtin_fminf(m.mn, __builtin_fminf(a, b));
                        m.mx = __builtin_fmaxf(m.mx, __builtin_fmaxf(a, b));
                    }
                }
            }
            m = group_reduce(m, 64);
            if (lane == 0) { s_mn[wave] = m.mn; s_mx[wave] = m.mx; s_nan[wave] = m.nan; }
            __syncthreads();
            m.mn = __builtin_fminf(__builtin_fminf(s_mn[0], s_mn[1]), __builtin_fminf(s_mn[2], s_mn[3]));
            m.mx = __builtin_fmaxf(__builtin_fmaxf(s_mx[0], s_mx[1]), __builtin_fmaxf(s_mx[2], s_mx[3]));
            m.nan = s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3];
        }
        float s, z = 0.0f;
        if constexpr (FP8) s = compute_qparams_float<DT>(m, QP_FP8, 1.0f);
        else compute_qparams<DT>(m, 8, symmetric, s, z);
        if (tid == 0) {
            store1<DT>(scale_out, row, s);
            if (zp_out) zp_out[row] = (int8_t)(int)z;
        }
        // (fp16: the float8 words use the scalar shortcut with its fix-up; the int8 words' packed-fp16 pipeline needs finite data, which
        // a row that produced a finite non-zero scale has — a NaN or inf in the row makes the scale NaN / inf and `fast` false)
        const bool fast = fast_scale_ok<DT>(s);
        const float rs = 1.0f / s;
        const bool use_zp = !FP8 && !symmetric && z != 0.0f;  // block-uniform
        u32x2* rout = out + row * upr;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = i * kBlock + tid;
            if (u >= upr) continue;
            uint32_t lo, hi;
            if constexpr (FP8) {
                // with the (all-zero) zero point of a calibrated scheme present, as in the compressor's call: -0.0 + 0 = +0.0
                if (fast) f8_quant_words<DT, true, true>(r[i], s, rs, 0.0f, lo, hi);
                else f8_quant_words<DT, false, true>(r[i], s, rs, 0.0f, lo, hi);
            } else {
                if (fast) { if (use_zp) q8_quant_words<DT, true, true>(r[i], s, rs, z, -128, 127, lo, hi); else q8_quant_words<DT, true, false>(r[i], s, rs, z, -128, 127, lo, hi); }
                else { if (use_zp) q8_quant_words<DT, false, true>(r[i], s, rs, z, -128, 127, lo, hi); else q8_quant_words<DT, false, false>(r[i], s, rs, z, -128, 127, lo, hi); }
                lo ^= 0x80808080u; hi ^= 0x80808080u;  // (code + 128) -> two's-complement code
            }
            stream_store8(rout + u, u32x2{lo, hi});
        }
        __syncthreads();  // the reduction scratch is reused by the next row
    }
}

// the same one-pass compress with one WAVE per row (rows of at most 64 * MAXU units): the row lives in the wave's registers
// (MAXU 16-byte loads in flight per lane, each wave load 1 KiB contiguous), the reduction is DPP + two wave shuffles — no LDS,
// no barrier, no phase in which a whole workgroup waits for its slowest wave.
template <int DT, int MAXU, bool FP8>
__global__ __launch_bounds__(kBlock) void rtn_channel8_wave_kernel(const u32x4* __restrict__ in, int64_t rows, int upr, int symmetric,
                                                                   u32x2* __restrict__ out, void* __restrict__ scale_out, int8_t* __restrict__ zp_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u32x4* rin = in + row * upr;
    u32x4 r[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = i * 64 + lane;
        if (u < upr) r[i] = rin[u];
    }
    MinMax m;
    m.mn = __builtin_inff(); m.mx = -__builtin_inff(); m.nan = 0;
    if (FP8 || symmetric) {  // wave-uniform
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = i * 64 + lane;
            if (u < upr) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, r[i].x), r[i].y), r[i].z), r[i].w);
        }
        m = absmax_finish<DT>(absmax_group_reduce(acc, 64));
    } else {
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = i * 64 + lane;
            if (u < upr) {
                const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a, b;
                    unpack2<DT>(ws[j], a, b);
                    m.nan |= (a != a) | (b != b);
                    m.mn = __builtin_fminf(m.mn, __builtin_fminf(a, b));
                    m.mx = __builtin_fmaxf(m.mx, __builtin_fmaxf(a, b));
                }
            }
        }
        m = group_reduce(m, 64);
    }
    float s, z = 0.0f;
    if constexpr (FP8) s = compute_qparams_float<DT>(m, QP_FP8, 1.0f);
    else compute_qparams<DT>(m, 8, symmetric, s, z);
    if (lane == 0) {
        store1<DT>(scale_out, row, s);
        if (zp_out) zp_out[row] = (int8_t)(int)z;
    }
    const bool fast = fast_scale_ok<DT>(s);  // fp16: see rtn_channel8_kernel
    const float rs = 1.0f / s;
    const bool use_zp = !FP8 && !symmetric && z != 0.0f;  // wave-uniform
    u32x2* rout = out + row * upr;
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = i * 64 + lane;
        if (u >= upr) continue;
        uint32_t lo, hi;
        if constexpr (FP8) {
            if (fast) f8_quant_words<DT, true, true>(r[i], s, rs, 0.0f, lo, hi);
            else f8_quant_words<DT, false, true>(r[i], s, rs, 0.0f, lo, hi);
        } else {
            if (fast) { if (use_zp) q8_quant_words<DT, true, true>(r[i], s, rs, z, -128, 127, lo, hi); else q8_quant_words<DT, true, false>(r[i], s, rs, z, -128, 127, lo, hi); }
            else { if (use_zp) q8_quant_words<DT, false, true>(r[i], s, rs, z, -128, 127, lo, hi); else q8_quant_words<DT, false, false>(r[i], s, rs, z, -128, 127, lo, hi); }
            lo ^= 0x80808080u; hi ^= 0x80808080u;
        }
        stream_store8(rout + u, u32x2{lo, hi});
    }
}

// fake_quantize fast path (forward_helpers.py:180-215): x, scale and the result share one 16-bit dtype.
// Same flat unit stream: lane = UNROLL units one block apart, 16 B in, 16 B out.
template <int DT, int UNROLL, bool HAS_ZP>
__global__ __launch_bounds__(kBlock) void fq16_kernel(W4Params p, float qmin, float qmax, int fkind) {
    const int64_t stride = (int64_t)gridDim.x * kBlock * UNROLL;
    const u32x4* in = static_cast<const u32x4*>(p.x);
    for (int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x; base < p.units; base += stride) {
        u32x4 raw[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u = base + (int64_t)i * kBlock;
            if (u < p.units) raw[i] = in[u];
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t u = base + (int64_t)i * kBlock;
            if (u >= p.units) continue;
            const int64_t si = w4_scale_index(p, u);
            const float s = load_as_f<DT>(p.scale, si);
            const float z = HAS_ZP ? round_to<DT>(load_rt(p.zp, p.zdt, si)) : 0.0f;  // zp.to(x.dtype) == zp.to(scale.dtype) here
            const float rs = DT == CT_BF16 ? bf16_fast_rcp(s) : f16_newton_rcp(s);
            const uint32_t ws[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0, x1;
                unpack2<DT>(ws[j], x0, x1);
                const float t0 = quant_core<DT>(x0, s, HAS_ZP, z, qmin, qmax, rs, fkind), t1 = quant_core<DT>(x1, s, HAS_ZP, z, qmin, qmax, rs, fkind);
                v[2 * j] = dequant_core<DT>(t0, HAS_ZP, z, s);
                v[2 * j + 1] = dequant_core<DT>(t1, HAS_ZP, z, s);
            }
            store8<DT>(p.out, u * 8, v);
        }
    }
}

// fake_quantize, INT codes, flat scale layout (scale index = unit >> ushift), int8 or no zero point: the lean form — exact grid, the
// scale / zero-point loads ahead of the 16-byte loads, a finite-data test per unit (sum of squares, one dot2 per pair: the float
// result has to carry a NaN through, which the packed path below does not), arithmetic on float PAIRS (v_pk_mul_f32 / v_pk_add_f32,
// one v_cvt_pk per rounding).  6 (symmetric) / 8.5 VALU per element against ~18 in fq16_kernel: 54.3 -> see DESIGN 5.2.
template <int DT>
__device__ __forceinline__ float fq_sumsq(uint32_t d, float acc) {
    if constexpr (DT == CT_BF16) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, d), __builtin_bit_cast(b2, d), acc, false);
    } else {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(qh2_t, d), __builtin_bit_cast(qh2_t, d), acc, false);
    }
}

template <int DT, bool HAS_ZP>
__device__ __forceinline__ uint32_t fq16_pair(uint32_t d, float s, float rs, float z, float qmin, float qmax) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float x0, x1;
    unpack2<DT>(d, x0, x1);
    float t0, t1;
    if constexpr (DT == CT_BF16) {
        const f2 t = f2{x0, x1} * f2{rs, rs};
        t0 = t.x; t1 = t.y;
    } else {
        t0 = fast_quotient<CT_F16>(x0, s, rs);
        t1 = fast_quotient<CT_F16>(x1, s, rs);
    }
    round2<DT>(t0, t1);
    if (HAS_ZP) {
        const f2 t = f2{t0, t1} + f2{z, z};
        t0 = t.x; t1 = t.y;
        round2<DT>(t0, t1);
    }
    // finite by construction: v_med3_f32 is the clamp
    t0 = __builtin_rintf(__builtin_amdgcn_fmed3f(t0, qmin, qmax));
    t1 = __builtin_rintf(__builtin_amdgcn_fmed3f(t1, qmin, qmax));
    f2 q = {t0, t1};
    if (HAS_ZP) q = q - f2{z, z};  // |q - z| <= 255: exact in bf16 / fp16, the reference's rounding of the difference is the identity
    q = q * f2{s, s};
    if constexpr (DT == CT_BF16) {
        typedef bf16_t b2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(q, b2));
    } else {
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(q, qh2_t));
    }
}

template <int DT, bool HAS_ZP>
__global__ __launch_bounds__(kBlock) void fq16_lean_kernel(const u32x4* __restrict__ in, const uint16_t* __restrict__ scale, const int8_t* __restrict__ zp,
                                                           u32x4* __restrict__ out, int64_t units, int ushift, float qmin, float qmax) {
    constexpr int U = 2;
    const int64_t base = (int64_t)blockIdx.x * (U * kBlock) + threadIdx.x;
    uint32_t sbits[U];
    float z[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        const int64_t si = (u < units ? u : 0) >> ushift;
        sbits[i] = __builtin_nontemporal_load(scale + si);
        z[i] = HAS_ZP ? (float)__builtin_nontemporal_load(zp + si) : 0.0f;  // int8: exact in bf16 / fp16
    }
    asm volatile("" ::: "memory");  // keep the small loads ahead of the big ones
    u32x4 raw[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u < units) raw[i] = in[u];
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        if (u >= units) continue;
        const float s = DT == CT_BF16 ? bf16_bits_to_f(sbits[i]) : f16_bits_to_f(sbits[i]);
        const uint32_t ws[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fq_sumsq<DT>(ws[j], acc);
        const bool fast = fast_scale_ok<DT>(s) && acc <= 3.0e38f;  // every element finite (a huge finite one also takes the exact path)
        uint32_t o[4];
        if (fast) {
            const float rs = DT == CT_BF16 ? 1.0f / s : f16_newton_rcp(s);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fq16_pair<DT, HAS_ZP>(ws[j], s, rs, z[i], qmin, qmax);
        } else {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0, x1;
                unpack2<DT>(ws[j], x0, x1);
                const float t0 = quant_core<DT>(x0, s, HAS_ZP, z[i], qmin, qmax), t1 = quant_core<DT>(x1, s, HAS_ZP, z[i], qmin, qmax);
                v[2 * j] = dequant_core<DT>(t0, HAS_ZP, z[i], s);
                v[2 * j + 1] = dequant_core<DT>(t1, HAS_ZP, z[i], s);
            }
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (DT == CT_BF16) {
                    typedef bf16_t b2 __attribute__((ext_vector_type(2)));
                    o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v[2 * j], v[2 * j + 1]}, b2));
                } else {
                    o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{v[2 * j], v[2 * j + 1]}, qh2_t));
                }
            }
        }
        stream_store16(out + u, u32x4{o[0], o[1], o[2], o[3]});
    }
}

// ------------------------------------------------------------------------------------------
// diagnostics: exhaustive check of the reciprocal fast path
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void selftest_bf16_div_kernel(uint32_t s_lo, uint32_t s_hi,
                                                                   unsigned long long* mismatches) {
    unsigned long long local = 0;
    for (uint32_t sb = s_lo + blockIdx.x; sb < s_hi; sb += gridDim.x) {
        const float s = bf16_bits_to_f(sb);
        const float as = __builtin_fabsf(s);
        if (!((as >= 0x1p-64f) && (as <= 0x1p64f))) continue;  // outside the fast-path range
        const float rs = 1.0f / s;
        for (uint32_t xb = threadIdx.x; xb < 65536u; xb += kBlock) {
            const float x = bf16_bits_to_f(xb);
            const float a = round_to<CT_BF16>(x * rs);
            const float b = round_to<CT_BF16>(x / s);
            const bool an = a != a, bn = b != b;
            bool bad = (an != bn);
            if (!an && !bn && f_bits(a) != f_bits(b)) {
                // differing results below 2^-20 cannot change any integer code (they round to
                // 0 with or without a zero point); anything else is a real mismatch
                bad = (__builtin_fabsf(a) >= 0x1p-20f) || (__builtin_fabsf(b) >= 0x1p-20f);
            }
            local += bad ? 1ull : 0ull;
        }
    }
    if (local) atomicAdd(mismatches, local);
}

// can the flat W4 kernels take this call?
static bool w4_eligible(int dt, int sdt, int tdt_or_odt, int bits, int64_t rows, int64_t cols, int64_t rdiv,
                        int64_t cdiv, const int32_t* col_group, const void* a, const void* b) {
    if (bits != 4 || col_group) return false;
    if (!(dt == CT_BF16 || dt == CT_F16) || sdt != dt || tdt_or_odt != dt) return false;
    if (rows <= 0 || cols <= 0 || cols % 8 || cdiv % 8) return false;
    if (!(rdiv == 1 || rdiv >= rows)) {
        // block strategy rows: fine, handled by the non-flat index
    }
    return aligned16(a) && aligned16(b);
}

// can the flat 8-bit-storage kernels take this call?  (16-bit x / scale of one dtype, unit-aligned groups)
static bool q8_eligible(int dt, int sdt, int tdt_or_odt, int64_t rows, int64_t cols, int64_t cdiv, const int32_t* col_group,
                        const void* a, const void* b) {
    if (col_group) return false;
    if (!(dt == CT_BF16 || dt == CT_F16) || sdt != dt || tdt_or_odt != dt) return false;
    if (rows <= 0 || cols <= 0 || cols % 16 || (cdiv % 8 && cdiv < cols)) return false;
    return aligned16(a) && aligned16(b);
}

static W4Params make_w4(const void* x, const void* scale, const void* zp, int zdt, void* out, int64_t rows,
                        int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols) {
    W4Params p;
    p.x = x; p.scale = scale; p.zp = zp; p.out = out; p.zdt = zdt;
    p.units = rows * (cols / 8);
    p.upr = cols / 8;
    p.rdiv = rdiv; p.scale_cols = scale_cols;
    int64_t c = cdiv > cols ? cols : cdiv;  // a group wider than the row is the whole row
    p.upg = c / 8;
    p.upg_shift = log2_exact(p.upg);
    p.flat_scale = (rdiv == 1 && cols % c == 0 && scale_cols == cols / c) ? 1 : 0;
    if (rdiv >= rows && cdiv >= cols) {  // one scale for the whole tensor: index 0 without any division
        p.flat_scale = 1;
        p.upg_shift = 62;
    }
    if (!p.flat_scale && p.units < ((int64_t)1 << 31) && p.upr >= 1 && rdiv >= 1 && rdiv < ((int64_t)1 << 31) && p.upg >= 1) {
        // Granlund-Montgomery with N = 31: magic = ceil(2^(31 + L) / d), L = ceil(log2 d), magic < 2^32, exact for n < 2^31
        auto magic31 = [](int64_t d, uint32_t& magic, int& shift) {
            int L = 0;
            while (((int64_t)1 << L) < d) ++L;
            shift = 31 + L;
            magic = (uint32_t)((((uint64_t)1 << shift) + (uint64_t)(d - 1)) / (uint64_t)d);
        };
        magic31(p.upr, p.upr_magic, p.upr_shift);
        magic31(rdiv, p.rdiv_magic, p.rdiv_shift);
        magic31(p.upg, p.upg_magic, p.upg_mshift);
        p.nf_fast = 1;
    }
    return p;
}

static unsigned w4_grid(int64_t items, int unroll) {
    // exact grids measured faster than capped + grid-stride on MI355X (tools/kbench); the stride
    // loop only matters beyond 2^31 / kBlock workgroups
    int64_t g = cdiv64(items, (int64_t)kBlock * unroll);
    if (g > (int64_t)1 << 30) g = (int64_t)1 << 30;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// units per lane of the 8-bit dequantize kernels (8-byte loads, 16-byte stores, one block apart), in residency rounds of the chip's lanes — 256 CUs x 8
// workgroups of 256 lanes.  HBM-cold, real shapes, int8 and FP8 (tools/shape_sweep_floats.py with the rule overridden, profiles/r06_dequant8_units_per_lane.txt),
// U = 2 / 4 / 8: 3584^2 (3.1 rounds) 10.5 / 11.0 / 9.9 us, 8192 x 2048 (4) 11.1 / 11.4 / 10.0, 4096 x 6144 (6) 14.8 / 15.8 / 13.8, 5120^2 (6.25) 15.3 / 16.4 / 14.7,
// 8192 x 4096 (8) 18.9 / 19.9 / 20.8, 8192^2 (16) 33 / 34 / 31-35: eight units per lane while the tensor is 2 .. 8 rounds of lanes (fewer, longer workgroups
// fill the chip once), two beyond — and two whenever a lane's units do not share their scale (per-row groups: 5120^2 g128 15.5 / 16.3 / 16.5, asymmetric
// 15.9 / 16.5 / 19.1), or, with zero points, above six rounds (channel-wise asymmetric 5120^2 15.9 / 16.6 / 17.9).  (The round-3 rule — 2 / 4 / 8 at <= 2 / 4 / 8
// rounds — came from squares measured with their codes in the Infinity Cache: it cost 10 % at 3584^2 and at 8192 x 4096.)
static int decomp_unroll(int64_t units, bool per_row_groups, bool has_zp) {
    const int64_t round_lanes = (int64_t)kCUs * 8 * kBlock;
    if (per_row_groups || units <= 2 * round_lanes || units >= 8 * round_lanes) return 2;
    if (has_zp && units > 6 * round_lanes) return 2;
    return 8;
}
// The W4 and packed-W8 decompress (group scales: one scale load per unit): four units per lane only where they make the tensor ONE residency round
// (2 .. 4 rounds of lanes: 4096^2 9.8 -> 8.5 us); eight never pay — HBM-cold, round 6: 4096 x 6144 12.9 (U = 2) / 14.4 (4) / 14.0 (8) us, 5120^2
// 13.4 / 15.1 / 14.5, packed W8 5120^2 15.0 / 17.2 / 23.2 (117 VGPRs at U = 8) — decomp_unroll()'s mid band cost those kernels 10-35 %.
static int decomp_unroll_grouped(int64_t units) {
    const int64_t round_lanes = (int64_t)kCUs * 8 * kBlock;
    return (units > 2 * round_lanes && units <= 4 * round_lanes) ? 4 : 2;
}
#define CT_FOR_UNROLL(u, ...)                                   \
    do {                                                        \
        if ((u) == 8) { constexpr int U = 8; __VA_ARGS__; }      \
        else if ((u) == 4) { constexpr int U = 4; __VA_ARGS__; } \
        else { constexpr int U = 2; __VA_ARGS__; }               \
    } while (0)

// activation-ordered W4 (w4_gidx_rows_kernel): one dtype for weight / scale / result, row-wise scales, group table aligned
static bool w4_gidx_ok(int dt, int sdt, int tdt_or_odt, int bits, int64_t rows, int64_t cols, int64_t rdiv, int64_t scale_cols,
                       const int32_t* col_group, const void* a, const void* b) {
    return bits == 4 && col_group && (dt == CT_BF16 || dt == CT_F16) && sdt == dt && tdt_or_odt == dt && rows > 0 && cols > 0 && cols % 8 == 0 && rdiv == 1 &&
           scale_cols >= 1 && scale_cols <= kGidxMaxGroups && aligned16(col_group) && aligned16(a) && aligned16(b) &&
           cdiv64(rows, kGidxRows) * cdiv64(cols / 8, kBlock) < ((int64_t)1 << 31);
}
template <bool COMPRESS>
static int launch_w4_gidx(const W4Params& w, int dt, const void* zp, const int32_t* col_group, int64_t rows, ct_stream_t stream, const char* what) {
    const int chunks = (int)cdiv64(w.upr, (COMPRESS ? kGidxCompressUnits : kGidxDecompressUnits) * kBlock);
    dim3 g((unsigned)(cdiv64(rows, kGidxRows) * chunks));
    // the LDS tables are sized for 128 / 256 / 512 / 1024 groups per row (round 6: rows of 28672 columns — 224 groups — ran on the 1024-group tables at
    // 3-5 workgroups per CU)
#define CT_GIDXR_G(DT, ZP, MAXG_) hipLaunchKernelGGL((w4_gidx_rows_kernel<DT, ZP, COMPRESS, kGidxRows, (COMPRESS ? kGidxCompressUnits : kGidxDecompressUnits), MAXG_>), \
                                                     g, dim3(kBlock), 0, as_stream(stream), w, col_group, chunks, rows)
#define CT_GIDXR(DT, ZP) do { if (w.scale_cols <= kGidxSmallGroups) CT_GIDXR_G(DT, ZP, kGidxSmallGroups); else if (w.scale_cols <= 256) CT_GIDXR_G(DT, ZP, 256); \
                              else if (w.scale_cols <= 512) CT_GIDXR_G(DT, ZP, 512); else CT_GIDXR_G(DT, ZP, kGidxMaxGroups); } while (0)
    if (dt == CT_BF16) { if (zp) CT_GIDXR(CT_BF16, true); else CT_GIDXR(CT_BF16, false); }
    else { if (zp) CT_GIDXR(CT_F16, true); else CT_GIDXR(CT_F16, false); }
#undef CT_GIDXR
#undef CT_GIDXR_G
    return hip_check(hipGetLastError(), what);
}

// fp32 flat kernels (f32_quads_kernel): groups of a multiple of 8 columns (or the whole row), no g_idx, 16-byte aligned float side,
// 4-byte aligned code side
static bool f32_quads_ok(int64_t rows, int64_t cols, int64_t cdiv, const int32_t* col_group, const void* floats, const void* codes) {
    return !col_group && rows > 0 && cols > 0 && cols % 8 == 0 && (cdiv % 8 == 0 || cdiv >= cols) && aligned16(floats) &&
           (codes == nullptr || (reinterpret_cast<uintptr_t>(codes) & 3u) == 0) && rows * (cols / 8) < ((int64_t)1 << 38);
}
template <int MODE>
static int launch_f32_quads(const W4Params& w, const void* zp, int sdt, float qmin, float qmax, ct_stream_t stream, const char* what, uint32_t xoff = 0u) {
    dim3 g(w4_grid(2 * w.units, MODE == F32_DQ ? 8 : 4));
    if (zp) hipLaunchKernelGGL((f32_quads_kernel<MODE, true>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, qmin, qmax, xoff);
    else hipLaunchKernelGGL((f32_quads_kernel<MODE, false>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, qmin, qmax, xoff);
    return hip_check(hipGetLastError(), what);
}

}  // namespace ct

using namespace ct;

extern "C" {

static void set_float_kind(QParams& p, int fkind, const float* gscale) {
    p.fkind = fkind;
    if (fkind == 1) { p.qmin = -448.0f; p.qmax = 448.0f; }  // torch.finfo(float8_e4m3fn) (utils/helpers.py:212-214)
    if (fkind == 2) { p.qmin = -6.0f; p.qmax = 6.0f; }      // FP4_E2M1_DATA (utils/helpers.py:215-217)
    p.gscale = gscale;
    if (gscale) p.sdt_arith = CT_F32;
}

static int quantize_impl(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                         int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                         int bits, int fkind, int tdt, void* out, int odt, ct_stream_t stream, const float* gscale = nullptr) {
    CT_REQUIRE(gscale == nullptr || tdt == CT_F32, "a global scale makes the quotient float32; got result dtype %d", tdt);
    CT_REQUIRE(bits >= 1 && bits <= 8, "num_bits must be in [1, 8], got %d", bits);
    CT_REQUIRE(xt_ok(xdt, tdt), "unsupported (x dtype, result dtype) = (%d, %d)", xdt, tdt);
    if (fkind) CT_REQUIRE(odt == CT_F8E4M3 || is_float_dt(odt), "unsupported output dtype %d", odt);
    else CT_REQUIRE(odt == CT_I8 || odt == CT_I32 || is_float_dt(odt), "unsupported output dtype %d", odt);
    QParams p;
    int rc = fill_qparams(p, x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, bits, out, odt);
    if (rc) return rc;
    set_float_kind(p, fkind, gscale);
    if (rows == 0 || cols == 0) return CT_OK;
    if (!gscale && fkind == 0 && xdt == CT_F32 && tdt == CT_F32 && odt == CT_I8 && is_float_dt(sdt) && f32_quads_ok(rows, cols, cdiv, col_group, x, out)) {
        W4Params w = make_w4(x, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        if ((reinterpret_cast<uintptr_t>(out) & 7u) == 0) {
            dim3 g(w4_grid(w.units, 2));
            if (zp) hipLaunchKernelGGL((f32_quant_units_kernel<true>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, p.qmin, p.qmax);
            else hipLaunchKernelGGL((f32_quant_units_kernel<false>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, p.qmin, p.qmax);
            CT_LAUNCH_CHECK("ct_quantize[f32 units]");
        }
        return launch_f32_quads<F32_Q>(w, zp, sdt, p.qmin, p.qmax, stream, "ct_quantize[f32]");
    }
    if (!gscale && fkind == 1 && xdt == CT_F32 && tdt == CT_F32 && odt == CT_F8E4M3 && is_float_dt(sdt) && f32_quads_ok(rows, cols, cdiv, col_group, x, out) &&
        (reinterpret_cast<uintptr_t>(out) & 7u) == 0) {
        W4Params w = make_w4(x, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        dim3 g(w4_grid(w.units, 2));
        if (zp) hipLaunchKernelGGL((f32_quant_units_kernel<true, true>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, p.qmin, p.qmax);
        else hipLaunchKernelGGL((f32_quant_units_kernel<false, true>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, p.qmin, p.qmax);
        CT_LAUNCH_CHECK("ct_quantize[f32 -> fp8 units]");
    }
    if (!gscale && fkind != 2 && (fkind ? odt == CT_F8E4M3 : odt == CT_I8) && q8_eligible(xdt, sdt, tdt, rows, cols, cdiv, col_group, x, out)) {
        W4Params w = make_w4(x, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        const bool shared = (cdiv % 16 == 0) || cdiv >= cols;
        dim3 g8(w4_grid(w.units / 2, 1));
        const int qmin = -(1 << (bits - 1)), qmax = (1 << (bits - 1)) - 1;
#define CT_Q8Q(DT, ZP, SH) do { if (fkind) hipLaunchKernelGGL((f8_quant_kernel<DT, ZP, SH>), g8, dim3(kBlock), 0, as_stream(stream), w); \
                                else hipLaunchKernelGGL((q8_quant_kernel<DT, ZP, SH, 0>), g8, dim3(kBlock), 0, as_stream(stream), w, qmin, qmax); } while (0)
        if (xdt == CT_BF16) {
            if (zp) { if (shared) CT_Q8Q(CT_BF16, true, true); else CT_Q8Q(CT_BF16, true, false); }
            else { if (shared) CT_Q8Q(CT_BF16, false, true); else CT_Q8Q(CT_BF16, false, false); }
        } else {
            if (zp) { if (shared) CT_Q8Q(CT_F16, true, true); else CT_Q8Q(CT_F16, true, false); }
            else { if (shared) CT_Q8Q(CT_F16, false, true); else CT_Q8Q(CT_F16, false, false); }
        }
#undef CT_Q8Q
        CT_LAUNCH_CHECK("ct_quantize[flat8]");
    }
    dim3 grid = grid_2d(rows, cdiv64(cols, 8));
    CT_DISPATCH_XT(xdt, tdt, hipLaunchKernelGGL((quant_units_kernel<X, T, MODE_Q>), grid, dim3(kBlock), 0, as_stream(stream), p));
    CT_LAUNCH_CHECK("ct_quantize");
}

static int fake_quantize_impl(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                              int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                              int bits, int fkind, int tdt, void* out, int odt, ct_stream_t stream, const float* gscale = nullptr) {
    CT_REQUIRE(bits >= 1 && bits <= 8, "num_bits must be in [1, 8], got %d", bits);
    CT_REQUIRE(xt_ok(xdt, tdt), "unsupported (x dtype, result dtype) = (%d, %d)", xdt, tdt);
    CT_REQUIRE(is_float_dt(odt), "unsupported output dtype %d", odt);
    CT_REQUIRE(gscale == nullptr || tdt == CT_F32, "a global scale makes the quotient float32; got result dtype %d", tdt);
    QParams p;
    int rc = fill_qparams(p, x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, bits, out, odt);
    if (rc) return rc;
    set_float_kind(p, fkind, gscale);
    if (rows == 0 || cols == 0) return CT_OK;
    if (!gscale && p.fkind == 0 && xdt == CT_F32 && tdt == CT_F32 && odt == CT_F32 && sdt == CT_F32 && f32_quads_ok(rows, cols, cdiv, col_group, x, nullptr) && aligned16(out)) {
        W4Params w = make_w4(x, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        return launch_f32_quads<F32_FQ>(w, zp, sdt, p.qmin, p.qmax, stream, "ct_fake_quantize[f32]");
    }
    if (!gscale && odt == xdt && !col_group && (xdt == CT_BF16 || xdt == CT_F16) && sdt == xdt && tdt == xdt && cols % 8 == 0 &&
        (cdiv % 8 == 0 || cdiv >= cols) && aligned16(x) && aligned16(out)) {
        W4Params w = make_w4(x, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        constexpr int U = 2;
        if (p.fkind == 0 && w.flat_scale && w.upg_shift >= 0 && (!zp || zdt == CT_I8) && w.units < ((int64_t)1 << 38)) {
            dim3 gl((unsigned)cdiv64(w.units, (int64_t)kBlock * 2));
#define CT_FQL(DT, ZP) hipLaunchKernelGGL((fq16_lean_kernel<DT, ZP>), gl, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), \
                                           static_cast<const uint16_t*>(scale), static_cast<const int8_t*>(zp), static_cast<u32x4*>(out), w.units, w.upg_shift, p.qmin, p.qmax)
            if (xdt == CT_BF16) { if (zp) CT_FQL(CT_BF16, true); else CT_FQL(CT_BF16, false); }
            else { if (zp) CT_FQL(CT_F16, true); else CT_FQL(CT_F16, false); }
#undef CT_FQL
            CT_LAUNCH_CHECK("ct_fake_quantize[fq16 lean]");
        }
        dim3 gf(w4_grid(w.units, U));
#define CT_FQ(DT, ZP) hipLaunchKernelGGL((fq16_kernel<DT, U, ZP>), gf, dim3(kBlock), 0, as_stream(stream), w, p.qmin, p.qmax, p.fkind)
        if (xdt == CT_BF16) { if (zp) CT_FQ(CT_BF16, true); else CT_FQ(CT_BF16, false); }
        else { if (zp) CT_FQ(CT_F16, true); else CT_FQ(CT_F16, false); }
#undef CT_FQ
        CT_LAUNCH_CHECK("ct_fake_quantize[fq16]");
    }
    dim3 grid = grid_2d(rows, cdiv64(cols, 8));
    CT_DISPATCH_XT(xdt, tdt, hipLaunchKernelGGL((quant_units_kernel<X, T, MODE_FQ>), grid, dim3(kBlock), 0, as_stream(stream), p));
    CT_LAUNCH_CHECK("ct_fake_quantize");
}

int ct_quantize(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                int bits, int tdt, void* out, int odt, ct_stream_t stream) {
    return quantize_impl(x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, bits, 0, tdt, out, odt, stream);
}

int ct_fake_quantize(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                     int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                     int bits, int tdt, void* out, int odt, ct_stream_t stream) {
    return fake_quantize_impl(x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, bits, 0, tdt, out, odt, stream);
}

int ct_quantize_fp8(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                    int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                    int tdt, void* out, int odt, ct_stream_t stream) {
    return quantize_impl(x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, 8, 1, tdt, out, odt, stream);
}

int ct_fake_quantize_fp8(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                         int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                         int tdt, void* out, int odt, ct_stream_t stream) {
    return fake_quantize_impl(x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, 8, 1, tdt, out, odt, stream);
}

int ct_quantize_fp4(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                    int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                    const float* global_scale, int tdt, void* out, int odt, ct_stream_t stream) {
    CT_REQUIRE(is_float_dt(odt), "FLOAT 4-bit values are returned in a float dtype, got %d", odt);
    return quantize_impl(x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, 4, 2, tdt, out, odt, stream, global_scale);
}

int ct_fake_quantize_fp4(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                         int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                         const float* global_scale, int tdt, void* out, int odt, ct_stream_t stream) {
    return fake_quantize_impl(x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, 4, 2, tdt, out, odt, stream, global_scale);
}

static int dequantize_impl(const void* xq, int qdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                           int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                           void* out, int odt, ct_stream_t stream, const float* gscale);

int ct_dequantize(const void* xq, int qdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                  int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                  void* out, int odt, ct_stream_t stream) {
    return dequantize_impl(xq, qdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, out, odt, stream, nullptr);
}

int ct_dequantize_gs(const void* xq, int qdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                     int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                     const float* global_scale, void* out, int odt, ct_stream_t stream) {
    CT_REQUIRE(global_scale != nullptr, "ct_dequantize_gs needs a global scale");
    return dequantize_impl(xq, qdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, out, odt, stream, global_scale);
}

static int dequantize_impl(const void* xq, int qdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                           int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                           void* out, int odt, ct_stream_t stream, const float* gscale) {
    CT_REQUIRE(is_float_dt(odt), "unsupported output dtype %d", odt);
    CT_REQUIRE(qdt == CT_I8 || qdt == CT_I32 || qdt == CT_F8E4M3 || is_float_dt(qdt), "unsupported x_q dtype %d", qdt);
    QParams p;
    int rc = fill_qparams(p, xq, qdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, 8, out, odt);
    if (rc) return rc;
    if (rows == 0 || cols == 0) return CT_OK;
    p.vec = (cols % 8 == 0) && aligned16(out) && ((reinterpret_cast<uintptr_t>(xq) & 7u) == 0);
    set_float_kind(p, 0, gscale);
    if (!gscale && qdt == CT_I8 && sdt == CT_F32 && odt == CT_F32 && f32_quads_ok(rows, cols, cdiv, col_group, out, xq)) {
        W4Params w = make_w4(xq, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        return launch_f32_quads<F32_DQ>(w, zp, sdt, 0.0f, 0.0f, stream, "ct_dequantize[f32]");
    }
    if (!gscale && qdt == CT_F8E4M3 && sdt == CT_F32 && odt == CT_F32 && f32_quads_ok(rows, cols, cdiv, col_group, out, xq)) {
        W4Params w = make_w4(xq, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        dim3 g(w4_grid(2 * w.units, 8));
        if (zp) hipLaunchKernelGGL((f32_quads_kernel<F32_DQ, true, true>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, 0.0f, 0.0f, 0u);
        else hipLaunchKernelGGL((f32_quads_kernel<F32_DQ, false, true>), g, dim3(kBlock), 0, as_stream(stream), w, sdt, 0.0f, 0.0f, 0u);
        CT_LAUNCH_CHECK("ct_dequantize[fp8 -> f32]");
    }
    if (!gscale && (qdt == CT_I8 || qdt == CT_F8E4M3) && !col_group && (sdt == CT_BF16 || sdt == CT_F16) && odt == sdt && rows > 0 && cols % 8 == 0 &&
        (cdiv % 8 == 0 || cdiv >= cols) && aligned16(out) && (reinterpret_cast<uintptr_t>(xq) & 7u) == 0) {
        W4Params w = make_w4(xq, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        const int unroll = decomp_unroll(w.units, rdiv == 1 && cdiv < cols, zp != nullptr);
        dim3 g8(w4_grid(w.units, unroll));
#define CT_Q8D(DT, ZP) CT_FOR_UNROLL(unroll, if (qdt == CT_F8E4M3) hipLaunchKernelGGL((f8_dequant_kernel<DT, U, ZP>), g8, dim3(kBlock), 0, as_stream(stream), w); \
                                             else hipLaunchKernelGGL((q8_dequant_kernel<DT, U, ZP, 0>), g8, dim3(kBlock), 0, as_stream(stream), w))
        if (sdt == CT_BF16) { if (zp) CT_Q8D(CT_BF16, true); else CT_Q8D(CT_BF16, false); }
        else { if (zp) CT_Q8D(CT_F16, true); else CT_Q8D(CT_F16, false); }
#undef CT_Q8D
        CT_LAUNCH_CHECK("ct_dequantize[flat8]");
    }
    dim3 grid = grid_2d(rows, cdiv64(cols, 8));
    switch (p.sdt_arith) {
        case CT_BF16: hipLaunchKernelGGL((dequant_units_kernel<CT_BF16>), grid, dim3(kBlock), 0, as_stream(stream), p); break;
        case CT_F16: hipLaunchKernelGGL((dequant_units_kernel<CT_F16>), grid, dim3(kBlock), 0, as_stream(stream), p); break;
        default: hipLaunchKernelGGL((dequant_units_kernel<CT_F32>), grid, dim3(kBlock), 0, as_stream(stream), p); break;
    }
    CT_LAUNCH_CHECK("ct_dequantize");
}

int ct_quant_pack(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt, int64_t rows,
                  int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols, const int32_t* col_group,
                  int bits, int tdt, int32_t* packed, ct_stream_t stream) {
    CT_REQUIRE(bits >= 1 && bits <= 8, "Packing is only supported for num_bits in [1, 8], got %d", bits);
    CT_REQUIRE(xt_ok(xdt, tdt), "unsupported (x dtype, result dtype) = (%d, %d)", xdt, tdt);
    QParams p;
    int rc = fill_qparams(p, x, xdt, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, bits, packed, CT_I32);
    if (rc) return rc;
    if (rows == 0 || cols == 0) return CT_OK;
    if (w4_eligible(xdt, sdt, tdt, bits, rows, cols, rdiv, cdiv, col_group, x, packed) && cols % 32 == 0) {
        W4Params w = make_w4(x, scale, zp, zdt, packed, rows, cols, rdiv, cdiv, scale_cols);
        const bool shared = (cdiv % 32 == 0) || cdiv >= cols;  // 4 consecutive units share a scale
        dim3 grid(w4_grid(w.units / 4, 1));
        if (shared && w.flat_scale && w.upg_shift >= 2 && w.upg_shift < 62 && (zp == nullptr || zdt == CT_I8) && w.units / 4 < ((int64_t)1 << 38)) {
            const int gshift = w.upg_shift - 2;
            const int64_t groups = w.units / 4;
            dim3 gl((unsigned)cdiv64(groups, kBlock));
#define CT_W4L(DT, ZP) hipLaunchKernelGGL((w4_quant_pack_lean_kernel<DT, ZP>), gl, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), static_cast<const uint16_t*>(scale), static_cast<const int8_t*>(zp), reinterpret_cast<u32x4*>(packed), groups, gshift)
            if (xdt == CT_BF16) { if (zp) CT_W4L(CT_BF16, true); else CT_W4L(CT_BF16, false); }
            else { if (zp) CT_W4L(CT_F16, true); else CT_W4L(CT_F16, false); }
#undef CT_W4L
            CT_LAUNCH_CHECK("ct_quant_pack[w4 lean]");
        }
#define CT_W4Q(DT, ZP, SH) hipLaunchKernelGGL((w4_quant_pack_kernel<DT, ZP, SH>), grid, dim3(kBlock), 0, as_stream(stream), w)
        if (xdt == CT_BF16) {
            if (zp) { if (shared) CT_W4Q(CT_BF16, true, true); else CT_W4Q(CT_BF16, true, false); }
            else { if (shared) CT_W4Q(CT_BF16, false, true); else CT_W4Q(CT_BF16, false, false); }
        } else {
            if (zp) { if (shared) CT_W4Q(CT_F16, true, true); else CT_W4Q(CT_F16, true, false); }
            else { if (shared) CT_W4Q(CT_F16, false, true); else CT_W4Q(CT_F16, false, false); }
        }
#undef CT_W4Q
        CT_LAUNCH_CHECK("ct_quant_pack[w4]");
    }
    if (bits == 8 && xdt == CT_F32 && tdt == CT_F32 && is_float_dt(sdt) && f32_quads_ok(rows, cols, cdiv, col_group, x, packed)) {
        W4Params w = make_w4(x, scale, zp, zdt, packed, rows, cols, rdiv, cdiv, scale_cols);  // a word = four (code + 128) bytes = one quad
        return launch_f32_quads<F32_Q>(w, zp, sdt, -128.0f, 127.0f, stream, "ct_quant_pack[w8 f32]", 0x80808080u);
    }
    if (w4_gidx_ok(xdt, sdt, tdt, bits, rows, cols, rdiv, scale_cols, col_group, x, packed)) {
        W4Params w = make_w4(x, scale, zp, zdt, packed, rows, cols, rdiv, cols, scale_cols);
        return launch_w4_gidx<true>(w, xdt, zp, col_group, rows, stream, "ct_quant_pack[w4 g_idx]");
    }
    if (bits == 4 && tdt == CT_F32 && is_float_dt(xdt) && !col_group && is_float_dt(sdt) && cols % 8 == 0 && cdiv % 8 == 0 && aligned16(x) &&
        (reinterpret_cast<uintptr_t>(packed) & 3u) == 0 && rows * (cols / 8) < ((int64_t)1 << 38)) {
        W4Params w = make_w4(x, scale, zp, zdt, packed, rows, cols, rdiv, cdiv, scale_cols);
        dim3 gf(w4_grid(w.units, 2));
#define CT_W4F(X) do { if (zp) hipLaunchKernelGGL((w4_quant_pack_f32_kernel<X, true>), gf, dim3(kBlock), 0, as_stream(stream), w, sdt); \
                       else hipLaunchKernelGGL((w4_quant_pack_f32_kernel<X, false>), gf, dim3(kBlock), 0, as_stream(stream), w, sdt); } while (0)
        if (xdt == CT_F32) CT_W4F(CT_F32); else if (xdt == CT_BF16) CT_W4F(CT_BF16); else CT_W4F(CT_F16);
#undef CT_W4F
        CT_LAUNCH_CHECK("ct_quant_pack[w4 f32]");
    }
    if (bits == 8 && cols % 32 == 0 && q8_eligible(xdt, sdt, tdt, rows, cols, cdiv, col_group, x, packed)) {
        // 8-bit words are four (code + 128) bytes: the int8 stream kernel with OFF = 128
        W4Params w = make_w4(x, scale, zp, zdt, packed, rows, cols, rdiv, cdiv, scale_cols);
        const bool shared = (cdiv % 16 == 0) || cdiv >= cols;
        dim3 g8(w4_grid(w.units / 2, 1));
#define CT_Q8P(DT, ZP, SH) hipLaunchKernelGGL((q8_quant_kernel<DT, ZP, SH, 128>), g8, dim3(kBlock), 0, as_stream(stream), w, -128, 127)
        if (xdt == CT_BF16) {
            if (zp) { if (shared) CT_Q8P(CT_BF16, true, true); else CT_Q8P(CT_BF16, true, false); }
            else { if (shared) CT_Q8P(CT_BF16, false, true); else CT_Q8P(CT_BF16, false, false); }
        } else {
            if (zp) { if (shared) CT_Q8P(CT_F16, true, true); else CT_Q8P(CT_F16, true, false); }
            else { if (shared) CT_Q8P(CT_F16, false, true); else CT_Q8P(CT_F16, false, false); }
        }
#undef CT_Q8P
        CT_LAUNCH_CHECK("ct_quant_pack[w8]");
    }
    // round 6: the widths next to 4 and 8 in the common checkpoint layout (ct_quant_wb.hip)
    if (wb_layout_ok(xdt, sdt, tdt, bits, zdt, zp, rows, cols, rdiv, cdiv, scale_cols, col_group, x, packed))
        return launch_wb_quant_pack(x, xdt, scale, zp, rows, cols, cdiv, bits, packed, stream);
    p.vec = (cols % 8 == 0) && aligned16(x);
    const int64_t packed_cols = cdiv64(cols * bits, 32);
    dim3 grid = grid_2d(rows, cdiv64(cols, 32));
    return bits <= 4 ? launch_quant_pack_g32_lo(p, xdt, tdt, bits, packed_cols, grid, stream) : launch_quant_pack_g32_hi(p, xdt, tdt, bits, packed_cols, grid, stream);
}

int ct_rtn_quant_pack_w4(const void* x, int xdt, int64_t rows, int64_t cols, int64_t group, int symmetric, int32_t* packed, void* scale_out,
                         int8_t* zp_out, ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "the one-pass round-to-nearest compress takes 16-bit float weights, got dtype %d", xdt);
    CT_REQUIRE(rows >= 0 && cols >= 0 && group >= 1, "bad shape");
    CT_REQUIRE(group % 32 == 0 && group <= 2048 && log2_exact(group / 32) >= 0, "group size must be 32 * 2^k <= 2048, got %lld", (long long)group);
    CT_REQUIRE(cols % group == 0, "columns (%lld) must be a multiple of the group size %lld", (long long)cols, (long long)group);
    CT_REQUIRE(aligned16(x) && aligned16(packed), "buffers must be 16-byte aligned");
    CT_REQUIRE(scale_out && zp_out, "scale and zero-point outputs are required");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t lanes = rows * (cols / 32);
    const int lpg = (int)(group / 32);
    CT_REQUIRE(cdiv64(lanes, kBlock) < ((int64_t)1 << 31), "tensor too large for one launch");
    dim3 grid((unsigned)cdiv64(lanes, kBlock));
#define CT_RTN4(DT, SY) hipLaunchKernelGGL((rtn_w4_kernel<DT, SY>), grid, dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), lanes, lpg, \
                                           reinterpret_cast<u32x4*>(packed), scale_out, zp_out)
    if (xdt == CT_BF16) { if (symmetric) CT_RTN4(CT_BF16, true); else CT_RTN4(CT_BF16, false); }
    else { if (symmetric) CT_RTN4(CT_F16, true); else CT_RTN4(CT_F16, false); }
#undef CT_RTN4
    CT_LAUNCH_CHECK("ct_rtn_quant_pack_w4");
}

int ct_rtn_quant_channel8(const void* x, int xdt, int64_t rows, int64_t cols, int fp8, int symmetric, void* out, void* scale_out, int8_t* zp_out,
                          ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "the one-pass channel-wise compress takes 16-bit float weights, got dtype %d", xdt);
    CT_REQUIRE(rows >= 0 && cols >= 0 && cols % 8 == 0 && cols <= 8 * 8 * kBlock, "columns (%lld) must be a multiple of 8 and at most %d", (long long)cols,
               8 * 8 * kBlock);
    CT_REQUIRE(!fp8 || symmetric, "FLOAT 8-bit round-to-nearest is symmetric");
    CT_REQUIRE(aligned16(x) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && scale_out != nullptr, "misaligned buffers");
    CT_REQUIRE(symmetric || zp_out != nullptr, "asymmetric quantization needs the zero-point output");
    if (rows == 0 || cols == 0) return CT_OK;
    const int upr = (int)(cols / 8);
    // measured at 8192^2 (10 rotating sets): workgroup per row 37.6 (fp8) / 40.5 (int8) / 53.4 us (int8 asymmetric: three reductions
    // through LDS); wave per row 46.2 / 46.2 / 49.9 us.  So: wave per row for asymmetric schemes only.
    // Short rows (round 6, HBM-cold, profiles/r06_rtn8_wave_or_workgroup.txt): a workgroup per row of 1536 / 2048 columns keeps 3-4 KB in flight per
    // workgroup between two barriers — 24576 x 1536: 33.7 (int8) / 44.1 us (fp8) against 23.2 / 25.7 with a wave per row, 16384 x 2048: 24.6 / 34.1 against
    // 21.2 / 23.9; from 3584 columns on the workgroup form wins (8192 x 3584: 19.2 / 21.4 against 34.5 / 22.5).  So: wave per row for rows of at most 2048
    // columns, and for asymmetric schemes up to 8192.
    // (wave per row beyond 2048 columns stays behind the workgroup form with the 16-unit instantiation too: 14336 x 4096 36.4 against 32.8 us.  The 8-unit
    // instantiation is not used: hipcc gives it 235-252 VGPRs — 91 for 16 units — which made rows of 2049-4096 columns 34.5 / 65 us instead of 19.9 / 36.4.)
    const bool want_wave = !symmetric || upr <= 256;
    if (want_wave && upr <= 64 * 16 && cdiv64(rows, kBlock / 64) < ((int64_t)1 << 31)) {  // rows of up to 8192 elements: one wave per row
        const int needw = (upr + 63) / 64;
        const unsigned gw = (unsigned)cdiv64(rows, kBlock / 64);
#define CT_RW8(DT, MU, F8) hipLaunchKernelGGL((rtn_channel8_wave_kernel<DT, MU, F8>), dim3(gw), dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), rows, upr, \
                                              symmetric, static_cast<u32x2*>(out), scale_out, zp_out)
#define CT_RW8_U(DT, F8) do { if (needw <= 2) CT_RW8(DT, 2, F8); else if (needw <= 4) CT_RW8(DT, 4, F8); else CT_RW8(DT, 16, F8); } while (0)
        if (xdt == CT_BF16) { if (fp8) CT_RW8_U(CT_BF16, true); else CT_RW8_U(CT_BF16, false); }
        else { if (fp8) CT_RW8_U(CT_F16, true); else CT_RW8_U(CT_F16, false); }
#undef CT_RW8_U
#undef CT_RW8
        CT_LAUNCH_CHECK("ct_rtn_quant_channel8[wave]");
    }
    const int need = (upr + kBlock - 1) / kBlock;
    const unsigned grid = (unsigned)(rows < ((int64_t)1 << 30) ? rows : ((int64_t)1 << 30));
#define CT_RC8(DT, MU, F8) hipLaunchKernelGGL((rtn_channel8_kernel<DT, MU, F8>), dim3(grid), dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), rows, upr, \
                                              symmetric, static_cast<u32x2*>(out), scale_out, zp_out)
#define CT_RC8_U(DT, F8) do { if (need <= 1) CT_RC8(DT, 1, F8); else if (need <= 2) CT_RC8(DT, 2, F8); else if (need <= 4) CT_RC8(DT, 4, F8); else CT_RC8(DT, 8, F8); } while (0)
    if (xdt == CT_BF16) { if (fp8) CT_RC8_U(CT_BF16, true); else CT_RC8_U(CT_BF16, false); }
    else { if (fp8) CT_RC8_U(CT_F16, true); else CT_RC8_U(CT_F16, false); }
#undef CT_RC8_U
#undef CT_RC8
    CT_LAUNCH_CHECK("ct_rtn_quant_channel8");
}

int ct_unpack_dequant(const int32_t* packed, int64_t rows, int64_t words, int64_t cols, int bits,
                      const void* scale, int sdt, const void* zp, int zdt, int64_t rdiv, int64_t cdiv,
                      int64_t scale_cols, const int32_t* col_group, void* out, int odt, ct_stream_t stream) {
    CT_REQUIRE(bits >= 1 && bits <= 8, "Unpacking is only supported for num_bits in [1, 8], got %d", bits);
    CT_REQUIRE(is_float_dt(odt), "unsupported output dtype %d", odt);
    QParams p;
    int rc = fill_qparams(p, packed, CT_I32, scale, sdt, zp, zdt, rows, cols, rdiv, cdiv, scale_cols, col_group, bits, out, odt);
    if (rc) return rc;
    if (rows == 0 || cols == 0) return CT_OK;
    if (words == cols / 8 && w4_eligible(sdt, sdt, odt, bits, rows, cols, rdiv, cdiv, col_group, packed, out) &&
        (reinterpret_cast<uintptr_t>(packed) & 3u) == 0) {
        W4Params w = make_w4(packed, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        // units per lane: decomp_unroll_grouped()
        const int unroll = decomp_unroll_grouped(w.units);
        dim3 grid(w4_grid(w.units, unroll));
        // (a scales-first lean variant of this kernel measured SLOWER: 39-42 us vs 30 us)
        // one scale / zero-point load per 16-lane row when a row never straddles a scale group
        // (asymmetric only: on the symmetric kernel the same trick measured 29.5 -> 31.1 us)
        const bool rowlead = zp && w.flat_scale && w.upg_shift >= 4 && w.upg_shift < 62 && w.units % 16 == 0 && zdt == CT_I8;
        // round 5: groups of 128 on a tensor of many residency rounds fetch the wave's four scales / zero points by SCALAR loads — asymmetric
        // 29.85 -> 28.55 us at 8192^2 (71.0 -> 74.1 % of 8 TB/s), symmetric 28.6-29.2 -> 28.4; a tensor of ONE round is latency-bound and loses
        // with them (4096^2: 8.5 -> 10.2 us), so those keep the vector loads (tools/kbench/kbench_w4d.hip, profiles/r05_w4d_scale_modes.txt)
        const bool scalar = w.flat_scale && w.upg_shift == 4 && w.units % 64 == 0 && w.units > 8 * (int64_t)kCUs * 8 * kBlock && (!zp || zdt == CT_I8) &&
                            (reinterpret_cast<uintptr_t>(scale) & 7u) == 0 && (reinterpret_cast<uintptr_t>(zp) & 3u) == 0;
#define CT_W4D(DT, ZP) CT_FOR_UNROLL(unroll, if (scalar) hipLaunchKernelGGL((w4_unpack_dequant_kernel<DT, U, ZP, kW4ScaleScalar>), grid, dim3(kBlock), 0, as_stream(stream), w); \
                                             else if (rowlead) hipLaunchKernelGGL((w4_unpack_dequant_kernel<DT, U, ZP, kW4ScaleRowLead>), grid, dim3(kBlock), 0, as_stream(stream), w); \
                                             else hipLaunchKernelGGL((w4_unpack_dequant_kernel<DT, U, ZP, kW4ScalePerLane>), grid, dim3(kBlock), 0, as_stream(stream), w))
        if (sdt == CT_BF16) { if (zp) CT_W4D(CT_BF16, true); else CT_W4D(CT_BF16, false); }
        else { if (zp) CT_W4D(CT_F16, true); else CT_W4D(CT_F16, false); }
#undef CT_W4D
        CT_LAUNCH_CHECK("ct_unpack_dequant[w4]");
    }
    if (bits == 8 && words == cols / 4 && sdt == CT_F32 && odt == CT_F32 && f32_quads_ok(rows, cols, cdiv, col_group, out, packed)) {
        W4Params w = make_w4(packed, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        return launch_f32_quads<F32_DQ>(w, zp, sdt, 0.0f, 0.0f, stream, "ct_unpack_dequant[w8 f32]", 0x80808080u);
    }
    if (words == cols / 8 && w4_gidx_ok(sdt, sdt, odt, bits, rows, cols, rdiv, scale_cols, col_group, packed, out)) {
        W4Params w = make_w4(packed, scale, zp, zdt, out, rows, cols, rdiv, cols, scale_cols);
        return launch_w4_gidx<false>(w, sdt, zp, col_group, rows, stream, "ct_unpack_dequant[w4 g_idx]");
    }
    if (bits == 4 && words == cols / 8 && sdt == CT_F32 && odt == CT_F32 && !col_group && cols % 8 == 0 && cdiv % 8 == 0 && aligned16(out) &&
        (reinterpret_cast<uintptr_t>(packed) & 3u) == 0 && rows * (cols / 8) < ((int64_t)1 << 38)) {
        W4Params w = make_w4(packed, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        dim3 gf(w4_grid(2 * w.units, 4));
        if (zp) hipLaunchKernelGGL((w4_unpack_dequant_f32_kernel<true>), gf, dim3(kBlock), 0, as_stream(stream), w);
        else hipLaunchKernelGGL((w4_unpack_dequant_f32_kernel<false>), gf, dim3(kBlock), 0, as_stream(stream), w);
        CT_LAUNCH_CHECK("ct_unpack_dequant[w4 f32]");
    }
    if (bits == 8 && words == cols / 4 && cols % 32 == 0 && !col_group && (sdt == CT_BF16 || sdt == CT_F16) && odt == sdt && rows > 0 &&
        (cdiv % 8 == 0 || cdiv >= cols) && aligned16(out) && (reinterpret_cast<uintptr_t>(packed) & 7u) == 0) {
        W4Params w = make_w4(packed, scale, zp, zdt, out, rows, cols, rdiv, cdiv, scale_cols);
        // units per lane: decomp_unroll_grouped() (at 8 units per lane this variant — it un-biases the packed bytes — needs 117 VGPRs)
        const int unroll = decomp_unroll_grouped(w.units);
        dim3 g8(w4_grid(w.units, unroll));
#define CT_Q8U(DT, ZP) CT_FOR_UNROLL(unroll, hipLaunchKernelGGL((q8_dequant_kernel<DT, U, ZP, 128>), g8, dim3(kBlock), 0, as_stream(stream), w))
        if (sdt == CT_BF16) { if (zp) CT_Q8U(CT_BF16, true); else CT_Q8U(CT_BF16, false); }
        else { if (zp) CT_Q8U(CT_F16, true); else CT_Q8U(CT_F16, false); }
#undef CT_Q8U
        CT_LAUNCH_CHECK("ct_unpack_dequant[w8]");
    }
    if (words == (cols / 32) * bits && wb_layout_ok(sdt, sdt, odt, bits, zdt, zp, rows, cols, rdiv, cdiv, scale_cols, col_group, out, packed))
        return launch_wb_unpack_dequant(packed, scale, sdt, zp, rows, cols, cdiv, bits, out, stream);
    p.vec = (cols % 8 == 0) && aligned16(out);
    dim3 grid = grid_2d(rows, cdiv64(cols, 32));
    return bits <= 4 ? launch_unpack_dequant_g32_lo(p, sdt, bits, words, grid, stream) : launch_unpack_dequant_g32_hi(p, sdt, bits, words, grid, stream);
}

// fills the derived fields of one item; returns its block count or -1 (error set)
static int64_t w4_plan_item(ct_w4_item& it, int i, int direction, int64_t first_block) {
    const int64_t g = (it.group <= 0 || it.group > it.cols) ? it.cols : it.group;
    const bool ok = it.rows > 0 && it.cols > 0 && it.cols % 32 == 0 && g % 32 == 0 && it.cols % g == 0 && it.src && it.scale && it.dst &&
                    aligned16(it.src) && aligned16(it.dst);
    if (!ok) {
        set_error("ct_w4_batch_plan: item %d (rows %lld, cols %lld, group %lld) is not eligible for the batched W4 path "
                  "(needs cols %% 32 == 0, group %% 32 == 0, cols %% group == 0, 16-byte aligned buffers)", i, (long long)it.rows,
                  (long long)it.cols, (long long)it.group);
        return -1;
    }
    it.units = it.rows * (it.cols / 8);
    it.upg = (int32_t)(g / 8);
    it.upg_shift = log2_exact(it.upg);
    it.first_block = first_block;
    it.main_blocks = direction == 0 ? cdiv64(it.units / 4, kBlock) : cdiv64(it.units, (int64_t)kBlock * kBatchUnroll * kW4BatchIter);
    // n / G as (n * magic) >> shift, exact for n < 2^31 (Granlund-Montgomery with N = 31: magic = ceil(2^(31 + L) / G), L = ceil(log2 G), magic < 2^32)
    const int64_t G = it.cols / g;
    int L = 0;
    while (((int64_t)1 << L) < G) ++L;
    if (L > 31) {
        set_error("ct_w4_batch_plan: item %d has %lld groups per row", i, (long long)G);
        return -1;
    }
    it.g_shift = 31 + L;
    it.g_magic = (uint32_t)((((uint64_t)1 << it.g_shift) + (uint64_t)(G - 1)) / (uint64_t)G);  // 2^(31 + L) + G - 1 < 2^63
    int64_t tail = 0;
    if (it.zp_packed) {
        const bool fits = direction == 0
            ? it.zp != nullptr
            : (g == 128 && it.cols % 512 == 0 && it.units < ((int64_t)1 << 31) && aligned16(it.zp_packed) && (reinterpret_cast<uintptr_t>(it.scale) & 7u) == 0);
        if (!fits) {
            set_error(direction == 0 ? "ct_w4_batch_plan: item %d asks for packed zero points (zp_packed) without giving the int8 zero points (zp)"
                                     : "ct_w4_batch_plan: item %d (rows %lld, cols %lld, group %lld) cannot read its zero points in packed form "
                                       "(needs group == 128, cols %% 512 == 0, rows * cols < 2^34, zp_packed 16-byte and scale 8-byte aligned): "
                                       "unpack them first (ct_zp4_pack_dim0_batch) and pass zp",
                      i, (long long)it.rows, (long long)it.cols, (long long)it.group);
            return -1;
        }
        if (direction == 0 || it.zp != nullptr) tail = cdiv64(((it.rows + 7) / 8) * G, kBlock);
    }
    return it.main_blocks + tail;
}

int64_t ct_w4_batch_plan(ct_w4_item* items, int n, int direction) {
    if (n < 0 || (n > 0 && items == nullptr) || (direction != 0 && direction != 1)) {
        set_error("ct_w4_batch_plan: bad arguments");
        return -1;
    }
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t b = w4_plan_item(items[i], i, direction, blocks);
        if (b < 0) return -1;
        blocks += b;
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_w4_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    return blocks;
}

int ct_quant_pack_w4_zp(const void* x, int xdt, const void* scale, const int8_t* zp, int64_t rows, int64_t cols, int64_t group, int32_t* packed,
                        int32_t* zp_packed, ct_stream_t stream) {
    CT_REQUIRE(xdt == CT_BF16 || xdt == CT_F16, "ct_quant_pack_w4_zp: 16-bit weights only, got dtype %d", xdt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    CT_REQUIRE(zp != nullptr && zp_packed != nullptr && (reinterpret_cast<uintptr_t>(zp_packed) & 3u) == 0, "zp / zp_packed NULL or misaligned");
    if (rows == 0 || cols == 0) return CT_OK;
    ct_w4_item it{};
    it.src = x; it.scale = scale; it.zp = zp; it.dst = packed; it.rows = rows; it.cols = cols; it.group = group; it.zp_packed = zp_packed;
    const int64_t blocks = w4_plan_item(it, 0, 0, 0);
    if (blocks < 0) return CT_ERR_INVALID_ARG;
    CT_REQUIRE(blocks < ((int64_t)1 << 31), "tensor too large for one launch");
    if (xdt == CT_BF16) hipLaunchKernelGGL((w4_quant_pack_one_kernel<CT_BF16>), dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), it);
    else hipLaunchKernelGGL((w4_quant_pack_one_kernel<CT_F16>), dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), it);
    CT_LAUNCH_CHECK("ct_quant_pack_w4_zp");
}

int ct_unpack_dequant_w4_zp(const int32_t* packed, const void* scale, int sdt, const int32_t* zp_packed, int64_t rows, int64_t cols, int64_t group,
                            void* out, int8_t* zp_out, ct_stream_t stream) {
    CT_REQUIRE(sdt == CT_BF16 || sdt == CT_F16, "ct_unpack_dequant_w4_zp: 16-bit scales / results only, got dtype %d", sdt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    CT_REQUIRE(zp_packed != nullptr, "zp_packed is NULL");
    if (rows == 0 || cols == 0) return CT_OK;
    if (!(group == 128 && cols % 512 == 0 && rows * (cols / 8) < ((int64_t)1 << 31) && aligned16(zp_packed) && (reinterpret_cast<uintptr_t>(scale) & 7u) == 0))
        CT_UNSUPPORTED("ct_unpack_dequant_w4_zp: needs group == 128, cols %% 512 == 0, rows * cols < 2^34, zp_packed 16-byte and scale 8-byte aligned "
                       "(unpack the zero points with ct_unpack_int32_dim0 and call ct_unpack_dequant)");
    ct_w4_item it{};
    it.src = packed; it.scale = scale; it.zp = zp_out; it.dst = out; it.rows = rows; it.cols = cols; it.group = group;
    it.zp_packed = const_cast<int32_t*>(zp_packed);
    const int64_t blocks = w4_plan_item(it, 0, 1, 0);
    if (blocks < 0) return CT_ERR_INVALID_ARG;
    CT_REQUIRE(blocks < ((int64_t)1 << 31), "tensor too large for one launch");
    const int64_t stride = (int64_t)kBlock * kBatchUnroll;
    if (sdt == CT_BF16) hipLaunchKernelGGL((w4_unpack_dequant_one_kernel<CT_BF16>), dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), it, stride);
    else hipLaunchKernelGGL((w4_unpack_dequant_one_kernel<CT_F16>), dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), it, stride);
    CT_LAUNCH_CHECK("ct_unpack_dequant_w4_zp");
}

int ct_quant_pack_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_BF16 || dt == CT_F16, "batched W4 path: 16-bit weights only, got dtype %d", dt);
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    if (dt == CT_BF16) hipLaunchKernelGGL((w4_quant_pack_batch_kernel<CT_BF16>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n);
    else hipLaunchKernelGGL((w4_quant_pack_batch_kernel<CT_F16>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n);
    CT_LAUNCH_CHECK("ct_quant_pack_batch");
}

int ct_unpack_dequant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_BF16 || dt == CT_F16, "batched W4 path: 16-bit weights only, got dtype %d", dt);
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    const int64_t stride = (int64_t)kBlock * kBatchUnroll;
    if (dt == CT_BF16) hipLaunchKernelGGL((w4_unpack_dequant_batch_kernel<CT_BF16>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n, stride);
    else hipLaunchKernelGGL((w4_unpack_dequant_batch_kernel<CT_F16>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n, stride);
    CT_LAUNCH_CHECK("ct_unpack_dequant_batch");
}

int64_t ct_q8_batch_plan(ct_w4_item* items, int n, int direction) {
    if (n < 0 || (n > 0 && items == nullptr) || (direction != 0 && direction != 1)) {
        set_error("ct_q8_batch_plan: bad arguments");
        return -1;
    }
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        ct_w4_item& it = items[i];
        const int64_t numel = it.rows * it.cols;
        it.main_blocks = 0;
        if (it.group < 0) {  // block strategy: -group = (rows per block << 24) | columns per block
            const int64_t bh = (-it.group) >> 24, bw = (-it.group) & 0xffffff;
            const bool okb = it.rows > 0 && it.cols > 0 && bh >= 1 && bw >= 16 && (bh & (bh - 1)) == 0 && (bw & (bw - 1)) == 0 && it.cols % bw == 0 && it.src && it.scale && it.dst &&
                             aligned16(it.src) && aligned16(it.dst) && numel / 8 < ((int64_t)1 << 31);
            if (!okb) {
                set_error("ct_q8_batch_plan: item %d (rows %lld, cols %lld, block %lld x %lld) is not eligible for the batched 8-bit path (block sides must be powers of "
                          "two, the width >= 16 and a divisor of cols, fewer than 2^34 elements, 16-byte aligned buffers)", i, (long long)it.rows, (long long)it.cols,
                          (long long)bh, (long long)bw);
                return -1;
            }
            it.units = numel / 8;
            it.upg = (int32_t)(bw / 8);
            it.upg_shift = log2_exact(it.upg);
            it.main_blocks = log2_exact(bh) + 1;
            const int64_t upr = it.cols / 8;  // n / upr as (n * magic) >> shift, exact for n < 2^31 (Granlund-Montgomery, N = 31)
            int L = 0;
            while (((int64_t)1 << L) < upr) ++L;
            it.g_shift = 31 + L;
            it.g_magic = (uint32_t)((((uint64_t)1 << it.g_shift) + (uint64_t)(upr - 1)) / (uint64_t)upr);
            it.first_block = blocks;
            blocks += direction == 0 ? cdiv64(it.units / 2, kBlock) : cdiv64(it.units, (int64_t)kBlock * kBatchUnroll * kBatchIter);
            continue;
        }
        const bool whole = it.rows > 0 && it.cols > 0 && it.group >= numel;  // per tensor
        const int64_t g = whole ? numel : ((it.group <= 0 || it.group > it.cols) ? it.cols : it.group);
        const bool ok = it.rows > 0 && it.cols > 0 && it.cols % 16 == 0 && g % 16 == 0 && (whole || it.cols % g == 0) && it.src && it.scale && it.dst &&
                        aligned16(it.src) && aligned16(it.dst);
        if (!ok) {
            set_error("ct_q8_batch_plan: item %d (rows %lld, cols %lld, group %lld) is not eligible for the batched 8-bit path "
                      "(needs cols %% 16 == 0, group %% 16 == 0, cols %% group == 0 or group == rows * cols, 16-byte aligned buffers)", i,
                      (long long)it.rows, (long long)it.cols, (long long)it.group);
            return -1;
        }
        it.units = numel / 8;
        if (whole || g / 8 > 0x7fffffff) {  // one scale for everything: index 0 without a division (w4_scale_index: u >> 62)
            it.upg = 0;
            it.upg_shift = 62;
            if (!whole) { set_error("ct_q8_batch_plan: group too large"); return -1; }
        } else {
            it.upg = (int32_t)(g / 8);
            it.upg_shift = log2_exact(it.upg);
        }
        it.first_block = blocks;
        blocks += direction == 0 ? cdiv64(it.units / 2, kBlock) : cdiv64(it.units, (int64_t)kBlock * kBatchUnroll * kBatchIter);
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_q8_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    return blocks;
}

int ct_q8_quant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, int fp8, int bits, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_BF16 || dt == CT_F16, "batched 8-bit path: 16-bit weights only, got dtype %d", dt);
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    CT_REQUIRE((fp8 == 1 || fp8 == 2) || (bits >= 1 && bits <= 8), "num_bits must be in [1, 8], got %d", bits);
    if (n == 0 || total_blocks == 0) return CT_OK;
    const bool f8codes = fp8 == 1 || fp8 == 2;
    const int qmin = f8codes ? 0 : -(1 << (bits - 1)), qmax = f8codes ? 0 : (1 << (bits - 1)) - 1;
    CT_REQUIRE(fp8 >= 0 && fp8 <= 3, "fp8 must be 0 (int8 codes), 1 (float8 codes, int8 zero points), 2 (float8 codes, float8 zero points) or 3 (int8 codes + 128 in int32 words), got %d", fp8);
    const int zdt = fp8 == 2 ? CT_F8E4M3 : CT_I8;
    const dim3 grid((unsigned)total_blocks);
    if (fp8 == 3) {  // pack-quantized 8-bit (pack_to_int32 of 8-bit codes, helpers.py:39-75): dst = int32 (rows, cols / 4)
        CT_REQUIRE(bits == 8, "packed 8-bit words need num_bits == 8, got %d", bits);
        if (dt == CT_BF16) hipLaunchKernelGGL((q8_quant_batch_kernel<CT_BF16, false, 128>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, qmin, qmax, zdt);
        else hipLaunchKernelGGL((q8_quant_batch_kernel<CT_F16, false, 128>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, qmin, qmax, zdt);
        CT_LAUNCH_CHECK("ct_q8_quant_batch[packed]");
    }
    if (dt == CT_BF16) {
        if (fp8) hipLaunchKernelGGL((q8_quant_batch_kernel<CT_BF16, true>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, qmin, qmax, zdt);
        else hipLaunchKernelGGL((q8_quant_batch_kernel<CT_BF16, false>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, qmin, qmax, zdt);
    } else {
        if (fp8) hipLaunchKernelGGL((q8_quant_batch_kernel<CT_F16, true>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, qmin, qmax, zdt);
        else hipLaunchKernelGGL((q8_quant_batch_kernel<CT_F16, false>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, qmin, qmax, zdt);
    }
    CT_LAUNCH_CHECK("ct_q8_quant_batch");
}

int ct_q8_dequant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, int fp8, ct_stream_t stream) {
    CT_REQUIRE(dt == CT_BF16 || dt == CT_F16, "batched 8-bit path: 16-bit weights only, got dtype %d", dt);
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    const int64_t stride = (int64_t)kBlock * kBatchUnroll;
    CT_REQUIRE(fp8 >= 0 && fp8 <= 3, "fp8 must be 0, 1, 2 (float8 codes with float8 zero points) or 3 (int8 codes + 128 in int32 words), got %d", fp8);
    const int zdt = fp8 == 2 ? CT_F8E4M3 : CT_I8;
    const dim3 grid((unsigned)total_blocks);
    if (fp8 == 3) {
        if (dt == CT_BF16) hipLaunchKernelGGL((q8_dequant_batch_kernel<CT_BF16, false, 128>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, stride, zdt);
        else hipLaunchKernelGGL((q8_dequant_batch_kernel<CT_F16, false, 128>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, stride, zdt);
        CT_LAUNCH_CHECK("ct_q8_dequant_batch[packed]");
    }
    if (dt == CT_BF16) {
        if (fp8) hipLaunchKernelGGL((q8_dequant_batch_kernel<CT_BF16, true>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, stride, zdt);
        else hipLaunchKernelGGL((q8_dequant_batch_kernel<CT_BF16, false>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, stride, zdt);
    } else {
        if (fp8) hipLaunchKernelGGL((q8_dequant_batch_kernel<CT_F16, true>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, stride, zdt);
        else hipLaunchKernelGGL((q8_dequant_batch_kernel<CT_F16, false>), grid, dim3(kBlock), 0, as_stream(stream), items_dev, n, stride, zdt);
    }
    CT_LAUNCH_CHECK("ct_q8_dequant_batch");
}

int ct_selftest_bf16_div(uint32_t s_lo_bits, uint32_t s_hi_bits, unsigned long long* mismatches, ct_stream_t stream) {
    CT_REQUIRE(s_lo_bits <= s_hi_bits && s_hi_bits <= 65536u, "bad scale bit range");
    hipError_t e = hipMemsetAsync(mismatches, 0, sizeof(unsigned long long), as_stream(stream));
    if (e != hipSuccess) return hip_check(e, "ct_selftest_bf16_div memset");
    if (s_lo_bits == s_hi_bits) return CT_OK;
    unsigned n = s_hi_bits - s_lo_bits;
    hipLaunchKernelGGL(selftest_bf16_div_kernel, dim3(n < 4096 ? n : 4096), dim3(kBlock), 0, as_stream(stream), s_lo_bits, s_hi_bits, mismatches);
    CT_LAUNCH_CHECK("ct_selftest_bf16_div");
}

}  // extern "C"
