// ct_pack.hip — unfused pack_to_int32 / unpack_from_int32 (reference
// compressors/pack_quantized/helpers.py:20-180) for callers that hold int8 codes, and the
// row-wise (packed_dim=0) variants used for zero points (compressors/pack_quantized/base.py:107-110,
// 147-153).
//
// Layout: element i of a row occupies bits [i*B, i*B+B) of the row's little-endian bitstream;
// 32 elements <-> B int32 words.  One 32-element group per lane: all bit positions are
// compile-time constants after unrolling, every word is written exactly once, no atomics.
#include "ct_common.h"

namespace ct {

// The reference accumulates with scatter_add_ on int32 and never masks the biased code, so
// out-of-range int8 input yields the wrapping SUM of (u << off) plus the arithmetic-shifted
// spill; in-range codes make the adds ORs of disjoint bits.  Adds are used here too, so the
// result is bit-identical for any int8 input.
template <int BITS>
__device__ __forceinline__ void pack_insert(uint32_t (&words)[BITS + 1], int idx /*0..31, constant*/, int q) {
    const int u = q + (1 << (BITS - 1));
    const int pos = idx * BITS;
    const int w = pos >> 5, sh = pos & 31;
    words[w] += (uint32_t)u << sh;
    if (sh + BITS > 32) words[w + 1] += (uint32_t)(u >> (32 - sh));
}

template <int BITS>
__device__ __forceinline__ int unpack_extract(const uint32_t (&words)[BITS + 1], int idx) {
    const int pos = idx * BITS;
    const int w = pos >> 5, sh = pos & 31;
    uint32_t code = words[w] >> sh;
    if (sh + BITS > 32) code |= words[w + 1] << (32 - sh);
    return (int)(code & ((1u << BITS) - 1u)) - (1 << (BITS - 1));
}

template <int BITS>
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(const int8_t* __restrict__ q, int64_t rows, int64_t cols,
                                                           int32_t* __restrict__ out, int64_t out_stride,
                                                           int64_t packed_cols, int vec) {
    const int64_t gpr = (cols + 31) >> 5;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int8_t* qr = q + row * cols;
        for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < gpr; g += (int64_t)gridDim.x * kBlock) {
            uint32_t words[BITS + 1];
#pragma unroll
            for (int j = 0; j <= BITS; ++j) words[j] = 0;
            const int64_t c0 = g << 5;
            if (vec && c0 + 32 <= cols) {
                const u32x4* p = reinterpret_cast<const u32x4*>(qr + c0);
                u32x4 a = p[0], b = p[1];
                const uint32_t ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 32; ++i) pack_insert<BITS>(words, i, (int)(int8_t)(ws[i >> 2] >> (8 * (i & 3))));
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    if (c0 + i < cols) pack_insert<BITS>(words, i, (int)qr[c0 + i]);
            }
            store_words<BITS>(out + row * out_stride + g * BITS, words, packed_cols - g * BITS);
        }
    }
}

template <int BITS>
__global__ __launch_bounds__(kBlock) void unpack_rows_kernel(const int32_t* __restrict__ p, int64_t rows, int64_t words_per_row,
                                                             int64_t p_stride, int64_t cols, int8_t* __restrict__ out,
                                                             int vec) {
    const int64_t gpr = (cols + 31) >> 5;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int32_t* pr = p + row * p_stride;
        int8_t* orow = out + row * cols;
        for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < gpr; g += (int64_t)gridDim.x * kBlock) {
            uint32_t words[BITS + 1];
            load_words<BITS>(pr + g * BITS, words, words_per_row - g * BITS);
            words[BITS] = 0;
            const int64_t c0 = g << 5;
            if (vec && c0 + 32 <= cols) {
                uint32_t ws[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) ws[j] = 0;
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    ws[i >> 2] |= ((uint32_t)unpack_extract<BITS>(words, i) & 0xffu) << (8 * (i & 3));
                u32x4* o = reinterpret_cast<u32x4*>(orow + c0);
                stream_store16(o, u32x4{ws[0], ws[1], ws[2], ws[3]});
                stream_store16(o + 1, u32x4{ws[4], ws[5], ws[6], ws[7]});
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i)
                    if (c0 + i < cols) orow[c0 + i] = (int8_t)unpack_extract<BITS>(words, i);
            }
        }
    }
}

// flat forms of the two common widths (contiguous rows, cols % 32 == 0: the packed tensor is one stream of words).
// The lane-per-group kernel above stores 32 bytes per lane (two lane-strided 16-byte stores: 21.9 us for 8192^2 int4);
// here a lane owns HALF a group resp. one word pair -> exactly one 16-byte store, 1 KiB contiguous per wave instruction.
//   4 bits: 2 words (8 B) in -> 16 codes: nibbles split with two masks, interleaved with v_perm_b32, un-biased per byte
//           as (u + 0x78) ^ 0x80 (no carry between bytes: u <= 15).
//   8 bits: 4 words (16 B) in -> 16 codes: u ^ 0x80.
template <int BITS, int UNROLL>
__global__ __launch_bounds__(kBlock) void unpack_flat_kernel(const uint32_t* __restrict__ p, int64_t items, u32x4* __restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x;
    if constexpr (BITS == 4) {
        u32x2 w[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t it = base + (int64_t)i * kBlock;
            if (it < items) w[i] = reinterpret_cast<const u32x2*>(p)[it];
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t it = base + (int64_t)i * kBlock;
            if (it >= items) continue;
            uint32_t o[4];
            const uint32_t ws[2] = {w[i].x, w[i].y};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t lo = ws[h] & 0x0f0f0f0fu, hi = (ws[h] >> 4) & 0x0f0f0f0fu;
                const uint32_t a = __builtin_amdgcn_perm(hi, lo, 0x05010400u), b = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
                o[2 * h] = (a + 0x78787878u) ^ 0x80808080u;
                o[2 * h + 1] = (b + 0x78787878u) ^ 0x80808080u;
            }
            stream_store16(out + it, u32x4{o[0], o[1], o[2], o[3]});
        }
    } else {
        u32x4 w[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t it = base + (int64_t)i * kBlock;
            if (it < items) w[i] = reinterpret_cast<const u32x4*>(p)[it];
        }
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            const int64_t it = base + (int64_t)i * kBlock;
            if (it < items) stream_store16(out + it, u32x4{w[i].x ^ 0x80808080u, w[i].y ^ 0x80808080u, w[i].z ^ 0x80808080u, w[i].w ^ 0x80808080u});
        }
    }
}

// flat 4-bit pack (contiguous rows, cols % 32 == 0): a lane owns HALF a pack group — 16 codes = one 16-byte load (1 KiB contiguous
// per wave instruction; the lane-per-group kernel reads 32 B per lane as two lane-strided 16-byte loads, the shape that measured
// 13 points lower in DESIGN.md 5.1) -> two words = one 8-byte streaming store.  Same wrapping adds as pack_insert, so
// out-of-range int8 input stays bit-identical with the reference's scatter_add_.
template <int UNROLL>
__global__ __launch_bounds__(kBlock) void pack_flat4_kernel(const u32x4* __restrict__ q, int64_t items, u32x2* __restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * kBlock * UNROLL + threadIdx.x;
    u32x4 r[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t it = base + (int64_t)i * kBlock;
        if (it < items) r[i] = q[it];
    }
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        const int64_t it = base + (int64_t)i * kBlock;
        if (it >= items) continue;
        const uint32_t ws[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
        uint32_t w[2] = {0x88888888u, 0x88888888u};  // the +8 of every code
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k >> 3] += (uint32_t)(int)(int8_t)(ws[k >> 2] >> (8 * (k & 3))) << (4 * (k & 7));
        stream_store8(out + it, u32x2{w[0], w[1]});
    }
}

// packed along rows: lane (g, c) packs rows [32g, 32g+32) of column c; lanes are consecutive
// in c so every access is coalesced across the wave.
template <int BITS>
__global__ __launch_bounds__(kBlock) void pack_dim0_kernel(const int8_t* __restrict__ q, int64_t rows, int64_t cols,
                                                           int32_t* __restrict__ out, int64_t packed_rows) {
    const int64_t groups = (rows + 31) >> 5;
    for (int64_t g = blockIdx.y; g < groups; g += gridDim.y)
        for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < cols; c += (int64_t)gridDim.x * kBlock) {
            uint32_t words[BITS + 1];
#pragma unroll
            for (int j = 0; j <= BITS; ++j) words[j] = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int64_t r = (g << 5) + i;
                if (r < rows) pack_insert<BITS>(words, i, (int)q[r * cols + c]);
            }
#pragma unroll
            for (int j = 0; j < BITS; ++j)
                if (g * BITS + j < packed_rows) out[(g * BITS + j) * cols + c] = (int32_t)words[j];
        }
}

template <int BITS>
__global__ __launch_bounds__(kBlock) void unpack_dim0_kernel(const int32_t* __restrict__ p, int64_t words_rows,
                                                             int64_t cols, int64_t rows, int8_t* __restrict__ out) {
    const int64_t groups = (rows + 31) >> 5;
    for (int64_t g = blockIdx.y; g < groups; g += gridDim.y)
        for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < cols; c += (int64_t)gridDim.x * kBlock) {
            uint32_t words[BITS + 1];
#pragma unroll
            for (int j = 0; j < BITS; ++j)
                words[j] = (g * BITS + j < words_rows) ? (uint32_t)p[(g * BITS + j) * cols + c] : 0u;
            words[BITS] = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int64_t r = (g << 5) + i;
                if (r < rows) out[r * cols + c] = (int8_t)unpack_extract<BITS>(words, i);
            }
        }
}

// batched zero-point packing (4 bits, packed along rows): ONE launch for the zero points of a whole checkpoint — the
// `pack_to_int32(zp, bits, packed_dim=0)` / `unpack_from_int32(..., packed_dim=0)` calls of PackedQuantizationCompressor
// (compressors/pack_quantized/base.py:107-110,147-153) that the per-module loop issues once per asymmetric module.  Table type
// and workgroup search as in the W4 batch (ct_quant.hip); item = (src, dst, rows, cols) of the UNPACKED zero-point matrix.
__device__ __forceinline__ const ct_w4_item& zp_batch_find(const ct_w4_item* __restrict__ items, int n, int64_t block) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= block) lo = mid; else hi = mid - 1;
    }
    return items[lo];
}

template <bool PACK>
__global__ __launch_bounds__(kBlock) void zp4_dim0_batch_kernel(const ct_w4_item* __restrict__ items, int n) {
    const ct_w4_item& it = zp_batch_find(items, n, blockIdx.x);
    const int64_t rows = it.rows, cols = it.cols;
    const int64_t idx = ((int64_t)blockIdx.x - it.first_block) * kBlock + threadIdx.x;  // (row group, column): consecutive lanes = consecutive columns
    if (idx >= it.units) return;
    const int64_t g = idx / cols, c = idx - g * cols;
    const int64_t packed_rows = (rows * 4 + 31) >> 5;
    uint32_t words[5] = {0u, 0u, 0u, 0u, 0u};
    if (PACK) {
        const int8_t* q = static_cast<const int8_t*>(it.src);
        int32_t* out = static_cast<int32_t*>(it.dst);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int64_t r = (g << 5) + i;
            if (r < rows) pack_insert<4>(words, i, (int)q[r * cols + c]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (g * 4 + j < packed_rows) out[(g * 4 + j) * cols + c] = (int32_t)words[j];
    } else {
        const int32_t* p = static_cast<const int32_t*>(it.src);
        int8_t* out = static_cast<int8_t*>(it.dst);
#pragma unroll
        for (int j = 0; j < 4; ++j) words[j] = (g * 4 + j < packed_rows) ? (uint32_t)p[(g * 4 + j) * cols + c] : 0u;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int64_t r = (g << 5) + i;
            if (r < rows) out[r * cols + c] = (int8_t)unpack_extract<4>(words, i);
        }
    }
}

static dim3 grid_rows(int64_t rows, int64_t items_per_row) {
    int64_t gx = cdiv64(items_per_row, kBlock);
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    int64_t gy = rows < 1 ? 1 : rows;
    int64_t cap = (8 * kCUs * 4) / gx;
    if (cap < 1) cap = 1;
    if (gy > cap) gy = cap;
    if (gy > 65535) gy = 65535;
    return dim3((unsigned)gx, (unsigned)gy, 1);
}

#define CT_BITS_SWITCH(bits, ...)                            \
    switch (bits) {                                          \
        case 1: { constexpr int B = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int B = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int B = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int B = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int B = 5; __VA_ARGS__; } break; \
        case 6: { constexpr int B = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int B = 7; __VA_ARGS__; } break; \
        case 8: { constexpr int B = 8; __VA_ARGS__; } break; \
    }

// ------------------------------------------------------------------------------------------
// Activation ordering: the group of every column (forward_helpers.py:147-175 — the weight's columns are permuted by argsort(g_idx), groups of
// `group_size` consecutive SORTED columns share a scale, the result is permuted back): col_group[c] = rank(c) / group_size with rank = the position of c
// in the stable sort of g_idx, or c / group_size while g_idx still holds a -1.  The host composed this from two argsorts and five more tensor ops PER
// CALL (~60 us of launches, nothing cached: a g_idx can be rewritten in place) — more than the weight pass of most modules.  One small launch:
//   any -1 -> mode 0; every value g in [0, cols / group_size) exactly group_size times -> mode 1 (the usual case: the sorted positions of the columns of
//   value g are [g gs, (g + 1) gs), whatever the order of ties: col_group = g_idx); else mode 2: per column, the columns with a smaller value and the
//   EARLIER columns with an equal one are counted.
// ------------------------------------------------------------------------------------------
constexpr int kGidxBins = 4096;

// ONE launch: every workgroup classifies the whole g_idx by itself (at most 112 KB, from L2 after the first reader; four columns per lane and trip) and then
// writes the table entries of its own 1024 columns — the first version ran a one-workgroup classifier in front of the table kernel: two dependent launches,
// 6-7 us in front of every weight launch (an 8B-shaped activation-ordered tree: 4.98 ms, of which ~1.5 ms were these)
constexpr int kGidxBlock = 1024;
__global__ __launch_bounds__(kGidxBlock) void gidx_col_group_kernel(const int32_t* __restrict__ g_idx, int64_t cols, int64_t group_size, int32_t* __restrict__ mode_out,
                                                                    int32_t* __restrict__ out) {
    __shared__ int hist[kGidxBins];
    __shared__ int flags[2];  // any -1, any value outside [0, G) or a group that is not exactly group_size columns
    __shared__ int32_t tile[1024];
    const int64_t G = group_size > 0 ? cols / group_size : 0;
    const bool binned = group_size > 0 && cols % group_size == 0 && G <= kGidxBins;
    for (int i = threadIdx.x; i < kGidxBins; i += kGidxBlock) hist[i] = 0;
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    __syncthreads();
    auto take = [&](int32_t v) {
        if (v == -1) flags[0] = 1;
        if (binned && v >= 0 && v < G) atomicAdd(&hist[v], 1);
        else flags[1] = 1;
    };
    const bool vec = (reinterpret_cast<uintptr_t>(g_idx) & 15u) == 0;
    const int64_t quads = vec ? cols / 4 : 0;
    for (int64_t q = threadIdx.x; q < quads; q += kGidxBlock) {
        const u32x4 v = reinterpret_cast<const u32x4*>(g_idx)[q];
        take((int32_t)v.x); take((int32_t)v.y); take((int32_t)v.z); take((int32_t)v.w);
    }
    for (int64_t c = quads * 4 + threadIdx.x; c < cols; c += kGidxBlock) take(g_idx[c]);
    __syncthreads();
    if (binned && !flags[1]) {
        for (int64_t g = threadIdx.x; g < G; g += kGidxBlock)
            if (hist[g] != group_size) flags[1] = 1;
    }
    __syncthreads();
    const int m = flags[0] ? 0 : (flags[1] ? 2 : 1);  // the same in every workgroup
    if (blockIdx.x == 0 && threadIdx.x == 0) mode_out[0] = m;
    const int64_t c = (int64_t)blockIdx.x * kGidxBlock + threadIdx.x;
    if (m != 2) {
        if (c < cols) out[c] = m == 0 ? (int32_t)(c / group_size) : g_idx[c];
        return;
    }
    // the general case: the rank of column c in the stable sort of g_idx, by counting (tiles of g_idx in LDS, broadcast reads: O(cols) per column, rare)
    const int32_t v = c < cols ? g_idx[c] : 0;
    int64_t rank = 0;
    for (int64_t t0 = 0; t0 < cols; t0 += 1024) {
        __syncthreads();
        tile[threadIdx.x] = t0 + threadIdx.x < cols ? g_idx[t0 + threadIdx.x] : 0x7fffffff;
        __syncthreads();
        const int lim = (int)(cols - t0 < 1024 ? cols - t0 : 1024);
        for (int i = 0; i < lim; ++i) {
            const int32_t u = tile[i];
            rank += (u < v) || (u == v && t0 + i < c);
        }
    }
    if (c < cols) out[c] = (int32_t)(rank / group_size);
}

}  // namespace ct

using namespace ct;

extern "C" {

int ct_pack_int32(const int8_t* q, int64_t rows, int64_t cols, int bits, int32_t* out, int64_t out_row_stride,
                  ct_stream_t stream) {
    CT_REQUIRE(bits >= 1 && bits <= 8, "Packing is only supported for num_bits in [1, 8], got %d", bits);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    const int64_t packed_cols = cdiv64(cols * bits, 32);
    CT_REQUIRE(out_row_stride >= packed_cols, "output row stride %lld < %lld packed words", (long long)out_row_stride,
               (long long)packed_cols);
    if (rows == 0 || cols == 0) return CT_OK;
    if (bits == 8 && cols % 32 == 0 && out_row_stride == packed_cols && aligned16(q) && aligned16(out)) {
        // 8 bits: u = q + 128 fits its byte for every int8 input, so the word is the four bytes with their sign bits flipped —
        // the same 16 bytes in, 16 bytes out shape as unpack_flat_kernel<8> (26.3 us for the lane-per-group form at 8192^2)
        constexpr int U = 2;
        const int64_t items = rows * cols / 16;
        const int64_t g = cdiv64(items, (int64_t)kBlock * U);
        CT_REQUIRE(g < ((int64_t)1 << 31), "tensor too large for one launch");
        hipLaunchKernelGGL((unpack_flat_kernel<8, U>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const uint32_t*>(q), items,
                           reinterpret_cast<u32x4*>(out));
        CT_LAUNCH_CHECK("ct_pack_int32[flat8]");
    }
    if (bits == 4 && cols % 32 == 0 && out_row_stride == packed_cols && aligned16(q) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0) {
        constexpr int U = 2;
        const int64_t items = rows * cols / 16;
        const int64_t g = cdiv64(items, (int64_t)kBlock * U);
        CT_REQUIRE(g < ((int64_t)1 << 31), "tensor too large for one launch");
        hipLaunchKernelGGL((pack_flat4_kernel<U>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const u32x4*>(q), items,
                           reinterpret_cast<u32x2*>(out));
        CT_LAUNCH_CHECK("ct_pack_int32[flat4]");
    }
    const int vec = (cols % 16 == 0) && aligned16(q);
    dim3 grid = grid_rows(rows, cdiv64(cols, 32));
    CT_BITS_SWITCH(bits, hipLaunchKernelGGL((pack_rows_kernel<B>), grid, dim3(kBlock), 0, as_stream(stream), q, rows, cols, out,
                                            out_row_stride, packed_cols, vec));
    CT_LAUNCH_CHECK("ct_pack_int32");
}

int ct_unpack_int32(const int32_t* p, int64_t rows, int64_t words, int64_t p_row_stride, int64_t cols, int bits,
                    int8_t* out, ct_stream_t stream) {
    CT_REQUIRE(bits >= 1 && bits <= 8, "Unpacking is only supported for num_bits in [1, 8], got %d", bits);
    CT_REQUIRE(rows >= 0 && cols >= 0 && words >= 0, "negative shape");
    CT_REQUIRE(p_row_stride >= words, "packed row stride smaller than the row");
    if (rows == 0 || cols == 0) return CT_OK;
    if ((bits == 4 || bits == 8) && cols % 32 == 0 && p_row_stride == words && words == cols * bits / 32 && aligned16(out) && aligned16(p)) {
        constexpr int U = 2;
        const int64_t items = rows * cols / 16;  // 16 codes -> one 16-byte store
        const int64_t g = cdiv64(items, (int64_t)kBlock * U);
        CT_REQUIRE(g < ((int64_t)1 << 31), "tensor too large for one launch");
        if (bits == 4) hipLaunchKernelGGL((unpack_flat_kernel<4, U>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const uint32_t*>(p), items,
                                          reinterpret_cast<u32x4*>(out));
        else hipLaunchKernelGGL((unpack_flat_kernel<8, U>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const uint32_t*>(p), items,
                                reinterpret_cast<u32x4*>(out));
        CT_LAUNCH_CHECK("ct_unpack_int32[flat]");
    }
    const int vec = (cols % 16 == 0) && aligned16(out);
    dim3 grid = grid_rows(rows, cdiv64(cols, 32));
    CT_BITS_SWITCH(bits, hipLaunchKernelGGL((unpack_rows_kernel<B>), grid, dim3(kBlock), 0, as_stream(stream), p, rows, words,
                                            p_row_stride, cols, out, vec));
    CT_LAUNCH_CHECK("ct_unpack_int32");
}

int64_t ct_zp4_batch_plan(ct_w4_item* items, int n) {
    if (n < 0 || (n > 0 && items == nullptr)) {
        set_error("ct_zp4_batch_plan: bad arguments");
        return -1;
    }
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        ct_w4_item& it = items[i];
        if (!(it.rows > 0 && it.cols > 0 && it.src && it.dst)) {
            set_error("ct_zp4_batch_plan: item %d has an empty shape or a NULL pointer", i);
            return -1;
        }
        it.units = cdiv64(it.rows, 32) * it.cols;  // one lane per (32-row group, column)
        it.first_block = blocks;
        blocks += cdiv64(it.units, kBlock);
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_zp4_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    return blocks;
}

int ct_zp4_pack_dim0_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int direction, ct_stream_t stream) {
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31) && (direction == 0 || direction == 1), "bad batch arguments");
    if (n == 0 || total_blocks == 0) return CT_OK;
    if (direction == 0) hipLaunchKernelGGL(zp4_dim0_batch_kernel<true>, dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n);
    else hipLaunchKernelGGL(zp4_dim0_batch_kernel<false>, dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n);
    CT_LAUNCH_CHECK("ct_zp4_pack_dim0_batch");
}

int ct_pack_int32_dim0(const int8_t* q, int64_t rows, int64_t cols, int bits, int32_t* out, ct_stream_t stream) {
    CT_REQUIRE(bits >= 1 && bits <= 8, "Packing is only supported for num_bits in [1, 8], got %d", bits);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t packed_rows = cdiv64(rows * bits, 32);
    dim3 grid = grid_rows(cdiv64(rows, 32), cols);
    CT_BITS_SWITCH(bits, hipLaunchKernelGGL((pack_dim0_kernel<B>), grid, dim3(kBlock), 0, as_stream(stream), q, rows, cols, out,
                                            packed_rows));
    CT_LAUNCH_CHECK("ct_pack_int32_dim0");
}

int ct_unpack_int32_dim0(const int32_t* p, int64_t words, int64_t cols, int64_t rows, int bits, int8_t* out,
                         ct_stream_t stream) {
    CT_REQUIRE(bits >= 1 && bits <= 8, "Unpacking is only supported for num_bits in [1, 8], got %d", bits);
    CT_REQUIRE(rows >= 0 && cols >= 0 && words >= 0, "negative shape");
    if (rows == 0 || cols == 0) return CT_OK;
    dim3 grid = grid_rows(cdiv64(rows, 32), cols);
    CT_BITS_SWITCH(bits, hipLaunchKernelGGL((unpack_dim0_kernel<B>), grid, dim3(kBlock), 0, as_stream(stream), p, words, cols, rows,
                                            out));
    CT_LAUNCH_CHECK("ct_unpack_int32_dim0");
}

int ct_gidx_col_group(const int32_t* g_idx, int64_t cols, int64_t group_size, int32_t* col_group, int32_t* mode_word, ct_stream_t stream) {
    CT_REQUIRE(cols >= 0 && group_size > 0, "ct_gidx_col_group: cols >= 0 and group_size > 0, got %lld / %lld", (long long)cols, (long long)group_size);
    CT_REQUIRE(g_idx != nullptr && col_group != nullptr && mode_word != nullptr, "ct_gidx_col_group: null buffer");
    CT_REQUIRE(cdiv64(cols, kGidxBlock) < ((int64_t)1 << 31), "too many columns for one launch");
    if (cols == 0) return CT_OK;
    hipLaunchKernelGGL(gidx_col_group_kernel, dim3((unsigned)cdiv64(cols, kGidxBlock)), dim3(kGidxBlock), 0, as_stream(stream), g_idx, cols, group_size, mode_word, col_group);
    CT_LAUNCH_CHECK("ct_gidx_col_group");
}

}  // extern "C"
