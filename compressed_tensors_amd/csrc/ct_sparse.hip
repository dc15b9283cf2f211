// ct_sparse.hip — sparse-bitmask and sparse-24-bitmask codecs for gfx950.
//
// Formats (reference config/base.py:17-18; bit order of utils/helpers.py:306-343 =
// numpy.packbits(..., bitorder="little")): byte j of a bitmask row, bit k  <=>  x[r, 8j+k] != 0.
// compress:   values = x[x != 0] (row-major), bitmask, row_offsets = exclusive cumsum of row nnz
// decompress: out = zeros; out[mask] = values
//
// Data movement design: a workgroup owns one row "super-chunk" of 8192 columns; every lane owns
// 4 units of 8 elements spaced one block apart (one bitmask byte and one 16-byte vector of 16-bit
// elements each), so every dense-side access instruction is a contiguous 1 KiB per wave.
// Ranks come from wavefront ballots: for bit position k of the lanes' mask bytes,
// __ballot gives the 64-lane bit plane and v_mbcnt counts the set bits below the lane — 8 planes
// give the exclusive prefix of a unit with no LDS traffic and no cross-lane dependency chain;
// the per-wave totals are scalar popcounts.  Compaction/expansion goes through an LDS staging
// buffer so that the value side is ONE contiguous run per super-chunk moved with aligned
// 16-byte vectors (scalar head/tail); only LDS sees the 2-byte scattered accesses.
#include "ct_common.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <random>

namespace ct {

// ------------------------------------------------------------------------- element helpers
template <int ES> struct ElemT;
template <> struct ElemT<1> { typedef uint8_t type; };
template <> struct ElemT<2> { typedef uint16_t type; };
template <> struct ElemT<4> { typedef uint32_t type; };

// non-zero test on raw bits: floats ignore the sign bit (-0.0 == 0), NaN is non-zero
template <int ES>
__device__ __forceinline__ bool nz(typename ElemT<ES>::type bits, bool is_float) {
    if constexpr (ES == 1) return bits != 0;
    else if constexpr (ES == 2) return is_float ? (bits & 0x7fffu) != 0 : bits != 0;
    else return is_float ? (bits & 0x7fffffffu) != 0 : bits != 0;
}

// load 8 consecutive elements (vector when `vec`, else scalar with bound n)
template <int ES>
__device__ __forceinline__ void load_unit(const void* base, int64_t i0, int n, bool vec,
                                          typename ElemT<ES>::type (&e)[8]) {
    typedef typename ElemT<ES>::type T;
    const T* p = static_cast<const T*>(base) + i0;
    if (vec && n == 8) {
        if constexpr (ES == 2) {
            u32x4 w = *reinterpret_cast<const u32x4*>(p);
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { e[2 * j] = (T)(ws[j] & 0xffffu); e[2 * j + 1] = (T)(ws[j] >> 16); }
        } else if constexpr (ES == 4) {
            u32x4 a = reinterpret_cast<const u32x4*>(p)[0], b = reinterpret_cast<const u32x4*>(p)[1];
            e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
        } else {
            u32x2 w = *reinterpret_cast<const u32x2*>(p);
#pragma unroll
            for (int k = 0; k < 4; ++k) { e[k] = (T)(w.x >> (8 * k)); e[4 + k] = (T)(w.y >> (8 * k)); }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = k < n ? p[k] : (T)0;
    }
}

template <int ES>
__device__ __forceinline__ void store_unit_bits(void* base, int64_t i0, int n, bool vec,
                                                const typename ElemT<ES>::type (&e)[8]) {
    typedef typename ElemT<ES>::type T;
    T* p = static_cast<T*>(base) + i0;
    if (vec && n == 8) {
        if constexpr (ES == 2) {
            stream_store16(p, u32x4{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                                    (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)});
        } else if constexpr (ES == 4) {
            stream_store16(p, u32x4{e[0], e[1], e[2], e[3]});
            stream_store16(p + 4, u32x4{e[4], e[5], e[6], e[7]});
        } else {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo |= (uint32_t)e[k] << (8 * k); hi |= (uint32_t)e[4 + k] << (8 * k); }
            stream_store8(p, u32x2{lo, hi});
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < n) p[k] = e[k];
    }
}

// ------------------------------------------------------------------------- pack / unpack bitmasks
__global__ __launch_bounds__(kBlock) void pack_bitmasks_kernel(const uint8_t* __restrict__ mask, int64_t rows, int64_t cols,
                                                               uint8_t* __restrict__ out) {
    const int64_t bcols = (cols + 7) >> 3;
    const int64_t total = rows * bcols;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / bcols, j = i - r * bcols;
        const uint8_t* m = mask + r * cols + (j << 3);
        const int n = (int)((cols - (j << 3)) < 8 ? (cols - (j << 3)) : 8);
        uint32_t b = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < n && m[k]) b |= 1u << k;
        out[i] = (uint8_t)b;
    }
}

__global__ __launch_bounds__(kBlock) void unpack_bitmasks_kernel(const uint8_t* __restrict__ packed, int64_t rows, int64_t cols,
                                                                 uint8_t* __restrict__ mask) {
    const int64_t bcols = (cols + 7) >> 3;
    const int64_t total = rows * bcols;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = i / bcols, j = i - r * bcols;
        const uint32_t b = packed[i];
        uint8_t* m = mask + r * cols + (j << 3);
        const int n = (int)((cols - (j << 3)) < 8 ? (cols - (j << 3)) : 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < n) m[k] = (uint8_t)((b >> k) & 1u);
    }
}

// ------------------------------------------------------------------------- compress pass 1
// one workgroup per row (grid-strided): bitmask bytes + row nnz.  Streaming read, 16 B per lane.
template <int ES>
__global__ __launch_bounds__(kBlock) void bitmask_count_kernel(const void* __restrict__ x, bool is_float, int64_t rows, int64_t cols,
                                                               uint8_t* __restrict__ bitmask, int64_t* __restrict__ row_counts, int vec) {
    typedef typename ElemT<ES>::type T;
    __shared__ int s_wave[kBlock / 64];
    const int64_t bcols = (cols + 7) >> 3;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        int cnt = 0;
        for (int64_t u0 = 0; u0 < bcols; u0 += 4 * kBlock) {
            T e[4][8];
            int n[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t u = u0 + (int64_t)i * kBlock + threadIdx.x;
                const int64_t rem = cols - (u << 3);
                n[i] = (u < bcols) ? (rem >= 8 ? 8 : (int)rem) : 0;
                if (n[i] > 0) load_unit<ES>(x, row * cols + (u << 3), n[i], vec, e[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (n[i] <= 0) continue;
                const int64_t u = u0 + (int64_t)i * kBlock + threadIdx.x;
                uint32_t m = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) m |= (k < n[i] && nz<ES>(e[i][k], is_float)) ? (1u << k) : 0u;
                bitmask[row * bcols + u] = (uint8_t)m;
                cnt += __popc(m);
            }
        }
        // wave reduction, then 4 partials through LDS
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) t += s_wave[w];
            row_counts[row] = (int64_t)t;
        }
        __syncthreads();
    }
}

// popcount of each bitmask row (rebuilds row counts from a stored bitmask)
__global__ __launch_bounds__(kBlock) void bitmask_row_popcount_kernel(const uint8_t* __restrict__ bitmask, int64_t rows, int64_t cols,
                                                                      int64_t* __restrict__ row_counts) {
    __shared__ int s_wave[kBlock / 64];
    const int64_t bcols = (cols + 7) >> 3;
    const int tail_bits = (int)(cols & 7);
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        int cnt = 0;
        for (int64_t u = threadIdx.x; u < bcols; u += kBlock) {
            uint32_t m = bitmask[row * bcols + u];
            if (tail_bits && u == bcols - 1) m &= (1u << tail_bits) - 1u;
            cnt += __popc(m);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d, 64);
        if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) t += s_wave[w];
            row_counts[row] = (int64_t)t;
        }
        __syncthreads();
    }
}

// single-workgroup exclusive scan of int64 counts: every lane owns 8 consecutive counts per
// 8192-element tile (vector loads), wave scan by shuffles, 16 wave totals through LDS
__global__ __launch_bounds__(1024) void exclusive_scan_i64_kernel(const int64_t* __restrict__ counts, int64_t n,
                                                                  int64_t* __restrict__ offsets, int64_t* __restrict__ total) {
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    constexpr int PER = 8;
    for (int64_t base = 0; base < n; base += 1024 * PER) {
        const int64_t i0 = base + (int64_t)threadIdx.x * PER;
        int64_t v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) v[k] = (i0 + k < n) ? counts[i0 + k] : 0;
        int64_t sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) sum += v[k];
        int64_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int64_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int64_t wbase = s_carry, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            int64_t t = s_wave[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        int64_t run = wbase + inc - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (i0 + k < n) offsets[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = s_carry;
}

// ------------------------------------------------------------------------- ballot ranks
constexpr int kUPL = 4;                    // units per lane
constexpr int kSuper = kBlock * kUPL * 8;  // columns per workgroup iteration (8192)

// exclusive rank of this lane's unit among the wave's units (sum of mask popcounts of lower
// lanes) and the wave total, from 8 ballot bit planes
__device__ __forceinline__ void wave_rank(uint32_t m, uint32_t& pre, int& wave_total) {
    pre = 0;
    wave_total = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long plane = __ballot((m >> k) & 1u);
        pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(plane >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)plane, pre));
        wave_total += __popcll(plane);
    }
}

// ranks of the kUPL units of every lane inside the super-chunk, in unit order i*kBlock + lane.
// s_tot: [kUPL][waves] wave totals.  Returns the super-chunk total.
__device__ __forceinline__ int superchunk_ranks(const uint32_t (&m)[kUPL], int (*s_tot)[kBlock / 64], uint32_t (&rank)[kUPL]) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kUPL; ++i) {
        int wt;
        wave_rank(m[i], rank[i], wt);
        if ((threadIdx.x & 63) == 0) s_tot[i][wave] = wt;
    }
    __syncthreads();
    int running = 0;
#pragma unroll
    for (int i = 0; i < kUPL; ++i) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            const int t = s_tot[i][w];
            if (w == wave) rank[i] += running;  // everything before (i, wave) in unit order
            running += t;
        }
    }
    return running;
}

// ------------------------------------------------------------------------- compress pass 2
// one workgroup per row: compact each super-chunk into LDS, then copy the contiguous run to
// values[] with aligned 16-byte stores
template <int ES>
__global__ __launch_bounds__(kBlock) void bitmask_scatter_kernel(const void* __restrict__ x, bool is_float, int64_t rows, int64_t cols,
                                                                 const int64_t* __restrict__ row_offsets, void* __restrict__ values, int vec) {
    typedef typename ElemT<ES>::type T;
    __shared__ __attribute__((aligned(16))) T s_val[kSuper];
    __shared__ int s_tot[kUPL][kBlock / 64];
    T* vout = static_cast<T*>(values);
    constexpr int VE = 16 / ES;  // elements per 16-byte vector
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        int64_t run = row_offsets[row];
        for (int64_t cbase = 0; cbase < cols; cbase += kSuper) {
            T e[kUPL][8];
            uint32_t m[kUPL];
#pragma unroll
            for (int i = 0; i < kUPL; ++i) {
                const int64_t c0 = cbase + (((int64_t)i * kBlock + threadIdx.x) << 3);
                const int64_t rem = cols - c0;
                const int n = rem >= 8 ? 8 : (rem > 0 ? (int)rem : 0);
                m[i] = 0;
                if (n > 0) {
                    load_unit<ES>(x, row * cols + c0, n, vec, e[i]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) m[i] |= (k < n && nz<ES>(e[i][k], is_float)) ? (1u << k) : 0u;
                }
            }
            uint32_t rank[kUPL];
            const int total = superchunk_ranks(m, s_tot, rank);
#pragma unroll
            for (int i = 0; i < kUPL; ++i) {
                int pos = (int)rank[i];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (m[i] & (1u << k)) s_val[pos++] = e[i][k];
            }
            __syncthreads();
            // copy s_val[0, total) -> vout[run, run+total): scalar head up to a 16-byte boundary
            // of the destination, vector body (LDS side read element-wise: its start is not
            // 16-byte aligned in general, and LDS is cheap while HBM is not), scalar tail
            const uintptr_t dst = reinterpret_cast<uintptr_t>(vout + run);
            int head = (int)(((16 - (dst & 15u)) & 15u) / ES);
            if (head > total) head = total;
            const int body_vecs = (total - head) / VE;
            for (int i = threadIdx.x; i < head; i += kBlock) vout[run + i] = s_val[i];
            for (int v = threadIdx.x; v < body_vecs; v += kBlock) {
                const T* sp = s_val + head + v * VE;
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (ES == 2) w[j] = (uint32_t)sp[2 * j] | ((uint32_t)sp[2 * j + 1] << 16);
                    else if constexpr (ES == 4) w[j] = sp[j];
                    else w[j] = (uint32_t)sp[4 * j] | ((uint32_t)sp[4 * j + 1] << 8) | ((uint32_t)sp[4 * j + 2] << 16) | ((uint32_t)sp[4 * j + 3] << 24);
                }
                stream_store16(vout + run + head + (int64_t)v * VE, u32x4{w[0], w[1], w[2], w[3]});
            }
            for (int i = head + body_vecs * VE + threadIdx.x; i < total; i += kBlock) vout[run + i] = s_val[i];
            run += total;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------- decompress
// per-workgroup lookup tables indexed by a bitmask byte: rank of each of its 8 bits (8 bytes) and,
// for 16-bit payloads, the four dword masks that zero the elements whose bit is clear
struct ExpandLut {
    uint32_t rank[256 * 2];
    uint32_t keep[256 * 4];
};

__device__ __forceinline__ void build_expand_lut(ExpandLut& lut) {
    const uint32_t mv = threadIdx.x;  // kBlock == 256 entries
    uint32_t r_lo = 0, r_hi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t rk = __popc(mv & ((1u << k) - 1u));
        if (k < 4) r_lo |= rk << (8 * k);
        else r_hi |= rk << (8 * (k - 4));
    }
    lut.rank[mv * 2] = r_lo;
    lut.rank[mv * 2 + 1] = r_hi;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        lut.keep[mv * 4 + j] = (((mv >> (2 * j)) & 1u) ? 0xffffu : 0u) | (((mv >> (2 * j + 1)) & 1u) ? 0xffff0000u : 0u);
}

// one workgroup per row: the super-chunk's contiguous value run is staged in LDS with aligned
// 16-byte loads, then every lane expands its 4 bitmask bytes into four 16-byte dense stores using
// the lookup tables (8 LDS gathers + 2 table reads per unit, ~4 VALU ops per element).
// When the row fits one super-chunk the run length is known up front (next row offset), so the
// value loads are issued together with the mask loads and overlap the ballots/ranks.
template <int ES>
__global__ __launch_bounds__(kBlock) void bitmask_decompress_kernel(const void* __restrict__ values, int64_t values_len,
                                                                    const uint8_t* __restrict__ bitmask,
                                                                    const int64_t* __restrict__ row_offsets, int64_t fixed_row_nnz,
                                                                    int64_t rows, int64_t cols, void* __restrict__ out, int vec_out,
                                                                    int vec_in) {
    typedef typename ElemT<ES>::type T;
    constexpr int VE = 16 / ES;
    constexpr int kStage = kSuper + 2 * VE;
    __shared__ __attribute__((aligned(16))) T s_val[kStage];
    __shared__ __attribute__((aligned(16))) ExpandLut s_lut;
    __shared__ int s_tot[kUPL][kBlock / 64];
    build_expand_lut(s_lut);  // visible after the first __syncthreads below
    // 2:4-regular rows (every nibble of the row's bitmask has exactly two bits: what a 2:4 codec writes): the values of quad q are
    // the row's values 2q and 2q + 1, no prefix is needed, and a lane turns 8 bytes of values + its mask bits into one 16-byte store
    // with byte permutes.  Checked per row (one pass over <= cols / 8 mask bytes + a barrier); any other row takes the general
    // path below.  Selector tables: [j][nibble] for byte payloads (the pair sits in bytes 2j, 2j+1 of its dword), [nibble][2]
    // for 16-bit payloads (v0 = bytes 0-1, v1 = bytes 2-3); 0x0c selects a zero byte.
    __shared__ uint32_t s_sel[64];
    if (threadIdx.x < 32) {
        const uint32_t nb = threadIdx.x & 15u, j = threadIdx.x >> 4;
        uint32_t sel8 = 0, sel16a = 0, sel16b = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool bit = (nb >> k) & 1u;
            const uint32_t rk = __popc(nb & ((1u << k) - 1u)) ? 1u : 0u;
            sel8 |= (bit ? (2u * j + rk) : 0x0cu) << (8 * k);
            const uint32_t h = bit ? ((2u * rk) | ((2u * rk + 1u) << 8)) : 0x0c0cu;  // the two bytes of element k
            if (k < 2) sel16a |= h << (16 * k);
            else sel16b |= h << (16 * (k - 2));
        }
        s_sel[threadIdx.x] = sel8;                 // [j * 16 + nibble]
        if (j == 0) {
            s_sel[32 + 2 * nb] = sel16a;           // [32 + 2 * nibble + {0, 1}]
            s_sel[32 + 2 * nb + 1] = sel16b;
        }
    }
    const T* vin = static_cast<const T*>(values);
    const int64_t bcols = (cols + 7) >> 3;
    const bool single = cols <= kSuper;
    const bool try_regular = vec_out && cols % 16 == 0 && (reinterpret_cast<uintptr_t>(bitmask) & 1u) == 0 && (reinterpret_cast<uintptr_t>(vin) & 7u) == 0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        int64_t run = row_offsets ? row_offsets[row] : row * fixed_row_nnz;
        if (try_regular) {  // workgroup-uniform
            const uint16_t* mrow = reinterpret_cast<const uint16_t*>(bitmask + row * bcols);
            bool reg = ((run * ES) & 7) == 0 && run >= 0 && run + (cols >> 1) <= values_len;
            for (int64_t i = threadIdx.x; i < (bcols >> 1); i += kBlock) {
                const uint32_t b = mrow[i];
                uint32_t t = b - ((b >> 1) & 0x5555u);
                t = (t & 0x3333u) + ((t >> 2) & 0x3333u);  // popcount of each nibble
                reg = reg && (t == 0x2222u);
            }
            if (__syncthreads_and(reg)) {
                const u32x2* vrow = reinterpret_cast<const u32x2*>(vin + run);
                u32x4* orow = reinterpret_cast<u32x4*>(static_cast<T*>(out) + row * cols);
                const int64_t nvec = cols * ES / 16;
                for (int64_t v = threadIdx.x; v < nvec; v += kBlock) {
                    const u32x2 val = vrow[v];
                    u32x4 o;
                    if constexpr (ES == 1) {
                        const uint32_t mm = mrow[v];
                        o.x = __builtin_amdgcn_perm(0u, val.x, s_sel[mm & 15u]);
                        o.y = __builtin_amdgcn_perm(0u, val.x, s_sel[16 + ((mm >> 4) & 15u)]);
                        o.z = __builtin_amdgcn_perm(0u, val.y, s_sel[(mm >> 8) & 15u]);
                        o.w = __builtin_amdgcn_perm(0u, val.y, s_sel[16 + (mm >> 12)]);
                    } else if constexpr (ES == 2) {
                        const uint32_t mm = reinterpret_cast<const uint8_t*>(mrow)[v];
                        const uint32_t n0 = mm & 15u, n1 = mm >> 4;
                        o.x = __builtin_amdgcn_perm(0u, val.x, s_sel[32 + 2 * n0]);
                        o.y = __builtin_amdgcn_perm(0u, val.x, s_sel[32 + 2 * n0 + 1]);
                        o.z = __builtin_amdgcn_perm(0u, val.y, s_sel[32 + 2 * n1]);
                        o.w = __builtin_amdgcn_perm(0u, val.y, s_sel[32 + 2 * n1 + 1]);
                    } else {
                        const uint32_t nb = (reinterpret_cast<const uint8_t*>(mrow)[v >> 1] >> (4 * (int)(v & 1))) & 15u;
                        uint32_t e[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t src = (nb & ((1u << k) - 1u)) ? val.y : val.x;  // a second kept element of the quad takes v1
                            e[k] = ((nb >> k) & 1u) ? src : 0u;
                        }
                        o = u32x4{e[0], e[1], e[2], e[3]};
                    }
                    stream_store16(orow + v, o);
                }
                continue;
            }
        }
        int64_t row_end = values_len;
        if (single) {
            if (row_offsets) { if (row + 1 < rows) row_end = row_offsets[row + 1]; }
            else row_end = run + fixed_row_nnz;
            if (row_end > values_len) row_end = values_len;
            if (row_end < run) row_end = run;
        }
        for (int64_t cbase = 0; cbase < cols; cbase += kSuper) {
            uint32_t m[kUPL];
#pragma unroll
            for (int i = 0; i < kUPL; ++i) {
                const int64_t c0 = cbase + (((int64_t)i * kBlock + threadIdx.x) << 3);
                const int64_t rem = cols - c0;
                m[i] = 0;
                if (rem > 0) {
                    m[i] = bitmask[row * bcols + (c0 >> 3)];
                    if (rem < 8) m[i] &= (1u << (int)rem) - 1u;
                }
            }
            // stage vin[run, run+len) at s_val[shift ...], shift = run's offset inside its 16-byte
            // vector, so that the global side of the staging loads is aligned
            auto stage = [&](int len) -> int {
                const uintptr_t src = reinterpret_cast<uintptr_t>(vin + run);
                const int shift = vec_in ? (int)((src & 15u) / ES) : 0;
                if (vec_in) {
                    const int nvec = (shift + len + VE - 1) / VE;
                    const int64_t e0 = run - shift;  // >= 0: the buffer base is 16-byte aligned
                    const u32x4* g = reinterpret_cast<const u32x4*>(vin + e0);
                    for (int v = threadIdx.x; v < nvec; v += kBlock) {
                        if (e0 + (int64_t)(v + 1) * VE <= values_len) {
                            reinterpret_cast<u32x4*>(s_val)[v] = g[v];
                        } else {  // a tail vector that would cross the end of the buffer
                            for (int j = 0; j < VE; ++j) {
                                const int64_t gi = e0 + (int64_t)v * VE + j;
                                s_val[v * VE + j] = gi < values_len ? vin[gi] : (T)0;
                            }
                        }
                    }
                } else {
                    for (int i = threadIdx.x; i < len; i += kBlock) s_val[i] = vin[run + i];
                }
                return shift;
            };
            int shift = 0;
            if (single) {
                int len = (int)(row_end - run);
                if (len > kSuper) len = kSuper;
                shift = stage(len);
            }
            uint32_t rank[kUPL];
            const int total = superchunk_ranks(m, s_tot, rank);  // contains a __syncthreads
            if (!single) {
                int len = total;
                if (run + len > values_len) len = (int)(values_len > run ? values_len - run : 0);
                shift = stage(len);
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < kUPL; ++i) {
                const int64_t c0 = cbase + (((int64_t)i * kBlock + threadIdx.x) << 3);
                const int64_t rem = cols - c0;
                const int n = rem >= 8 ? 8 : (rem > 0 ? (int)rem : 0);
                if (n == 0) continue;
                const uint32_t mv = m[i];
                const u32x2 rk = *reinterpret_cast<const u32x2*>(&s_lut.rank[mv * 2]);
                // a corrupt bitmask / offset pair must not read outside the staging buffer
                int p = shift + (int)rank[i];
                p = p > kStage - 8 ? kStage - 8 : p;
                const T* sp = s_val + p;
                T e[8];
                if constexpr (ES == 2) {
                    const u32x4 keep = *reinterpret_cast<const u32x4*>(&s_lut.keep[mv * 4]);
                    const uint32_t kk[4] = {keep.x, keep.y, keep.z, keep.w};
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t rr = j < 2 ? rk.x : rk.y;
                        const uint32_t r0 = (rr >> (16 * (j & 1))) & 0xffu, r1 = (rr >> (16 * (j & 1) + 8)) & 0xffu;
                        w[j] = ((uint32_t)sp[r0] | ((uint32_t)sp[r1] << 16)) & kk[j];
                    }
                    if (vec_out && n == 8) {
                        stream_store16(static_cast<T*>(out) + row * cols + c0, u32x4{w[0], w[1], w[2], w[3]});
                        continue;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { e[2 * j] = (T)(w[j] & 0xffffu); e[2 * j + 1] = (T)(w[j] >> 16); }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t r = ((k < 4 ? rk.x : rk.y) >> (8 * (k & 3))) & 0xffu;
                        const T v = sp[r];
                        e[k] = ((mv >> k) & 1u) ? v : (T)0;
                    }
                }
                store_unit_bits<ES>(out, row * cols + c0, n, vec_out, e);
            }
            run += total;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------- decompress, 16-bit fast path
// The generic kernel above ranks with 8 ballot planes per unit and gathers element by element
// (~25 VALU lane-ops per element: the kernel was issue-bound, not bandwidth-bound).  For 16-bit
// payloads with cols % 32 == 0 this kernel does the same job in ~5:
//  * mask side: a lane loads ONE dword of the bitmask (4 consecutive units), popcounts it; a DPP
//    wave scan + 4 wave totals give every unit's rank; the owner lane publishes
//    the 4 ranks in LDS and the consumer lane (unit i*256 + tid, so that every wave store
//    instruction writes 1 KiB contiguous) reads its rank back and loads its own mask byte.
//  * value side: output dword j of a unit (elements 2j, 2j+1) is a 32-bit WINDOW of the staged
//    value run starting at element q_j = rank + popc(m & ((1 << 2j) - 1)): two aligned LDS dwords
//    (one ds_read2_b32) and ONE v_perm_b32 whose selector does the funnel shift, places a lone
//    element in the right half and zeroes the rest.  The selector depends only on the two mask bits
//    and on whether the window starts mid-dword: an 8-entry LDS table (8 banks, conflict-free).  A
//    256 x 2-parity x 4 table keyed by the whole mask byte was 6 % slower: random 16-byte LUT reads
//    were half of all LDS cycles as bank conflicts, and its 9 KB cost two workgroups per CU.
// A tile is 8192 columns of one row.  SINGLE (cols <= 8192): the run length is known from the row
// offsets, so the value loads are issued together with the mask load and there is one barrier.
// Otherwise the tile's start inside the row is the popcount of the row's earlier mask bytes and
// the values are staged after the ranks (two barriers).
// Every LDS address is derived from the mask alone (rank <= 8192), so corrupt offsets can produce
// wrong data but never an out-of-range access; global reads are guarded by values_len.
constexpr int kTile16 = 8192;

__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// every bit of a 16-bit value doubled (bit k -> bits 2k, 2k + 1): the element mask of 32-bit payloads as a mask of their halves
__device__ __forceinline__ uint32_t double_bits16(uint32_t x) {
    x = (x | (x << 8)) & 0x00ff00ffu;
    x = (x | (x << 4)) & 0x0f0f0f0fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x | (x << 1);
}

// ------------------------------------------------------------------------------------------
// 8-bit payloads (round 6): the same LDS-window decompress at BYTE granularity — FP8 / int8 weights of a sparse checkpoint, moved as raw
// bytes.  A unit is still one 16-byte store = 16 elements = TWO mask bytes; a tile is 1024 units = 16384 columns of a row.  Mask side: a
// lane owns 4 consecutive units = two mask dwords, popcounts them, a DPP wave scan + 4 wave totals rank every unit, the owner publishes
// the unit's rank through LDS to the consumer lane (unit i * 256 + tid: a wave store instruction writes 1 KiB contiguous).  Value side:
// the tile's value run is staged global -> LDS; output dword j of a unit (elements 4j .. 4j + 3) is a window of the run starting at byte
// rank + popc(mask16 & ((1 << 4j) - 1)): two aligned LDS dwords and ONE v_perm_b32 whose selector — a 64-entry LDS table keyed by the
// window's alignment (2 bits) and the four mask bits — funnel-shifts, places up to four kept bytes and zeroes the rest.
// The general kernel (one element at a time through a per-row prefix) ran 8192^2 int8 at 47 us = 29 % of the HBM peak.
// Needs cols % 64 == 0 (whole lanes of 4 units, dword-aligned mask rows), 16-byte aligned values / out, 4-byte aligned bitmask.
// ------------------------------------------------------------------------------------------
// UPL = units per lane (4, 2 or 1): a tile is 256 x UPL units, so that a row of 8192 (4096) columns — every FP8 weight of an 8K (4K) model — is ONE
// FULL tile instead of half (a quarter) of a 16384-column one: with UPL = 4 at 8192 columns half the lanes idled through the barriers and the scan
// (35.5 us at 8192^2).
// FLAT (rows shorter than a tile, cols % 32 == 0): tiles of 1024 consecutive units of the flattened tensor, as the 16-bit kernel's flat form.
template <bool SINGLE, int UPL, bool FLAT = false>
__global__ __launch_bounds__(kBlock) void bitmask_decompress8_kernel(const uint8_t* __restrict__ vin, int64_t values_len, const uint8_t* __restrict__ bitmask,
                                                                     const int64_t* __restrict__ row_offsets, int64_t fixed_row_nnz, int64_t rows, int64_t cols,
                                                                     uint8_t* __restrict__ out) {
    static_assert(!FLAT || (!SINGLE && UPL == 4), "flat tiles are full-size tiles with the row prefix of the several-tiles form");
    constexpr int kUnits = kBlock * UPL;   // units per tile
    constexpr int kTile8 = kUnits * 16;    // elements per tile
    __shared__ __attribute__((aligned(16))) uint8_t s_val[kTile8 + 64];
    __shared__ uint32_t s_sel[64];  // v_perm_b32 selectors indexed by (window offset & 3) << 4 | mask nibble
    __shared__ __attribute__((aligned(16))) uint16_t s_rank[kUnits];  // wave-local rank of every unit (< 1024 x UPL)
    __shared__ __attribute__((aligned(16))) int s_tot[4];
    __shared__ __attribute__((aligned(16))) int s_pre[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (threadIdx.x < 64) {  // visible after the first barrier below
        const uint32_t x = threadIdx.x >> 4, nb = threadIdx.x & 15u;
        uint32_t sel = 0, src = x;  // bytes 0-3 of the selector's source = the low dword, 4-7 the high dword; 0x0c = constant zero
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool keep = (nb >> k) & 1u;
            sel |= (keep ? src : 0x0cu) << (8 * k);
            src += keep ? 1u : 0u;
        }
        s_sel[threadIdx.x] = sel;
    }
    const int64_t bcols = cols >> 4;  // units per row
    const int64_t tiles_per_row = (cols + kTile8 - 1) / kTile8;
    const int64_t ntiles = FLAT ? (rows * bcols + kUnits - 1) / kUnits : rows * tiles_per_row;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = FLAT ? (tile * kUnits) / bcols : (SINGLE ? tile : tile / tiles_per_row);
        const int64_t u0 = FLAT ? tile * kUnits - row * bcols : (SINGLE ? 0 : (tile - row * tiles_per_row) * kUnits);  // first unit of the tile in its row
        const int64_t left = FLAT ? rows * bcols - tile * kUnits : bcols - u0;
        const int nu = left < kUnits ? (int)left : kUnits;                      // units in this tile
        const uint8_t* mrow = bitmask + row * (cols >> 3);                      // two mask bytes per unit
        // the lane's own UPL consecutive units UPL tid .. UPL tid + UPL - 1 (the rank side)
        uint32_t own[UPL];
#pragma unroll
        for (int k = 0; k < UPL; ++k) own[k] = (UPL * tid + k < nu) ? (uint32_t)*reinterpret_cast<const uint16_t*>(mrow + 2 * (u0 + UPL * tid + k)) : 0u;
        uint32_t mb[UPL];  // the consumer lane's units i * 256 + tid: their 16 mask bits in the low half, the window offsets of dwords 1-3 above them
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int64_t v = u0 + i * kBlock + tid;
            const uint32_t m16 = (i * kBlock + tid < nu) ? (uint32_t)*reinterpret_cast<const uint16_t*>(mrow + 2 * v) : 0u;
            // (consumed here, ahead of the LDS-direct loads: hipcc waits vmcnt(0) at the first use of an ordinary load while one of those is in flight)
            mb[i] = m16 | ((uint32_t)__popc(m16 & 0xfu) << 16) | ((uint32_t)__popc(m16 & 0xffu) << 20) | ((uint32_t)__popc(m16 & 0xfffu) << 24);
        }
        int64_t run = row_offsets ? row_offsets[row] : row * fixed_row_nnz;
        run = run < 0 ? 0 : (run > values_len ? values_len : run);

        int shift = 0, nvec = 0;
        int64_t e0 = 0;
        auto stage_issue = [&](int total) {
            shift = (int)(run & 15);
            nvec = (shift + total + 15) >> 4;
            e0 = run - shift;
            const u32x4* g = reinterpret_cast<const u32x4*>(vin + e0);
#pragma unroll
            for (int k = 0; k < UPL; ++k) {
                const int v = tid + k * kBlock;
                if (v < nvec && e0 + (int64_t)(v + 1) * 16 <= values_len)
                    __builtin_amdgcn_global_load_lds(g + v, (__attribute__((address_space(3))) void*)(s_val + (k * kBlock + wave * 64) * 16), 16, 0, 0);
            }
        };
        auto stage_commit = [&]() {
#pragma unroll
            for (int k = 0; k < UPL; ++k) {
                const int v = tid + k * kBlock;
                if (v < nvec && e0 + (int64_t)(v + 1) * 16 > values_len) {  // a tail vector that would cross the end of the buffer
#pragma unroll 1
                    for (int j = 0; j < 16; ++j) {
                        const int64_t gi = e0 + (int64_t)v * 16 + j;
                        s_val[v * 16 + j] = gi < values_len ? vin[gi] : (uint8_t)0;
                    }
                }
            }
            if (tid == 0 && kUnits < nvec) {  // the one possible extra vector (shift > 0, full tile)
                const int v = kUnits;
#pragma unroll 1
                for (int j = 0; j < 16; ++j) {
                    const int64_t gi = e0 + (int64_t)v * 16 + j;
                    s_val[v * 16 + j] = gi < values_len ? vin[gi] : (uint8_t)0;
                }
            }
        };

        if constexpr (!SINGLE) {
            int c = 0;  // popcount of the row's mask bytes before this tile
            for (int64_t d = tid; d < (u0 >> 1); d += kBlock) c += __popc(reinterpret_cast<const uint32_t*>(mrow)[d]);
            c = wave_incl_scan(c);
            if (lane == 63) s_pre[wave] = c;
        }
        int c = 0;
#pragma unroll
        for (int k = 0; k < UPL; ++k) c += __popc(own[k]);
        const int incl = wave_incl_scan(c);
        if (lane == 63) s_tot[wave] = incl;
        {
            uint32_t r = (uint32_t)(incl - c);
#pragma unroll
            for (int k = 0; k < UPL; ++k) {
                s_rank[UPL * tid + k] = (uint16_t)r;
                r += __popc(own[k]);
            }
        }
        if constexpr (SINGLE) {
            int64_t row_end = values_len;
            if (row_offsets) { if (row + 1 < rows) row_end = row_offsets[row + 1]; }
            else row_end = run + fixed_row_nnz;
            int64_t len = row_end - run;
            len = len < 0 ? 0 : (len > kTile8 ? kTile8 : len);
            stage_issue((int)len);
            stage_commit();
        }
        __syncthreads();
        const int t0 = s_tot[0], t1 = s_tot[1], t2 = s_tot[2], t3 = s_tot[3];
        if constexpr (!SINGLE) {
            run += (int64_t)s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
            run = run > values_len ? values_len : run;
            stage_issue(t0 + t1 + t2 + t3);
            stage_commit();
            __syncthreads();
        }
        const int wprefix[4] = {0, t0, t0 + t1, t0 + t1 + t2};  // values before wave w's units
        const char* sv = reinterpret_cast<const char*>(s_val);
        uint8_t* orow = out + row * cols + (u0 << 4);
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int u = i * kBlock + tid;
            if (u >= nu) continue;
            // the owner of unit u is lane u / UPL: wave (u / UPL) / 64
            const int ow = (u / UPL) >> 6;
            const int wb = ow == 0 ? wprefix[0] : (ow == 1 ? wprefix[1] : (ow == 2 ? wprefix[2] : wprefix[3]));
            const uint32_t mv = mb[i] & 0xffffu;
            const uint32_t a0 = (uint32_t)s_rank[u] + (uint32_t)(wb + shift);  // byte offset of the unit's first kept element in the staged run
            const uint32_t offs[4] = {0u, (mb[i] >> 16) & 0xfu, (mb[i] >> 20) & 0xfu, (mb[i] >> 24) & 0xfu};
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t aj = a0 + offs[j];
                const uint32_t a = aj & ~3u;
                const uint32_t lo = *reinterpret_cast<const uint32_t*>(sv + a), hi = *reinterpret_cast<const uint32_t*>(sv + a + 4);
                w[j] = __builtin_amdgcn_perm(hi, lo, s_sel[((aj & 3u) << 4) | ((mv >> (4 * j)) & 15u)]);
            }
            stream_store16(orow + ((int64_t)u << 4), u32x4{w[0], w[1], w[2], w[3]});
        }
        __syncthreads();
    }
}

// ES = 4 (round 3): 32-bit payloads as pairs of 16-bit halves — the value run, the ranks, the windows and the stores are those of a
// 16-bit tensor with twice the columns whose mask has every bit doubled; only the mask loads (half as many real bytes) and the
// offsets (doubled on the way in) differ.
// Round 6: (1) the body is a device function over the tiles tile_first, tile_first + tile_step, ... — the single-tensor kernel below hands it
// blockIdx.x / gridDim.x, the table kernel (ct_bitmask_decompress_batch) one tile of a table row's tensor; its LDS is the caller's (one struct whatever the
// instantiation, so that a kernel that calls several of them does not add up their static arrays).
// (2) FORM 2, "flat" (16-bit payloads whose rows are SHORTER than a tile): a tile is 1024 consecutive units of the FLATTENED tensor, whatever rows they
// belong to.  The mask bytes, the outputs and — because a row's values start where the previous row's end: row_offsets is the exclusive prefix sum of the
// rows' counts, fixed_row_nnz trivially so — the value run of such a tile are all contiguous, so the only thing a tile needs from the row structure is where
// its FIRST unit's values start: row_offsets[row] + the popcount of that row's mask bytes in front of the tile (the prefix loop of the several-tiles-per-row
// form).  With one row per workgroup a 2048-column row moved 6 KB per workgroup behind two dependent memory latencies and a TinyLlama-shaped checkpoint
// decompressed at 2.5 TB/s — launches and tables alike (smaller tiles for short rows, 256 / 512 units, changed nothing: measured, dropped).
struct Decomp16Lds {
    __attribute__((aligned(16))) uint16_t val[kTile16 + 32];
    uint32_t sel8[8];  // v_perm_b32 selectors indexed by (window misaligned) << 2 | (b1 << 1) | b0
    __attribute__((aligned(16))) uint16_t rank[kTile16 / 8];  // wave-local rank of every unit
    __attribute__((aligned(16))) int tot[4];
    __attribute__((aligned(16))) int pre[4];
};

constexpr int kDecRows = 0, kDecSingle = 1, kDecFlat = 2;  // FORM: rows of several tiles / the row is one tile / flat tiles over short rows
template <int FORM, int ES>
__device__ __forceinline__ void bitmask_decompress16_tiles(Decomp16Lds& lds, const int64_t tile_first, const int64_t tile_step, const uint16_t* __restrict__ vin,
                                                           int64_t values_len_e, const uint8_t* __restrict__ bitmask, const int64_t* __restrict__ row_offsets,
                                                           int64_t fixed_row_nnz_e, int64_t rows, int64_t cols_e, uint16_t* __restrict__ out) {
    static_assert(FORM != kDecFlat || ES == 2, "the flat form moves 16-bit payloads");
    constexpr bool SINGLE = FORM == kDecSingle, FLAT = FORM == kDecFlat;
    constexpr int UPL = 4;
    constexpr int SH = ES == 4 ? 1 : 0;
    constexpr int kUnits = kBlock * UPL, kTile = kUnits * 8;  // units / 16-bit items per tile
    const int64_t values_len = values_len_e << SH, cols = cols_e << SH, fixed_row_nnz = fixed_row_nnz_e << SH;  // in 16-bit items from here on
    uint16_t* const s_val = lds.val;
    uint32_t* const s_sel8 = lds.sel8;
    uint16_t* const s_rank = lds.rank;
    int* const s_tot = lds.tot;
    int* const s_pre = lds.pre;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (threadIdx.x < 8) {  // visible after the first barrier below
        const uint32_t x = threadIdx.x >> 2, b0 = threadIdx.x & 1u, b1 = (threadIdx.x >> 1) & 1u;
        const uint32_t w01 = x ? 0x0302u : 0x0100u, w23 = x ? 0x0504u : 0x0302u;  // window bytes 0,1 / 2,3 (0x0c selects 0x00)
        s_sel8[threadIdx.x] = (b0 && b1) ? (w01 | (w23 << 16)) : b0 ? (w01 | 0x0c0c0000u) : b1 ? (0x0c0cu | (w01 << 16)) : 0x0c0c0c0cu;
    }
    const int64_t bcols = cols >> 3;
    const int64_t tiles_per_row = (cols + kTile - 1) / kTile;
    const int64_t ntiles = FLAT ? (rows * bcols + kUnits - 1) / kUnits : rows * tiles_per_row;
    for (int64_t tile = tile_first; tile < ntiles; tile += tile_step) {
        // FLAT: the tile's first unit is unit u0 of row `row`, and its units run on through the following rows (everything below addresses them
        // relative to the row's start, which is contiguous with the next row's)
        const int64_t row = FLAT ? (tile * kUnits) / bcols : (SINGLE ? tile : tile / tiles_per_row);
        const int64_t u0 = FLAT ? tile * kUnits - row * bcols : (SINGLE ? 0 : (tile - row * tiles_per_row) * kUnits);  // first unit of the tile in its row
        const int64_t left = FLAT ? rows * bcols - tile * kUnits : bcols - u0;
        const int nu = left < kUnits ? (int)left : kUnits;                      // units in this tile (a multiple of 4)
        const uint8_t* mrow = bitmask + row * (bcols >> SH);  // the real mask row: one byte per unit (ES = 2) / per two units (ES = 4)
        uint32_t md = 0;  // the mask bytes of the lane's own 4 consecutive units 4 tid .. (the rank side), one per byte
        if (UPL * tid < nu) {
            if constexpr (ES == 4) md = double_bits16(*reinterpret_cast<const uint16_t*>(mrow + ((u0 + 4 * tid) >> 1)));
            else md = *reinterpret_cast<const uint32_t*>(mrow + u0 + 4 * tid);
        }
        // the consumer lane's own mask bytes (unit i*256 + tid): same cache lines as the load above
        uint32_t mb[UPL];
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int64_t v = u0 + i * kBlock + tid;
            mb[i] = 0u;
            if (i * kBlock + tid < nu) {
                if constexpr (ES == 4) mb[i] = double_bits16(((uint32_t)mrow[v >> 1] >> (4 * (int)(v & 1))) & 0xfu);
                else mb[i] = (uint32_t)mrow[v];
            }
        }
        int64_t run = row_offsets ? (row_offsets[row] << SH) : row * fixed_row_nnz;
        run = run < 0 ? 0 : (run > values_len ? values_len : run);

        // stage vin[run, run + total) at s_val[shift ...]; shift = offset of `run` inside its 16-byte
        // vector, so that the global side of the staging loads is aligned
        int shift = 0, nvec = 0;
        int64_t e0 = 0;
        // the value run goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write pass; the
        // LDS image is lane-linear, which is exactly the order of the 16-byte vectors).  Vectors that would cross the end
        // of the buffer are filled element-wise afterwards.
        auto stage_issue = [&](int total) {
            shift = (int)(run & 7);
            nvec = (shift + total + 7) >> 3;
            e0 = run - shift;
            const u32x4* g = reinterpret_cast<const u32x4*>(vin + e0);
#pragma unroll
            for (int k = 0; k < UPL; ++k) {
                const int v = tid + k * kBlock;
                if (v < nvec && e0 + (int64_t)(v + 1) * 8 <= values_len)
                    __builtin_amdgcn_global_load_lds(g + v, (__attribute__((address_space(3))) void*)(s_val + (k * kBlock + wave * 64) * 8), 16, 0, 0);
            }
        };
        auto stage_commit = [&]() {
#pragma unroll
            for (int k = 0; k < UPL; ++k) {
                const int v = tid + k * kBlock;
                if (v < nvec && e0 + (int64_t)(v + 1) * 8 > values_len) {  // a tail vector that would cross the end of the buffer
#pragma unroll 1
                    for (int j = 0; j < 8; ++j) {
                        const int64_t gi = e0 + (int64_t)v * 8 + j;
                        s_val[v * 8 + j] = gi < values_len ? vin[gi] : (uint16_t)0;
                    }
                }
            }
            if (tid == 0 && kUnits < nvec) {  // the one possible extra vector (shift > 0, full tile)
                const int v = kUnits;
#pragma unroll 1
                for (int j = 0; j < 8; ++j) {
                    const int64_t gi = e0 + (int64_t)v * 8 + j;
                    s_val[v * 8 + j] = gi < values_len ? vin[gi] : (uint16_t)0;
                }
            }
        };

        if constexpr (!SINGLE) {
            // popcount of the row's mask bytes before this tile
            int c = 0;
            for (int64_t d = tid; d < (u0 >> (2 + SH)); d += kBlock) c += __popc(reinterpret_cast<const uint32_t*>(mrow)[d]) << SH;
            c = wave_incl_scan(c);
            if (lane == 63) s_pre[wave] = c;
        }

        // ranks: wave-local exclusive prefix of the lane's mask bytes + popcounts of its lower bytes.  Every ordinary load of
        // this tile (mask bytes, row offsets) is consumed BEFORE the LDS-direct loads are issued: hipcc waits
        // vmcnt(0) at the first use of an ordinary load while an LDS-direct load is in flight
        const int c = __popc(md);
        const int incl = wave_incl_scan(c);
        if (lane == 63) s_tot[wave] = incl;
        const uint32_t r0 = (uint32_t)(incl - c);
        const uint32_t r1 = r0 + __popc(md & 0xffu), r2 = r0 + __popc(md & 0xffffu), r3 = r0 + __popc(md & 0xffffffu);
        reinterpret_cast<u32x2*>(s_rank)[tid] = u32x2{r0 | (r1 << 16), r2 | (r3 << 16)};
        uint32_t offs4[UPL];  // window offsets of the lane's units, from its mask bytes
#pragma unroll
        for (int i = 0; i < UPL; ++i) offs4[i] = (2u * __popc(mb[i] & 3u)) | ((2u * __popc(mb[i] & 15u)) << 8) | ((2u * __popc(mb[i] & 63u)) << 16) | (mb[i] << 24);
        if constexpr (SINGLE) {
            int64_t row_end = values_len;
            if (row_offsets) { if (row + 1 < rows) row_end = row_offsets[row + 1] << SH; }
            else row_end = run + fixed_row_nnz;
            int64_t len = row_end - run;
            len = len < 0 ? 0 : (len > kTile ? kTile : len);
            stage_issue((int)len);
        }
        if constexpr (SINGLE) stage_commit();
        __syncthreads();
        const int t0 = s_tot[0], t1 = s_tot[1], t2 = s_tot[2], t3 = s_tot[3];
        if constexpr (!SINGLE) {
            run += (int64_t)s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
            run = run > values_len ? values_len : run;
            stage_issue(t0 + t1 + t2 + t3);
            stage_commit();
            __syncthreads();
        }
        // the owner lane of unit u = i*256 + tid is lane u/4 = i*64 + tid/4: wave i
        const int wprefix[4] = {0, t0, t0 + t1, t0 + t1 + t2};
        const char* sv = reinterpret_cast<const char*>(s_val);
        uint16_t* orow = out + row * cols + (u0 << 3);
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int u = i * kBlock + tid;
            if (u >= nu) continue;
            const int wb = wprefix[i];
            const uint32_t mv = offs4[i] >> 24;
            const uint32_t r = (uint32_t)s_rank[u] + (uint32_t)(wb + shift);
            const uint32_t a0 = 2u * r;
            uint32_t w[4];
            // selectors from the 8-entry table (8 distinct banks: conflict-free), window offsets from popcounts
            const uint32_t offs[4] = {0u, offs4[i] & 0xffu, (offs4[i] >> 8) & 0xffu, (offs4[i] >> 16) & 0xffu};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t aj = a0 + offs[j];
                const uint32_t a = aj & ~3u;
                const uint32_t lo = *reinterpret_cast<const uint32_t*>(sv + a), hi = *reinterpret_cast<const uint32_t*>(sv + a + 4);
                w[j] = __builtin_amdgcn_perm(hi, lo, s_sel8[((aj & 2u) << 1) | ((mv >> (2 * j)) & 3u)]);
            }
            stream_store16(orow + ((int64_t)u << 3), u32x4{w[0], w[1], w[2], w[3]});
        }
        __syncthreads();  // (skipping this barrier on the last tile measured 1 us SLOWER)
    }
}

template <int FORM, int ES>
__global__ __launch_bounds__(kBlock, FORM == kDecRows ? 6 : 1) void bitmask_decompress16_kernel(const uint16_t* __restrict__ vin, int64_t values_len_e,
                                                                      const uint8_t* __restrict__ bitmask,
                                                                      const int64_t* __restrict__ row_offsets, int64_t fixed_row_nnz_e,
                                                                      int64_t rows, int64_t cols_e, uint16_t* __restrict__ out) {
    __shared__ Decomp16Lds lds;
    bitmask_decompress16_tiles<FORM, ES>(lds, (int64_t)blockIdx.x, (int64_t)gridDim.x, vin, values_len_e, bitmask, row_offsets, fixed_row_nnz_e, rows, cols_e, out);
}

// Round 6: the same kernel over a TABLE of tensors (ct_bitmask_decompress_batch) — loading a sparse checkpoint is 154 (TinyLlama) to thousands of
// decompress launches of 1-23 MB each, ~5 us of host per launch around kernels of 3-10 us.  One grid: a workgroup finds its tensor by a binary
// search over the running tile count and expands one tile of it.
template <int ES>
__global__ __launch_bounds__(kBlock) void bitmask_decompress16_batch_kernel(const ct_bitmask_ditem* __restrict__ items, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ct_bitmask_ditem& it = items[lo];
    const int64_t tile = (int64_t)blockIdx.x - it.first_block;
    const int64_t never = (int64_t)1 << 62;  // one tile per workgroup
    __shared__ Decomp16Lds lds;  // ONE copy for the four forms below
    const uint16_t* vin = static_cast<const uint16_t*>(it.values);
    uint16_t* out = static_cast<uint16_t*>(it.out);
    // `single` (ct_bitmask_decompress_batch_plan): the tile form — kDecRows / kDecSingle / kDecFlat
    if constexpr (ES == 2) {
        if (it.single == kDecFlat) { bitmask_decompress16_tiles<kDecFlat, 2>(lds, tile, never, vin, it.values_len, it.bitmask, it.row_offsets, -1, it.rows, it.cols, out); return; }
    }
    if (it.single == kDecSingle) bitmask_decompress16_tiles<kDecSingle, ES>(lds, tile, never, vin, it.values_len, it.bitmask, it.row_offsets, -1, it.rows, it.cols, out);
    else bitmask_decompress16_tiles<kDecRows, ES>(lds, tile, never, vin, it.values_len, it.bitmask, it.row_offsets, -1, it.rows, it.cols, out);
}

// ------------------------------------------------------------------------- compress, fused flat form (16-bit)
// values = x[x != 0] needs the global exclusive prefix of the non-zero counts.  For 16-bit payloads
// with cols % 8 == 0 the tensor is ONE flat stream of 8-element units (a unit <-> one bitmask byte),
// rows only matter for row_offsets.  Decomposition, chosen so that no workgroup ever waits for
// another one:
//   wave-tile = 256 units (lane = 4 units 64 apart: every wave load instruction reads 1 KiB contiguous)
//   span      = `span` consecutive wave-tiles, processed serially by ONE wave (next tile prefetched)
//   block     = 4 waves = 4 consecutive spans
//   kernel A (flat16_count):   bitmask (dword stores through a wave-private LDS byte transpose),
//                              one count per span and one per block.
//   kernel B (flat16_scatter): block prefix = sum of the earlier blocks' counts (<= 4096 values, one
//                              coalesced pass), wave prefix adds the earlier spans of the block, then
//                              each wave walks its span with a running prefix: DPP wave scans rank the
//                              units, the non-zeros are compacted into a wave-private LDS slab at the
//                              16-byte phase of their destination and leave as aligned 16-byte
//                              streaming stores.  No barrier inside the loops, no scan kernel, no
//                              host round trip, no inter-workgroup hand-off.
// Measured alternatives at 8192^2 bf16 (profiles/): count + single-block scan + scatter
// 87 us; single-kernel decoupled look-back 171 us and count + non-blocking look-back scatter 133 us
// (a cross-CU status poll costs 3-5 us on the consumer under streaming load and the nearest resolved
// prefix is ~1000 tiles back, so every tile paid several dependent polls).  A second single-pass form — tile
// kept in registers, counts published, a TWO-LEVEL prefix (64-tile group sums + in-group counts: two independent
// 64-lane loads, no chain) — was bit-exact but no faster: 327 us with an atomic ticket per tile (4096 same-address
// device-scope atomics serialise at ~40 ns each), 267 us with agent-scope relaxed stores (they sit in the writer's
// XCD-local L2 until evicted), 102 us with system-scope write-through stores and loads; the same kernel WITHOUT
// waiting for the prefix runs in 45 us, so the 8-XCD hand-off alone costs more than the second read it saves
// (~130 KB of tile data per CU can be held in registers; at 22 GB/s per CU that allows a 6 us tile lifetime,
// the publish -> visible -> poll round trip is longer).
constexpr int kWT = 256;  // units per wave-tile

struct Flat16Plan {
    int64_t units, nblocks;
    int span;
};
constexpr int kMaxChunks = 64;
static Flat16Plan flat16_plan(int64_t rows, int64_t cols) {
    Flat16Plan p;
    p.units = rows * (cols / 8);
    const int64_t wts = cdiv64(p.units, kWT);
    int64_t span = cdiv64(wts, 4 * 4096);  // at most 4096 blocks: the block-prefix pass stays one sweep
    p.span = (int)(span < 4 ? 4 : span);  // measured flat between 2 and 8 at 8192^2
    p.nblocks = cdiv64(wts, 4 * (int64_t)p.span);
    if (p.nblocks < 1) p.nblocks = 1;
    return p;
}

// ---- the compaction primitives shared by the scatter kernels (both were VALU-bound: ~600 vector instructions per wave-tile)
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {  // byte offset of a __shared__ object inside the workgroup's LDS
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// the 8 non-zero flags of a unit, 14 VALU: (x & keep) -> v_pk_min_u16(.., 1) gives 0/1 per half; the four dwords are merged with
// v_lshl_or and the high halves folded down.  keep = 0x7fff7fff for floats (-0.0 is a zero), 0xffffffff for integers
__device__ __forceinline__ uint32_t nz_mask16_fast(const u32x4& r, uint32_t keep) {
    const uint32_t ws[4] = {r.x & keep, r.y & keep, r.z & keep, r.w & keep};
    const uint32_t one = 0x00010001u;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t f;  // (written as a vector min, hipcc turns it into two compares, two selects and a v_perm)
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(ws[j]), "v"(one));
        acc |= f << (2 * j);
    }
    return ((acc >> 15) & 0xaau) | (acc & 0x55u);
}

// 32-bit payloads seen as pairs of 16-bit halves: the unit's four non-zero flags, each doubled (bits 2j and 2j + 1 = element j), so that
// ranks, compaction and stores below move both halves of a kept element.  keep = 0x7fffffff for floats, 0xffffffff for integers
__device__ __forceinline__ uint32_t nz_mask32_pairs(const u32x4& r, uint32_t keep) {
    return ((r.x & keep) ? 0x03u : 0u) | ((r.y & keep) ? 0x0cu : 0u) | ((r.z & keep) ? 0x30u : 0u) | ((r.w & keep) ? 0xc0u : 0u);
}
// 8-bit payloads (round 6): one flag per BYTE of the unit, byte j of dword k -> bit 4 k + j (only counted: the row form makes its own masks)
__device__ __forceinline__ uint32_t nz_mask8(const u32x4& r) {
    const uint32_t ws[4] = {r.x, r.y, r.z, r.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = ws[k];
        const uint32_t nzb = ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w;  // bit 7 of each byte: the byte is non-zero
        m |= (((nzb >> 7) & 1u) | ((nzb >> 14) & 2u) | ((nzb >> 21) & 4u) | ((nzb >> 28) & 8u)) << (4 * k);
    }
    return m;
}
template <int ES>
__device__ __forceinline__ uint32_t nz_mask_unit(const u32x4& r, uint32_t keep) {
    if constexpr (ES == 4) return nz_mask32_pairs(r, keep);
    else if constexpr (ES == 1) return nz_mask8(r);
    else return nz_mask16_fast(r, keep);
}
// a doubled mask byte -> the element nibble (bits 0, 2, 4, 6)
__device__ __forceinline__ uint32_t pair_mask_nibble(uint32_t b) {
    return (b & 1u) | ((b >> 1) & 2u) | ((b >> 2) & 4u) | ((b >> 3) & 8u);
}

// ranks of a wave-tile's 4 x 64 units in unit order i * 64 + lane, from their non-zero counts (<= 8 each): two DPP wave scans over
// packed pairs of 16-bit counts, interleaved (four separate scans were 20 DPP adds + ~50 hazard nops).  Returns the tile's total.
__device__ __forceinline__ int tile_ranks(const uint32_t (&mm)[4], uint32_t (&rank)[4]) {
    const int c0 = __popc(mm[0]), c1 = __popc(mm[1]), c2 = __popc(mm[2]), c3 = __popc(mm[3]);
    int v[2] = {c0 | (c1 << 16), c2 | (c3 << 16)};
#define CT_SCAN_STEP(ctrl, rmask)                                                                    \
    {                                                                                                \
        const int a0 = __builtin_amdgcn_update_dpp(0, v[0], ctrl, rmask, 0xf, false);                \
        const int a1 = __builtin_amdgcn_update_dpp(0, v[1], ctrl, rmask, 0xf, false);                \
        v[0] += a0;                                                                                  \
        v[1] += a1;                                                                                  \
    }
    CT_SCAN_STEP(0x111, 0xf)  // row_shr:1
    CT_SCAN_STEP(0x112, 0xf)  // row_shr:2
    CT_SCAN_STEP(0x114, 0xf)  // row_shr:4
    CT_SCAN_STEP(0x118, 0xf)  // row_shr:8
    CT_SCAN_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
    CT_SCAN_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2, 3
#undef CT_SCAN_STEP
    const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane(v[0], 63), t23 = (uint32_t)__builtin_amdgcn_readlane(v[1], 63);
    const uint32_t T0 = t01 & 0xffffu, T1 = t01 >> 16, T2 = t23 & 0xffffu, T3 = t23 >> 16;
    rank[0] = ((uint32_t)v[0] & 0xffffu) - (uint32_t)c0;
    rank[1] = ((uint32_t)v[0] >> 16) - (uint32_t)c1 + T0;
    rank[2] = ((uint32_t)v[1] & 0xffffu) - (uint32_t)c2 + T0 + T1;
    rank[3] = ((uint32_t)v[1] >> 16) - (uint32_t)c3 + T0 + T1 + T2;
    return (int)(T0 + T1 + T2 + T3);
}

// LDS table, 256 x 8 x uint16: entry [m][k] = 2 * (number of kept elements of the unit before element k) when bit k of m is set,
// 0x2000 otherwise — an offset that pushes the address past every slab so that min(address, dump slot) selects the dump slot:
// an element costs one add + one min + the ds_write_b16 (was: bit test, two selects, add, shift)
constexpr int kLutBytes = 256 * 16;
__device__ __forceinline__ void build_compact_lut(uint16_t* lut, int tid) {
    if (tid < 256) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k0 = 2 * j, k1 = 2 * j + 1;
            const uint32_t e0 = ((tid >> k0) & 1) ? 2u * (uint32_t)__popc(tid & ((1 << k0) - 1)) : 0x2000u;
            const uint32_t e1 = ((tid >> k1) & 1) ? 2u * (uint32_t)__popc(tid & ((1 << k1) - 1)) : 0x2000u;
            w[j] = e0 | (e1 << 16);
        }
        reinterpret_cast<u32x4*>(lut)[tid] = u32x4{w[0], w[1], w[2], w[3]};
    }
}

// the kept elements of a wave-tile's four unit rows -> the wave's slab, at halfword position shift + rank[i] + (rank inside the unit)
__device__ __forceinline__ void compact_into_slab(const u32x4 (&cur)[4], const uint32_t (&mm)[4], const uint32_t (&rank)[4], int shift,
                                                  uint32_t slab_a, uint32_t dump_a, uint32_t lut_a) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 t = *(lds_cu32x4_t*)(uintptr_t)(lut_a + mm[i] * 16u);
        const uint32_t ts[4] = {t.x, t.y, t.z, t.w};
        const uint32_t ws[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
        const uint32_t ab = slab_a + 2u * ((uint32_t)shift + rank[i]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t off = (k & 1) ? (ts[k >> 1] >> 16) : (ts[k >> 1] & 0xffffu);
            const uint32_t a = min(ab + off, dump_a);
            *(lds_u16_t*)(uintptr_t)a = (uint16_t)((k & 1) ? (ws[k >> 1] >> 16) : ws[k >> 1]);
        }
    }
}

__device__ __forceinline__ void load_wt(const u32x4* __restrict__ x, int64_t units, int64_t wt, int lane, u32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t u = wt * kWT + i * 64 + lane;
        r[i] = u < units ? x[u] : u32x4{0u, 0u, 0u, 0u};
    }
}

__global__ __launch_bounds__(kBlock) void flat16_count_kernel(const u32x4* __restrict__ x, bool is_float, int64_t units, int span,
                                                              uint8_t* __restrict__ bitmask, int mask_dwords, int64_t* __restrict__ block_tot,
                                                              int32_t* __restrict__ span_tot) {
    __shared__ __attribute__((aligned(16))) uint8_t s_m[kBlock / 64][kWT];
    __shared__ int s_cnt[kBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t wt0 = ((int64_t)blockIdx.x * 4 + wave) * span;
    const uint32_t keepbits = is_float ? 0x7fff7fffu : 0xffffffffu;
    int cnt = 0;
    // four wave-tiles (16 KB, 16 loads per lane) requested before the first is used: the kernel only reads, and with "current +
    // next" a wave had 8 KB in flight (28-29 us at 8192^2 against 24 us for the observer's read of the same bytes)
    for (int j0 = 0; j0 < span; j0 += 4) {
        if ((wt0 + j0) * kWT >= units) break;  // wave-uniform
        u32x4 t[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j0 + q < span) load_wt(x, units, wt0 + j0 + q, lane, t[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t wt = wt0 + j0 + q;
            if (j0 + q >= span || wt * kWT >= units) break;  // wave-uniform
            uint32_t mm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mm[i] = nz_mask16_fast(t[q][i], keepbits);
                cnt += __popc(mm[i]);
            }
            if (mask_dwords) {
                // same-wave LDS operations execute in order: no barrier
#pragma unroll
                for (int i = 0; i < 4; ++i) s_m[wave][i * 64 + lane] = (uint8_t)mm[i];
                const uint32_t d = reinterpret_cast<const uint32_t*>(s_m[wave])[lane];
                const int64_t u = wt * kWT + 4 * lane;
                if (u < units) *reinterpret_cast<uint32_t*>(bitmask + u) = d;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t u = wt * kWT + i * 64 + lane;
                    if (u < units) bitmask[u] = (uint8_t)mm[i];
                }
            }
        }
    }
    cnt = __builtin_amdgcn_readlane(wave_incl_scan(cnt), 63);
    if (lane == 0) {
        span_tot[(int64_t)blockIdx.x * 4 + wave] = cnt;
        s_cnt[wave] = cnt;
    }
    __syncthreads();
    if (tid == 0) block_tot[blockIdx.x] = (int64_t)s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// `x`, `units`: this launch's chunk of the flat unit stream, which starts at unit `u0` of the tensor; `base`: device word holding the
// number of non-zeros before the chunk (NULL: 0); `total_out` receives base + the chunk's count
__global__ __launch_bounds__(kBlock) void flat16_scatter_kernel(const u32x4* __restrict__ x, bool is_float, int64_t units, int span, int64_t upr,
                                                                int64_t rows, uint16_t* __restrict__ vout, int64_t capacity,
                                                                int64_t* __restrict__ row_offsets, int64_t* __restrict__ total_out,
                                                                const int64_t* __restrict__ block_tot, const int32_t* __restrict__ span_tot,
                                                                int64_t u0, const int64_t* __restrict__ base) {
    constexpr int kSlabData = kWT * 8 + 8;      // compacted run (+ phase shift)
    constexpr int kSlab = kSlabData + 64;       // + one dump slot per lane
    __shared__ __attribute__((aligned(16))) uint16_t s_val[kBlock / 64][kSlab];
    __shared__ long long s_part[kBlock / 64];
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[256 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    build_compact_lut(s_lut, tid);
    // the first wave-tile is requested BEFORE the prefix sweep (up to 16 dependent loads per thread + a barrier: 1-3 us during which the
    // block used to have nothing in flight)
    uint16_t* slab = s_val[wave];
    const int64_t wt0 = ((int64_t)blockIdx.x * 4 + wave) * span;
    u32x4 cur[4], nxt[4];
    load_wt(x, units, wt0, lane, cur);
    // exclusive prefix of this block, then of this wave's span
    int64_t part = 0;
    for (int64_t b = tid; b < (int64_t)blockIdx.x; b += kBlock) part += block_tot[b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
    if (lane == 0) s_part[wave] = part;
    __syncthreads();
    int64_t run = (int64_t)s_part[0] + s_part[1] + s_part[2] + s_part[3] + (base ? *base : 0);
    for (int w = 0; w < wave; ++w) run += span_tot[(int64_t)blockIdx.x * 4 + w];

    const uint32_t keepbits = is_float ? 0x7fff7fffu : 0xffffffffu;
    const uint32_t slab_a = lds_addr(slab), dump_a = slab_a + 2u * (uint32_t)(kSlabData + lane), lut_a = lds_addr(s_lut);
    // the next row that starts at or after this wave's first unit (one 64-bit division per wave, not per wave-tile)
    int64_t next_r = (u0 + wt0 * kWT + upr - 1) / upr, next_u = next_r * upr;
    bool last_here = false;
    for (int j = 0; j < span; ++j) {
        const int64_t wt = wt0 + j;
        if (wt * kWT >= units) break;  // wave-uniform
        if (j + 1 < span) load_wt(x, units, wt + 1, lane, nxt);
        last_here = (wt + 1) * kWT >= units;
        // ranks in unit order i*64 + lane
        uint32_t mm[4], rank[4];
        int total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) mm[i] = nz_mask16_fast(cur[i], keepbits);
        total = tile_ranks(mm, rank);
        // row offsets of the rows that start inside this wave-tile (wave-uniform loop, usually 0-1 trips)
        {
            const int64_t ubeg = u0 + wt * kWT, uend = ubeg + kWT;  // tensor-wide unit numbers
            for (; next_r < rows && next_u < uend; ++next_r, next_u += upr) {
                const int q = (int)(next_u - ubeg);
                const int i = q >> 6, l = q & 63;
                const uint32_t rk = i == 0 ? rank[0] : (i == 1 ? rank[1] : (i == 2 ? rank[2] : rank[3]));
                if (lane == l) row_offsets[next_r] = run + rk;
            }
        }
        // compact into the wave's slab at the 16-byte phase of the destination
        // branch-free: every element is written, the zeros go to a per-lane dump slot (predicated
        // stores compile to an exec-mask save / branch / restore per element: measured slower)
        const int shift = (int)(run & 7);
        compact_into_slab(cur, mm, rank, shift, slab_a, dump_a, lut_a);
        // slab[shift, shift + total) -> vout[run, run + total): aligned 16-byte body, scalar head / tail
        const int64_t end = run + total;
        const int64_t e0 = run - shift;
        const int64_t body_lo = (run + 7) & ~(int64_t)7, body_hi = end & ~(int64_t)7;
        if (body_hi > body_lo) {
            const int nvec = (int)((body_hi - body_lo) >> 3);
            const int v0 = (int)((body_lo - e0) >> 3);
            for (int v = lane; v < nvec; v += 64) {
                const int64_t gi = body_lo + ((int64_t)v << 3);
                if (gi + 8 <= capacity) stream_store16(vout + gi, reinterpret_cast<const u32x4*>(slab)[v0 + v]);
                else for (int t = 0; t < 8; ++t) if (gi + t < capacity) vout[gi + t] = slab[gi + t - e0];
            }
            for (int64_t gi = run + lane; gi < body_lo; gi += 64) if (gi < capacity) vout[gi] = slab[gi - e0];
            for (int64_t gi = body_hi + lane; gi < end; gi += 64) if (gi < capacity) vout[gi] = slab[gi - e0];
        } else {
            for (int64_t gi = run + lane; gi < end; gi += 64) if (gi < capacity) vout[gi] = slab[gi - e0];
        }
        run = end;
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
    }
    if (last_here && lane == 0 && total_out) *total_out = run;
}

// ------------------------------------------------------------------------- hand-off words
// History of the single-pass attempts (all bit-exact, all removed; DESIGN.md 5.4 has the numbers).  A tile that waits for its
// prefix while holding its data in registers — decoupled look-back, two-level and three-level prefix words, XCD-local variants —
// never beat the two kernels at 8192^2 (65-330 us against 70): a system-scope store becomes visible to a system-scope load on
// another XCD after 3-5 us, independent of the polling rate, and thousands of tiles chained two or three such hops each.  The
// resident form below needs ONE hop for the whole tensor and does all the expensive work before it.
//
// Hand-off words travel at system scope: the store writes through to memory, the loads bypass the per-XCD L2s (agent scope is
// compiled to the same sc1 accesses on this multi-XCD part and measured identical).  Every word is 64 bits, (generation << 32) |
// value, written by ONE store: the generation is unique per launch (random start per process), so the workspace needs no clearing
// and stale words never look valid.
__device__ __forceinline__ void op_store(unsigned long long* slot, unsigned long long v) {
    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long op_load(const unsigned long long* slot) {
    return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------- compress, RESIDENT (16-bit), round 2
// The one-pass forms above lose to the hand-off because a tile can only live in registers for a few microseconds while thousands
// of tiles depend on one another.  This form turns the problem around: a workgroup's whole share of the tensor lives in its
// registers across ONE hand-off, and everything expensive happens BEFORE the hand-off, while the loads are still landing.
// Workgroup = 8 waves; a wave owns KEEP = 8 consecutive wave-tiles (32 KB, 128 VGPRs), a workgroup 256 KB:
//   phase A  all 32 loads of a wave are issued up front (256 KB in flight per workgroup: the memory-level parallelism comes from
//            the loads, not from occupancy).  As a tile lands: masks (-> an LDS plane, for the bitmask), ranks, compaction through the
//            wave's LDS slab at the tile's own origin, and the compacted tile is read back INTO THE SAME registers; its count stays
//            in a scalar.  The row offsets of rows that start in the tile are left in row_offsets[] relative to the tile.
//   hand-off the workgroup's count goes out as ONE generation-tagged 64-bit system-scope word; workgroup b waits for the words of
//            the b workgroups before it (lane t polls word t, t + 512, ...: every hop is direct, nothing is chained).  The bitmask
//            leaves while the counts travel.
//   phase B  stores only: each compacted tile goes to vout + (its offset) as 16-byte stores at 2-byte alignment (global memory
//            takes them), its last partial vector element-wise; the stashed row offsets get the tile's offset added.
// x is read once.  Workgroups beyond the chip's residency start as earlier ones retire (their reads overlap the others' stores).
// A count word that does not arrive within the time budget (2 ms) is not an error: the waiting workgroup counts that workgroup's
// share of x itself, so the kernel cannot fail or deadlock whatever the dispatch order or residency, and the wave that owns the
// last wave-tile writes *total directly (an earlier form reported failures through a fail word and needed a one-thread kernel
// after it to publish *total: 2.5-4 us per call; a generation-tagged CAS completion counter cost 580 us).
// Round 3 (per-workgroup time stamps, CT_BITMASK_RESIDENT=3, 8192^2): round 1's workgroups publish at 8 us (blocks 0-255) / 12 us
// (256-511) — the load phase, memory-bound —, have their prefix at 14.7 / 18.6 us (every workgroup waits for the SLOWEST earlier loader
// plus a 3-5 us hop: its polls queue behind the co-resident workgroup's streaming loads) and are done at 17 / 26 us; round 2 repeats that
// from 17-29 us to 49 us.  Tried against it: (1) publishing the count before the compaction (kept: the compaction now overlaps the wait;
// no change in time — the wait is set by the slowest loader, not by this workgroup's own compute); (2) a persistent STREAMING form —
// 64 / 32 KB batches handed out by an atomic ticket counter, two batches in registers, p1 (count + publish) / store previous / load
// next / p2 (compact) — bit-exact, but 69-87 us: every batch's prefix depends on every earlier batch, so each batch step is a chip-wide
// synchronisation (ticket + load + hop, three dependent memory round trips of 2-3 us under load for 64 KB of work); (3) delaying the
// start of blocks [CUs, 2 CUs) by one load phase so that a CU's two workgroups alternate load / store phases: 48.2 -> 44.6 us, kept for
// launches of at least two residency rounds; (4) other workgroup shapes without the stagger — 4 waves x 4 tiles (four per CU) 47.9 us,
// 8 x 2 (three per CU) 46.8, 8 x 3 50.5, 4 x 2 (six per CU) 47.9 against 48.2 for 8 x 4: the shape is not the lever either.
// Diagnostics (-DCT_DIAG: libct_hip_diag.so, built next to the product library and loaded only by the tests / dev tools that force the
// alternative forms): per-workgroup time stamps in the kernel's argument list and the CT_BITMASK_RESIDENT* environment knobs of the
// launcher.  The shipped library has neither.
#ifdef CT_DIAG
#define CT_STAMPS_PARAM unsigned long long* __restrict__ stamps,
#define CT_STAMPS_ARG(x) x,
#define CT_STAMP(k) if (stamps && tid == 0 && b < kResStampWGs) stamps[b * 4 + k] = wall_clock64()
#define CT_STAMP_T0(k) if (stamps && b < kResStampWGs) stamps[b * 4 + k] = wall_clock64()
#else
#define CT_STAMPS_PARAM
#define CT_STAMPS_ARG(x)
#define CT_STAMP(k) (void)0
#define CT_STAMP_T0(k) (void)0
#endif
typedef u32x4 u32x4_a2_t __attribute__((aligned(2)));
typedef u32x4 u32x4_a1_t __attribute__((aligned(1)));
typedef u32x2 u32x2_a1_t __attribute__((aligned(1)));
constexpr int kResKeep = 4;       // wave-tiles a wave keeps in registers (16 KB, 64 VGPRs)
constexpr int kResWaves = 8;      // waves per workgroup: ~106 VGPRs -> 4 waves per SIMD = two workgroups per CU
constexpr int kResMaxWGs = 8192;  // count words in the workspace: 1 GiB of 16-bit elements per launch, more goes in chunks
constexpr int kResStampWGs = 1024; // CT_BITMASK_RESIDENT=3: time stamps of the first workgroups (two residency rounds at 8192^2)
constexpr int kResRoundWords = 64; // "inclusive count through residency round r" words (kResMaxWGs / (2 x CUs) rounds at most)

// ES = 4 (round 3): 32-bit payloads ride the same kernel as pairs of halves — `units`, `upr`, `capacity`, the count words and every
// offset inside the kernel are in 16-byte units / 16-bit halves exactly as for ES = 2; only the non-zero test (per 32-bit element, flag
// doubled), the bitmask that leaves (one bit per ELEMENT: the nibbles of two adjacent units make a byte), the row offsets and the totals
// (halved on the way out) differ.
// (round 6: the body is a device function of the workgroup's index `b` inside ITS tensor and that tensor's workgroup count `nwg` — the single-tensor
// kernel below hands it blockIdx.x / gridDim.x, the batched kernel (ct_bitmask_compress_batch) a table row's)
template <int KEEP, int WAVES, int ES, int ROWB = 0>
__device__ __forceinline__ void flat16_resident_body(const int b, const int nwg, const u32x4* __restrict__ x, bool is_float, int64_t units, int64_t upr, int64_t rows, int tpw,
                                                     uint16_t* __restrict__ vout, int64_t capacity, uint8_t* __restrict__ bitmask,
                                                     int mask_dwords, int64_t* __restrict__ row_offsets, int64_t u0,
                                                     const unsigned long long* __restrict__ base, unsigned long long* __restrict__ slots,
                                                     unsigned long long* __restrict__ run_out, uint32_t gen, unsigned long long wait_ticks,
                                                     CT_STAMPS_PARAM int stagger_lo, int stagger_hi, unsigned stagger_ticks,
                                                     unsigned stagger_slope_q8, int round_wgs, unsigned long long* __restrict__ round_words) {
    constexpr int kSlabData = kWT * 8 + 8;      // compacted run of one wave-tile
    constexpr int kSlab = kSlabData + 64;       // + one dump slot per lane
    __shared__ __attribute__((aligned(16))) uint16_t s_val[WAVES][kSlab];
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[ROWB ? 8 : 256 * 8];
    __shared__ __attribute__((aligned(16))) uint32_t s_mask[ROWB ? 1 : WAVES][ROWB ? 1 : KEEP][ROWB ? 4 : 64];  // packed masks (unit i * 64 + lane in byte i)
    __shared__ int s_cnt[WAVES];
    __shared__ long long s_part[WAVES];
    __shared__ int s_miss[WAVES * 64];
    __shared__ int s_nmiss;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t wt0 = ((int64_t)b * WAVES + wave) * tpw;  // tpw <= KEEP wave-tiles per wave
    const uint32_t keepbits = is_float ? (ES == 4 ? 0x7fffffffu : 0x7fff7fffu) : 0xffffffffu;
    constexpr int SH = ES == 4 ? 1 : 0;  // halves per element, as a shift
    // round 6, ES = 1 (the row form only): everything downstream of the compaction counts in GRANULES — 16-bit halves for 16- / 32-bit payloads,
    // BYTES for 8-bit ones (`capacity`, the count words, the running totals, the offsets into `vout`); a 16-byte vector holds GPV of them
    static_assert(ES != 1 || ROWB, "8-bit payloads take the row form");
    constexpr int GRB = ES == 1 ? 1 : 2, GPV = 16 / GRB;
    if (tid == 0) s_nmiss = 0;
    // ---- stagger: the workgroups of the first residency round start their loads spread over one load phase (workgroup b waits
    // stagger_ticks + (b - stagger_lo) x slope), so that from then on a CU's two workgroups — and the chip as a whole — read, compute /
    // wait for a prefix and store at different times (without it the whole chip moves in lockstep: everybody loads, then everybody
    // stores — 2 x 25 us at 8192^2 for 36 us of traffic).  Later rounds inherit the spread: a workgroup starts when an earlier one retires.
    if (b >= stagger_lo && b < stagger_hi) {
        const unsigned long long s0 = wall_clock64();
        const unsigned long long wait = stagger_ticks + (((unsigned long long)(b - stagger_lo) * stagger_slope_q8) >> 8);
        while (wall_clock64() - s0 < wait) __builtin_amdgcn_s_sleep(16);
    }
    CT_STAMP(0);
    // ---- phase A.  Round 4: every load is UNCONDITIONAL straight-line code (unit index clamped to the chunk's last unit; a unit
    // beyond the chunk, or a tile beyond this wave's `tpw`, contributes no flags), so that hipcc's wait-count pass can count the
    // loads: a tile is consumed behind `s_waitcnt vmcnt(12 / 8 / 4 / 0)` as it lands.  With a load behind a branch the pass gives up
    // and waits for ALL sixteen before the first flag is computed (every s_waitcnt in the round-3 build was vmcnt(0)): the whole
    // load phase stalled the wave, and all four tiles' ranks + compaction (4.2 us, VALU-bound) then sat between the last load and
    // the first poll.  No vector-memory store may sit between the loads and the last tile's use for the same reason: the row
    // offsets of rows that start inside a tile are written after the hand-off word has left (ranks recomputed from the flags).
    u32x4 keep[KEEP][4];
    const uint32_t units32 = (uint32_t)units;  // a chunk holds at most kResMaxWGs workgroups x 8 waves x KEEP tiles x 256 units < 2^31
    const uint32_t ubase = (uint32_t)(wt0 * kWT) + (uint32_t)lane;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t u = ubase + (uint32_t)(i * kWT + q * 64);
            keep[i][q] = x[u < units32 ? u : units32 - 1u];
        }
    }
    if constexpr (!ROWB) {
        build_compact_lut(s_lut, tid);
        __syncthreads();
    }
    const uint32_t slab_a = lds_addr(s_val[wave]), dump_a = slab_a + 2u * (uint32_t)(kSlabData + lane), lut_a = lds_addr(s_lut);
    const u32x4* slab_v = reinterpret_cast<const u32x4*>(s_val[wave]);
    int tot[KEEP];  // wave-uniform
    int64_t next_r = (u0 + wt0 * kWT + upr - 1) / upr, next_u = next_r * upr;  // the next row that starts at or after the wave's first unit
    const int64_t first_r = next_r;
    uint32_t pks[KEEP];
    int cnt = 0;
    // ---- ROWB (round 5; shipped for 32-bit payloads): compaction by ROWS OF 64 ELEMENTS instead of by 16-byte units.  A tile is staged into the
    // wave's slab as it lands (4 x ds_write_b128) and read back TRANSPOSED, one element per lane (64 consecutive elements per instruction,
    // conflict-free); the non-zero test of a row is one compare whose result IS the row's 64 bitmask bits (a ballot in an SGPR pair), a
    // lane's rank is mbcnt of that mask plus a scalar running count, and the survivors of the row go back into the slab (in place: they
    // land at or below the row's own start) as ONE exec-masked store to CONSECUTIVE addresses.  No selector table, no dump slots, no DPP
    // scan, no LDS bank conflicts; lane r keeps the flags of row r, so the bitmask leaves as 8 contiguous bytes per lane.
    // Measured at 8192^2, 50 % zeros (profiles/r05_bitmask_rowform.jsonl): float32 78.6 -> 74.4 us (65.4 -> 69.1 % of the HBM peak) — 16 rows
    // per tile; 16-bit payloads (32 rows per tile) 41.6 -> 44.1 us, so they keep the unit form: per-workgroup stamps put the whole
    // difference into start -> published (7.2 -> 9.7 us).  The row form spends ~8 SCALAR instructions per row (ballot popcount, running
    // count, row base, exec install / restore) and a CU has ONE scalar unit for its 16 resident waves: 32 rows x 4 tiles x 16 waves x 8 =
    // 16 K issue cycles per residency round against 6 per row of vector work on four SIMDs.  Hand-written v_cmp_class_f16 + s_mov exec
    // (6 VALU + 8 SALU per row instead of the compiler's and / cmp / s_and_saveexec / s_cbranch / s_or) and the last tile's count ahead of
    // its compaction measured 44.1 / 46.6 us.  The unit form's LDS bank conflicts (52 % of its LDS cycles) are real but not its limit.
    constexpr int EPU = 16 / ES;            // elements per 16-byte unit
    constexpr int RROWS = kWT * EPU / 64;   // rows of 64 elements per wave-tile: 32 (16-bit) / 16 (32-bit)
    constexpr int UPROW = 64 / EPU;         // units per row
    uint32_t mlo[KEEP], mhi[KEEP];          // lane r: the flags of row r of tile i
    if constexpr (ROWB) {
        typedef typename ElemT<ES>::type elem_t;
        typedef __attribute__((address_space(3))) elem_t lds_elem_t;
        u32x4* slab_w = reinterpret_cast<u32x4*>(s_val[wave]);
        const uint32_t slab_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)slab_a);  // wave-uniform: the row addresses are SGPR + lane offset
        const uint32_t lane_e = (uint32_t)lane * (uint32_t)ES;
        const uint32_t ekeep = ES == 4 ? keepbits : (ES == 2 ? (keepbits & 0xffffu) : 0xffu);
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const uint32_t t_u0 = (uint32_t)(wt0 * kWT) + (uint32_t)(i * kWT);  // the tile's first unit inside the chunk
            const bool any = i < tpw && t_u0 < units32;                         // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) slab_w[q * 64 + lane] = keep[i][q];
            if (any && t_u0 + (uint32_t)kWT > units32) {  // the chunk's last, partial tile (wave-uniform, rare): units beyond the end are staged as zeros
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t_u0 + (uint32_t)(q * 64 + lane) >= units32) slab_w[q * 64 + lane] = u32x4{0u, 0u, 0u, 0u};
            }
            uint32_t lo = 0, hi = 0;
            int run = 0;
            if (any) {
                // (8-bit payloads: 64 rows per tile — read and compacted 32 rows at a time, so that the rows in flight stay 32 registers; the
                // survivors of rows [0, 32) land at or below row 32's start, never on a row that has not been read yet)
                constexpr int RCH = RROWS > 32 ? 32 : RROWS;
#pragma unroll
                for (int c0 = 0; c0 < RROWS; c0 += RCH) {
                    elem_t v[RCH];
#pragma unroll
                    for (int r = 0; r < RCH; ++r) v[r] = *(lds_elem_t*)(uintptr_t)(slab_s + (uint32_t)((c0 + r) * 64 * ES) + lane_e);  // every read of the chunk precedes its first in-place write
#pragma unroll
                    for (int r = 0; r < RCH; ++r) {
                        const uint32_t row_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(slab_s + (uint32_t)run * (uint32_t)ES));  // stays scalar
                        const bool nzr = ((uint32_t)v[r] & ekeep) != 0u;
                        const unsigned long long m = __ballot(nzr);
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (nzr) *(lds_elem_t*)(uintptr_t)(rank * (uint32_t)ES + row_base) = v[r];
                        run += __popcll(m);
                        // lane r keeps row r's flags (this clang has no writelane builtin; the ballot lives in an SGPR pair)
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((uint32_t)m), "n"(c0 + r));
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((uint32_t)(m >> 32)), "n"(c0 + r));
                    }
                }
            }
            mlo[i] = lo;
            mhi[i] = hi;
            tot[i] = run << SH;  // in granules (16-bit halves; bytes for 8-bit payloads), as everything downstream counts
            cnt += tot[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) keep[i][q] = slab_v[q * 64 + lane];  // vector q * 64 + lane of the compacted tile (garbage past tot)
        }
    }
    // ranks, compaction through the wave's slab, read back in place — of ONE wave-tile (LDS and VALU only)
    auto pass2 = [&](int i) {
        uint32_t mm[4], rank[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) mm[q] = (pks[i] >> (8 * q)) & 0xffu;
        tot[i] = tile_ranks(mm, rank);
        compact_into_slab(keep[i], mm, rank, 0, slab_a, dump_a, lut_a);
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[i][q] = slab_v[q * 64 + lane];  // vector q * 64 + lane of the compacted tile (garbage past tot)
    };
    // (per-workgroup stamps, gpurun_out/r04a: the 6.8 us between a workgroup's publish and its resolved prefix were its OWN pass 2 over
    // all four tiles plus one poll round trip, not the other workgroups' loads.)  The tiles before the last one are ranked and
    // compacted AS THEY LAND, in the shadow of the loads still in flight; only the LAST tile's flags gate the publish and only its
    // pass 2 sits behind it.
    if constexpr (!ROWB) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        uint32_t pk = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool live = i < tpw && ubase + (uint32_t)(i * kWT + q * 64) < units32;
            pk |= (live ? nz_mask_unit<ES>(keep[i][q], keepbits) : 0u) << (8 * q);
        }
        s_mask[wave][i][lane] = pk;
        pks[i] = pk;
        cnt += __popc(pk);
        if (i < KEEP - 1) pass2(i);
    }
    cnt = __builtin_amdgcn_readlane(wave_incl_scan(cnt), 63);
    }
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        int wg = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) wg += s_cnt[w];
        op_store(slots + b, ((unsigned long long)gen << 32) | (uint32_t)wg);
        CT_STAMP_T0(1);
    }
    // ---- the last tile's pass 2, while the word travels.  (Round 4, measured and dropped: requesting the round word and the first sweep
    // of count words HERE, ahead of this pass, so that the ~2.5 us poll round trip runs under it — with the flags masked branch-free so
    // that the pass does not wait for those loads — 46.3-47.0 us against 42.2: the early attempt fails more often than the late one, a
    // failed attempt costs a full extra round trip, and published -> resolved grew from 3.3 to 4.6 us; moving the bitmask stores
    // behind the hand-off changed nothing, 46.3.)
    if constexpr (!ROWB) pass2(KEEP - 1);
    // ---- rows that start inside one of this wave's tiles: their offset relative to the tile, completed in phase B
    if constexpr (ROWB) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int64_t ubeg = u0 + (wt0 + i) * kWT, uend = ubeg + kWT;
            if (i < tpw && next_r < rows && next_u < uend) {  // wave-uniform
                // kept elements before row r of the tile: exclusive scan of the rows' popcounts (lane r holds row r's flags)
                const int rc = lane < RROWS ? __popc(mlo[i]) + __popc(mhi[i]) : 0;
                const int excl = wave_incl_scan(rc) - rc;
                for (; next_r < rows && next_u < uend; ++next_r, next_u += upr) {
                    const int q = (int)(next_u - ubeg);          // unit inside the tile
                    const int r = q / UPROW, bit = (q % UPROW) * EPU;
                    const uint32_t flo = (uint32_t)__builtin_amdgcn_readlane((int)mlo[i], r), fhi = (uint32_t)__builtin_amdgcn_readlane((int)mhi[i], r);
                    const unsigned long long below = (((unsigned long long)fhi << 32) | flo) & ((1ull << bit) - 1ull);
                    const int rk = (__builtin_amdgcn_readlane(excl, r) + __popcll(below)) << SH;  // in halves; phase B adds the tile's offset
                    if (lane == (q & 63)) row_offsets[next_r] = (int64_t)rk;
                }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int64_t ubeg = u0 + (wt0 + i) * kWT, uend = ubeg + kWT;
        if (i < tpw && next_r < rows && next_u < uend) {  // wave-uniform; one row in four tiles at 8192 columns
            uint32_t mm[4], rank[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) mm[q] = (pks[i] >> (8 * q)) & 0xffu;
            (void)tile_ranks(mm, rank);
            for (; next_r < rows && next_u < uend; ++next_r, next_u += upr) {
                const int q = (int)(next_u - ubeg);
                const int qi = q >> 6, l = q & 63;
                const uint32_t rk = qi == 0 ? rank[0] : (qi == 1 ? rank[1] : (qi == 2 ? rank[2] : rank[3]));
                if (lane == l) row_offsets[next_r] = (int64_t)rk;
            }
        }
    }
    }
    // ---- the bitmask leaves while the counts travel.  ROWB: lane r holds the 64 flags of row r = 8 consecutive bitmask bytes
    if constexpr (ROWB) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int64_t wt = wt0 + i;
            if (i < tpw && wt * kWT < units && lane < RROWS) {
                // a mask byte covers 8 elements: one 16-bit unit, or two 32-bit units
                const int64_t ub = wt * kWT + (int64_t)lane * UPROW;  // first unit of this lane's row
                uint8_t* dst = bitmask + (ES == 4 ? (ub >> 1) : (ES == 1 ? ub * 2 : ub));  // a mask byte: two 32-bit units / one 16-bit unit / half an 8-bit unit
                if ((wt + 1) * kWT <= units) {
                    *reinterpret_cast<u32x2_a1_t*>(dst) = u32x2{mlo[i], mhi[i]};
                } else {
                    const unsigned long long mm = ((unsigned long long)mhi[i] << 32) | mlo[i];
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        if (ub + (ES == 4 ? 2 * t : (ES == 1 ? t / 2 : t)) < units) dst[t] = (uint8_t)(mm >> (8 * t));
                }
            }
        }
    } else {
    // output dword of lane L = units 4L .. 4L+3 of the tile = byte (L >> 4) of the packed masks of lanes 4 (L & 15) .. + 3
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
        const int64_t wt = wt0 + i;
        if (ES == 4 && i < tpw && wt * kWT < units) {  // wave-uniform; one bit per 32-bit element: lane L < 32 = units 8L .. 8L + 7 of the tile
            if (lane < 32) {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(s_mask[wave][i]) + 4 * (8 * (lane & 7)) + (lane >> 3);
                uint32_t d = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t) d |= pair_mask_nibble(src[4 * t]) << (4 * t);
                const int64_t u = wt * kWT + 8 * lane;  // first unit of this dword; units and u are even
                uint8_t* dst = bitmask + (u >> 1);
                if (mask_dwords && u + 8 <= units) *reinterpret_cast<uint32_t*>(dst) = d;
                else
                    for (int t = 0; t < 4; ++t)
                        if (u + 2 * t < units) dst[t] = (uint8_t)(d >> (8 * t));
            }
        } else if (i < tpw && wt * kWT < units) {  // wave-uniform
            if (mask_dwords) {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(s_mask[wave][i]) + 16 * (lane & 15) + (lane >> 4);
                const uint32_t d = (uint32_t)src[0] | ((uint32_t)src[4] << 8) | ((uint32_t)src[8] << 16) | ((uint32_t)src[12] << 24);
                const int64_t u = wt * kWT + 4 * lane;
                if (u < units) *reinterpret_cast<uint32_t*>(bitmask + u) = d;
            } else {
                const uint32_t pk = s_mask[wave][i][lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t u = wt * kWT + q * 64 + lane;
                    if (u < units) bitmask[u] = (uint8_t)(pk >> (8 * q));
                }
            }
        }
    }
    }
    // ---- hand-off: lane t watches the words of workgroups t, t + 512, ...  A word that does not arrive within the time budget is
    // not an error: the workgroup COUNTS that workgroup's share of x itself (self-help; never observed outside the forced test).
    // So nothing here can fail or deadlock, whatever the dispatch order or residency, and no status has to be reported.
    {
        long long part = 0;
        const unsigned long long t0 = wall_clock64();
        // a workgroup of residency round r >= 1 (b / round_wgs) starts when a round r - 1 workgroup retires: by then the LAST workgroup of
        // round r - 1 has long published the inclusive count through itself (one word, below), so b polls that word plus the b - r * R
        // words of its own round instead of all b.  If the word does not arrive within the budget: all b raw words, as before.
        int w_lo = 0;
        if (round_wgs > 0 && b >= round_wgs) {
            const int r = b / round_wgs;
            if (tid == 0) {
                s_part[0] = -1;
                for (;;) {
                    const unsigned long long v = op_load(round_words + (r - 1));
                    if ((uint32_t)(v >> 32) == gen) { s_part[0] = (long long)(uint32_t)v; break; }
                    if (wall_clock64() - t0 >= wait_ticks) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            __syncthreads();
            const long long head = s_part[0];
            __syncthreads();
            if (head >= 0) {
                w_lo = r * round_wgs;
                if (tid == 0) part = head;
            }
        }
        for (int w0 = w_lo; w0 < b; w0 += (WAVES * 64)) {  // workgroup-uniform trip count
            const int w = w0 + tid;
            const bool need = w < b;
            bool got = !need;
            uint32_t mine = 0;
            if (__builtin_amdgcn_ballot_w64(need) != 0) {  // wave-uniform
                for (;;) {
                    if (!got) {
                        const unsigned long long v = op_load(slots + w);
                        if ((uint32_t)(v >> 32) == gen) { mine = (uint32_t)v; got = true; }
                    }
                    if (__builtin_amdgcn_ballot_w64(!got) == 0) break;
                    if (wall_clock64() - t0 >= wait_ticks) break;
                    __builtin_amdgcn_s_sleep(4);
                }
            }
            part += mine;
            if (!got) s_miss[atomicAdd(&s_nmiss, 1)] = w;
            __syncthreads();
            const int nmiss = s_nmiss;  // workgroup-uniform; almost always 0
            for (int m = 0; m < nmiss; ++m) {
                const int64_t u_lo = (int64_t)s_miss[m] * WAVES * tpw * kWT;
                int64_t u_hi = u_lo + (int64_t)WAVES * tpw * kWT;
                if (u_hi > units) u_hi = units;
                for (int64_t u = u_lo + tid; u < u_hi; u += WAVES * 64) part += __popc(nz_mask_unit<ES>(x[u], keepbits));
            }
            __syncthreads();
            if (tid == 0) s_nmiss = 0;
            __syncthreads();
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
        if (lane == 0) s_part[wave] = part;
    }
    __syncthreads();
    CT_STAMP(2);
    if (round_wgs > 0 && tid == 0 && (b + 1) % round_wgs == 0 && b + 1 < nwg) {  // the last workgroup of a residency round
        long long incl = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) incl += s_part[w] + s_cnt[w];
        op_store(round_words + b / round_wgs, ((unsigned long long)gen << 32) | (uint32_t)incl);
    }
    {
        int64_t run = base ? (int64_t)*base << SH : 0;  // the running totals between chunks are in elements
#pragma unroll
        for (int w = 0; w < WAVES; ++w) run += s_part[w];
        for (int w = 0; w < wave; ++w) run += s_cnt[w];
        // ---- phase B: stores only
        next_r = first_r;
        next_u = first_r * upr;
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int64_t wt = wt0 + i;
            const int total = tot[i];
            if (i >= tpw) continue;  // wave-uniform
            {   // complete the row offsets stashed in phase A (same lane wrote them)
                const int64_t uend = u0 + (wt + 1) * kWT;
                for (; next_r < rows && next_u < uend; ++next_r, next_u += upr) {
                    const int l = (int)(next_u - (u0 + wt * kWT)) & 63;
                    if (lane == l) row_offsets[next_r] = (row_offsets[next_r] + run) >> SH;
                }
            }
            const int64_t lim = run + total < capacity ? run + total : capacity;  // one past the last granule this tile may write
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t gi = run + (int64_t)(q * 64 + lane) * GPV;
                if (gi + GPV <= lim) {
                    if constexpr (GRB == 1) __builtin_nontemporal_store(keep[i][q], reinterpret_cast<u32x4_a1_t*>(reinterpret_cast<uint8_t*>(vout) + gi));  // 1-byte aligned
                    else __builtin_nontemporal_store(keep[i][q], reinterpret_cast<u32x4_a2_t*>(vout + gi));  // 2-byte aligned
                } else if (gi < lim) {
                    const uint32_t ws[4] = {keep[i][q].x, keep[i][q].y, keep[i][q].z, keep[i][q].w};
                    if constexpr (GRB == 1) {
#pragma unroll
                        for (int t = 0; t < 16; ++t)
                            if (gi + t < lim) reinterpret_cast<uint8_t*>(vout)[gi + t] = (uint8_t)(ws[t >> 2] >> (8 * (t & 3)));
                    } else {
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            if (gi + t < lim) vout[gi + t] = (uint16_t)((t & 1) ? (ws[t >> 1] >> 16) : ws[t >> 1]);
                    }
                }
            }
            run += total;
            if ((wt + 1) * kWT >= units && wt * kWT < units && lane == 0) op_store(run_out, (unsigned long long)(run >> SH));  // the chunk's last wave-tile
        }
    }
    CT_STAMP(3);
}

template <int KEEP, int WAVES, int ES, int ROWB = 0>
__global__ __launch_bounds__(WAVES * 64, 4) void flat16_resident_kernel(const u32x4* __restrict__ x, bool is_float, int64_t units, int64_t upr, int64_t rows, int tpw,
                                                                    uint16_t* __restrict__ vout, int64_t capacity, uint8_t* __restrict__ bitmask,
                                                                    int mask_dwords, int64_t* __restrict__ row_offsets, int64_t u0,
                                                                    const unsigned long long* __restrict__ base, unsigned long long* __restrict__ slots,
                                                                    unsigned long long* __restrict__ run_out, uint32_t gen, unsigned long long wait_ticks,
                                                                    CT_STAMPS_PARAM int stagger_lo, int stagger_hi, unsigned stagger_ticks,
                                                                    unsigned stagger_slope_q8, int round_wgs, unsigned long long* __restrict__ round_words) {
    flat16_resident_body<KEEP, WAVES, ES, ROWB>((int)blockIdx.x, (int)gridDim.x, x, is_float, units, upr, rows, tpw, vout, capacity, bitmask, mask_dwords, row_offsets, u0, base,
                                                slots, run_out, gen, wait_ticks, CT_STAMPS_ARG(stamps) stagger_lo, stagger_hi, stagger_ticks, stagger_slope_q8, round_wgs,
                                                round_words);
}

// ------------------------------------------------------------------------------------------
// Round 6: the same kernel over a TABLE of tensors (ct_bitmask_compress_batch) — a checkpoint's sparse weights in one launch.  One launch of
// this kernel is a latency chain (load -> count -> publish -> poll the earlier workgroups' words -> store): 9 us for a 1 MB tensor, 10 us for
// 8 MB, 17 us for 23 MB — 2-27 % of the HBM rate — and a TinyLlama-shaped checkpoint is 154 such tensors, 1.96 ms launched one by one.  In one
// grid the tensors' chains run side by side (a workgroup only ever waits for workgroups of ITS tensor, which have lower block indices and are
// therefore dispatched before it), the HBM pipe is what limits, and the host issues one launch.  Every workgroup finds its tensor by a binary
// search over the running block count (wave-uniform scalar loads), then runs the body above with the row's fields; no stagger, no per-round
// words (they serve single tensors of several residency rounds).
// ------------------------------------------------------------------------------------------
// (A pointer read from the table is a GENERIC pointer to the compiler: the body's accesses become flat_load / flat_store here, as in the W4 table kernels
// — counted by vmcnt AND lgkmcnt.  An address-space round trip, `__builtin_assume(!is_shared && !is_private)` and a by-value copy of the row ahead of every
// store all left the flat ops in place with this hipcc; the table kernels reach 0.58-0.74 of the HBM peak with them.)
template <int ES>
__global__ __launch_bounds__(kResWaves * 64, 4) void flat16_resident_batch_kernel(const ct_bitmask_item* __restrict__ items, int n, unsigned long long* __restrict__ workspace,
                                                                              unsigned long long wait_ticks) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ct_bitmask_item it = items[lo];  // by value: every field is read here, ahead of any store
    flat16_resident_body<kResKeep, kResWaves, ES, ES == 2 ? 0 : 1>(
        (int)((int64_t)blockIdx.x - it.first_block), it.nwg, static_cast<const u32x4*>(it.x), it.is_float != 0, it.units, it.upr, it.rows, it.tpw,
        static_cast<uint16_t*>(it.values), it.values_capacity * (ES == 4 ? 2 : 1), it.bitmask, it.mask_dwords, it.row_offsets, 0, nullptr,
        workspace + it.slots_offset, reinterpret_cast<unsigned long long*>(it.total), it.gen, wait_ticks, CT_STAMPS_ARG(nullptr) 0, 0, 0u, 0u, 0, nullptr);
}

// a table of byte ranges copied by one launch (ct_copy_batch: the exact-size `values` of a batch leave the worst-case arena): 16 KiB per workgroup
__global__ __launch_bounds__(kBlock) void copy_batch_kernel(const ct_copy_item* __restrict__ items, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ct_copy_item& it = items[lo];
    const int64_t off = ((int64_t)blockIdx.x - it.first_block) * (kBlock * 64);
    const uint8_t* src = static_cast<const uint8_t*>(it.src) + off;
    uint8_t* dst = static_cast<uint8_t*>(it.dst) + off;
    const int64_t left = it.bytes - off < (int64_t)kBlock * 64 ? it.bytes - off : (int64_t)kBlock * 64;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
        u32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t o = ((int64_t)q * kBlock + threadIdx.x) * 16;
            if (o + 16 <= left) v[q] = *reinterpret_cast<const u32x4*>(src + o);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t o = ((int64_t)q * kBlock + threadIdx.x) * 16;
            if (o + 16 <= left) stream_store16(dst + o, v[q]);
        }
        const int64_t tail = left & ~(int64_t)15;
        if (threadIdx.x < (unsigned)(left - tail)) dst[tail + threadIdx.x] = src[tail + threadIdx.x];
    } else {
        for (int64_t o = threadIdx.x; o < left; o += kBlock) dst[o] = src[o];
    }
}

// ------------------------------------------------------------------------- 2:4
// magnitude key: |x| as an orderable integer; NaN sorts largest (as torch.topk does)
template <int ES>
__device__ __forceinline__ uint32_t abs_key(typename ElemT<ES>::type bits, bool is_float) {
    if constexpr (ES == 2) return is_float ? (bits & 0x7fffu) : (uint32_t)((int16_t)bits < 0 ? -(int)(int16_t)bits : (int)(int16_t)bits);
    else if constexpr (ES == 4) return is_float ? (bits & 0x7fffffffu) : (uint32_t)((int32_t)bits < 0 ? 0u - bits : bits);
    else { int v = (int8_t)bits; return (uint32_t)(v < 0 ? -v : v); }
}

// 4 keys -> 4-bit mask with exactly two bits: the two largest, ties to the lower index
__device__ __forceinline__ uint32_t top2_mask(const uint32_t (&k)[4]) {
    int a = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) if (k[j] > k[a]) a = j;
    int b = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j == a) continue;
        if (b < 0 || k[j] > k[b]) b = j;
    }
    return (1u << a) | (1u << b);
}

template <int ES>
__global__ __launch_bounds__(kBlock) void sparse24_compress_kernel(const void* __restrict__ x, bool is_float, int64_t units,
                                                                   void* __restrict__ values, uint8_t* __restrict__ bitmask,
                                                                   uint8_t* __restrict__ bytemask, int vec) {
    typedef typename ElemT<ES>::type T;
    for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < units; u += (int64_t)gridDim.x * kBlock) {
        T e[8];
        load_unit<ES>(x, u << 3, 8, vec, e);
        uint32_t m = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t k4[4] = {abs_key<ES>(e[4 * h], is_float), abs_key<ES>(e[4 * h + 1], is_float),
                                    abs_key<ES>(e[4 * h + 2], is_float), abs_key<ES>(e[4 * h + 3], is_float)};
            m |= top2_mask(k4) << (4 * h);
        }
        if (bitmask) bitmask[u] = (uint8_t)m;
        if (bytemask) {
#pragma unroll
            for (int k = 0; k < 8; ++k) bytemask[(u << 3) + k] = (uint8_t)((m >> k) & 1u);
        }
        if (values) {
            T o[4];
            int pos = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((m >> k) & 1u) o[pos++] = e[k];
            T* v = static_cast<T*>(values) + (u << 2);
            if constexpr (ES == 2) {
                *reinterpret_cast<u32x2*>(v) = u32x2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
            } else if constexpr (ES == 4) {
                *reinterpret_cast<u32x4*>(v) = u32x4{o[0], o[1], o[2], o[3]};
            } else {
                *reinterpret_cast<uint32_t*>(v) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
            }
        }
    }
}

// two adjacent units per lane (8- and 16-bit elements): 16 / 32 contiguous bytes in, ONE 8- / 16-byte streaming store of values and a
// 2-byte store of the two mask bytes, exact grid — the unit kernel above stores 4 / 8 bytes of values and single mask bytes per lane
// through a grid-stride loop (bf16 at 8192^2: see DESIGN 5.3)
// one dword of four int8-viewed elements (8-bit payloads): the quad's top-2 by |x| (ties to the lower index) -> mask nibble and the two kept
// bytes in index order.  SWAR: |x| of the four bytes in 5 ops; keys (|x| << 2 | 3 - index) in two packed 16-bit pairs; the top-2 network runs
// on v_pk_max_u16 / v_pk_min_u16 (the largest = max over the pair-wise maxima, the second = max3(min of the pair-wise maxima, the two
// pair-wise minima)); the kept bytes leave through one v_perm.  ~24 VALU per quad against ~45 for the element-wise form.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void top2_bytes(uint32_t x, uint32_t& nibble, uint32_t& kept16) {
    const uint32_t sgn = (x >> 7) & 0x01010101u;
    const uint32_t ab = (x ^ ((sgn << 8) - sgn)) + sgn;             // |int8| per byte, 0 .. 128: no carry between bytes
    const uint32_t a01 = __builtin_amdgcn_perm(0u, ab, 0x0c010c00u);  // (|x0|, |x1|) as 16-bit lanes
    const uint32_t a23 = __builtin_amdgcn_perm(0u, ab, 0x0c030c02u);
    const us2_t A = __builtin_bit_cast(us2_t, (a01 << 2) | 0x00020003u);  // keys: larger |x| wins, then the lower index
    const us2_t B = __builtin_bit_cast(us2_t, (a23 << 2) | 0x00000001u);
    const us2_t M = __builtin_elementwise_max(A, B), N = __builtin_elementwise_min(A, B);  // pairs (0,2) and (1,3)
    const us2_t Ms = {M.y, M.x}, Ns = {N.y, N.x};
    const us2_t T = __builtin_elementwise_max(M, Ms);                                        // the largest key, in both halves
    const us2_t S = __builtin_elementwise_max(__builtin_elementwise_min(M, Ms), __builtin_elementwise_max(N, Ns));  // the second largest
    const uint32_t i1 = 3u - ((uint32_t)T.x & 3u), i2 = 3u - ((uint32_t)S.x & 3u);
    nibble = (1u << i1) | (1u << i2);
    const uint32_t lo = i1 < i2 ? i1 : i2, hi = i1 < i2 ? i2 : i1;
    kept16 = __builtin_amdgcn_perm(0u, x, lo | (hi << 8) | 0x0c0c0000u);
}

// a quad of 16-bit FLOAT payloads (two dwords): keys (|bits| << 2 | 3 - index) as 32-bit integers, the same selection network on
// v_max_u32 / v_min_u32 / v_max3_u32, the two kept halfwords through one v_perm over the dword pair
__device__ __forceinline__ void top2_halves(uint32_t d0, uint32_t d1, uint32_t& nibble, uint32_t& kept32) {
    const uint32_t k0 = ((d0 & 0x7fffu) << 2) | 3u, k1 = (((d0 >> 16) & 0x7fffu) << 2) | 2u;
    const uint32_t k2 = ((d1 & 0x7fffu) << 2) | 1u, k3 = (((d1 >> 16) & 0x7fffu) << 2);
    const uint32_t m0 = k0 > k2 ? k0 : k2, n0 = k0 > k2 ? k2 : k0;   // pairs (0,2) and (1,3)
    const uint32_t m1 = k1 > k3 ? k1 : k3, n1 = k1 > k3 ? k3 : k1;
    const uint32_t t = m0 > m1 ? m0 : m1, u = m0 > m1 ? m1 : m0;
    uint32_t sec = n0 > n1 ? n0 : n1;
    sec = sec > u ? sec : u;
    const uint32_t i1 = 3u - (t & 3u), i2 = 3u - (sec & 3u);
    nibble = (1u << i1) | (1u << i2);
    const uint32_t lo = i1 < i2 ? i1 : i2, hi = i1 < i2 ? i2 : i1;
    // bytes of the pair (d1:d0): element e sits in bytes 2e, 2e + 1
    const uint32_t sel = (2u * lo) | ((2u * lo + 1u) << 8) | ((2u * hi) << 16) | ((2u * hi + 1u) << 24);
    kept32 = __builtin_amdgcn_perm(d1, d0, sel);
}

template <int ES>
__global__ __launch_bounds__(kBlock) void sparse24_pair_kernel(const void* __restrict__ x, bool is_float, int64_t pairs, void* __restrict__ values,
                                                               uint8_t* __restrict__ bitmask) {
    typedef typename ElemT<ES>::type T;
    const int64_t pr = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (pr >= pairs) return;
    T e[2][8];
    if constexpr (ES == 2) {
        const u32x4 a = static_cast<const u32x4*>(x)[2 * pr], b = static_cast<const u32x4*>(x)[2 * pr + 1];
        const uint32_t ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { e[j >> 2][2 * (j & 3)] = (T)(ws[j] & 0xffffu); e[j >> 2][2 * (j & 3) + 1] = (T)(ws[j] >> 16); }
    } else {
        const u32x4 a = static_cast<const u32x4*>(x)[pr];
        const uint32_t ws[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) e[j >> 1][4 * (j & 1) + k] = (T)(ws[j] >> (8 * k));
    }
    if constexpr (ES == 1) {
        const u32x4 a = static_cast<const u32x4*>(x)[pr];
        const uint32_t ws[4] = {a.x, a.y, a.z, a.w};
        uint32_t m16 = 0, k[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t nb;
            top2_bytes(ws[q], nb, k[q]);
            m16 |= nb << (4 * q);
        }
        __builtin_nontemporal_store((uint16_t)m16, reinterpret_cast<uint16_t*>(bitmask) + pr);
        stream_store8(static_cast<u32x2*>(values) + pr, u32x2{k[0] | (k[1] << 16), k[2] | (k[3] << 16)});
        return;
    }
    if constexpr (ES == 2) {
        if (is_float) {  // kernel-uniform
            const u32x4 a = static_cast<const u32x4*>(x)[2 * pr], b = static_cast<const u32x4*>(x)[2 * pr + 1];
            const uint32_t ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t m16 = 0, k[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t nb;
                top2_halves(ws[2 * q], ws[2 * q + 1], nb, k[q]);
                m16 |= nb << (4 * q);
            }
            __builtin_nontemporal_store((uint16_t)m16, reinterpret_cast<uint16_t*>(bitmask) + pr);
            stream_store16(static_cast<u32x4*>(values) + pr, u32x4{k[0], k[1], k[2], k[3]});
            return;
        }
    }
    uint32_t mm = 0;
    T o[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint32_t m = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t k4[4] = {abs_key<ES>(e[u][4 * h], is_float), abs_key<ES>(e[u][4 * h + 1], is_float),
                                    abs_key<ES>(e[u][4 * h + 2], is_float), abs_key<ES>(e[u][4 * h + 3], is_float)};
            m |= top2_mask(k4) << (4 * h);
        }
        int pos = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((m >> k) & 1u) o[u][pos++] = e[u][k];
        mm |= m << (8 * u);
    }
    __builtin_nontemporal_store((uint16_t)mm, reinterpret_cast<uint16_t*>(bitmask) + pr);
    if constexpr (ES == 2) {
        stream_store16(static_cast<u32x4*>(values) + pr, u32x4{(uint32_t)o[0][0] | ((uint32_t)o[0][1] << 16), (uint32_t)o[0][2] | ((uint32_t)o[0][3] << 16),
                                                                (uint32_t)o[1][0] | ((uint32_t)o[1][1] << 16), (uint32_t)o[1][2] | ((uint32_t)o[1][3] << 16)});
    } else {
        stream_store8(static_cast<u32x2*>(values) + pr, u32x2{(uint32_t)o[0][0] | ((uint32_t)o[0][1] << 8) | ((uint32_t)o[0][2] << 16) | ((uint32_t)o[0][3] << 24),
                                                              (uint32_t)o[1][0] | ((uint32_t)o[1][1] << 8) | ((uint32_t)o[1][2] << 16) | ((uint32_t)o[1][3] << 24)});
    }
}

static bool float_kind(int dt) { return is_float_dt(dt); }

static unsigned grid_1d(int64_t items) {
    int64_t g = cdiv64(items, kBlock);
    int64_t cap = (int64_t)kCUs * 32;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

static unsigned grid_rows_1d(int64_t rows) {
    int64_t g = rows;
    int64_t cap = (int64_t)kCUs * 32;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

#define CT_ES_SWITCH(es, ...)                                 \
    switch (es) {                                             \
        case 1: { constexpr int ES = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int ES = 2; __VA_ARGS__; } break; \
        case 4: { constexpr int ES = 4; __VA_ARGS__; } break; \
    }

}  // namespace ct

using namespace ct;

extern "C" {

int ct_pack_bitmasks(const uint8_t* mask, int64_t rows, int64_t cols, uint8_t* out, ct_stream_t stream) {
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    if (rows == 0 || cols == 0) return CT_OK;
    hipLaunchKernelGGL(pack_bitmasks_kernel, dim3(grid_1d(rows * cdiv64(cols, 8))), dim3(kBlock), 0, as_stream(stream), mask, rows, cols, out);
    CT_LAUNCH_CHECK("ct_pack_bitmasks");
}

int ct_unpack_bitmasks(const uint8_t* packed, int64_t rows, int64_t cols, uint8_t* mask, ct_stream_t stream) {
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    if (rows == 0 || cols == 0) return CT_OK;
    hipLaunchKernelGGL(unpack_bitmasks_kernel, dim3(grid_1d(rows * cdiv64(cols, 8))), dim3(kBlock), 0, as_stream(stream), packed, rows, cols, mask);
    CT_LAUNCH_CHECK("ct_unpack_bitmasks");
}

int ct_bitmask_count(const void* x, int dt, int64_t rows, int64_t cols, uint8_t* bitmask, int64_t* row_counts, ct_stream_t stream) {
    const int es = dt_size(dt);
    CT_REQUIRE(es == 1 || es == 2 || es == 4, "unsupported element dtype %d", dt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    if (rows == 0) return CT_OK;
    const int vec = (cols % 8 == 0) && aligned16(x);
    CT_ES_SWITCH(es, hipLaunchKernelGGL((bitmask_count_kernel<ES>), dim3(grid_rows_1d(rows)), dim3(kBlock), 0, as_stream(stream), x,
                                        float_kind(dt), rows, cols, bitmask, row_counts, vec));
    CT_LAUNCH_CHECK("ct_bitmask_count");
}

int ct_bitmask_row_popcount(const uint8_t* bitmask, int64_t rows, int64_t cols, int64_t* row_counts, ct_stream_t stream) {
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    if (rows == 0) return CT_OK;
    hipLaunchKernelGGL(bitmask_row_popcount_kernel, dim3(grid_rows_1d(rows)), dim3(kBlock), 0, as_stream(stream), bitmask, rows, cols, row_counts);
    CT_LAUNCH_CHECK("ct_bitmask_row_popcount");
}

int ct_exclusive_scan_i64(const int64_t* counts, int64_t n, int64_t* offsets, int64_t* total, ct_stream_t stream) {
    CT_REQUIRE(n >= 0, "negative length");
    hipLaunchKernelGGL(exclusive_scan_i64_kernel, dim3(1), dim3(1024), 0, as_stream(stream), counts, n, offsets, total);
    CT_LAUNCH_CHECK("ct_exclusive_scan_i64");
}

int ct_bitmask_scatter(const void* x, int dt, int64_t rows, int64_t cols, const int64_t* row_offsets, void* values, ct_stream_t stream) {
    const int es = dt_size(dt);
    CT_REQUIRE(es == 1 || es == 2 || es == 4, "unsupported element dtype %d", dt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    if (rows == 0 || cols == 0) return CT_OK;
    const int vec = (cols % 8 == 0) && aligned16(x);
    CT_ES_SWITCH(es, hipLaunchKernelGGL((bitmask_scatter_kernel<ES>), dim3(grid_rows_1d(rows)), dim3(kBlock), 0, as_stream(stream), x,
                                        float_kind(dt), rows, cols, row_offsets, values, vec));
    CT_LAUNCH_CHECK("ct_bitmask_scatter");
}


// the process-wide launch generation of the resident kernels' count words (see ct_bitmask_compress): unique per launch, random start
static std::atomic<uint32_t>& resident_generation() {
    static std::atomic<uint32_t> generation{[]() {
        std::random_device rd;
        return (uint32_t)rd() ^ (uint32_t)std::chrono::steady_clock::now().time_since_epoch().count();
    }()};
    return generation;
}

static int device_cus() {
    int dev = 0, cus = kCUs;
    if (hipGetDevice(&dev) == hipSuccess) {
        static std::atomic<int> cu_cache[64];
        int c = dev >= 0 && dev < 64 ? cu_cache[dev].load() : 0;
        if (c <= 0) {
            if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = kCUs;
            if (dev >= 0 && dev < 64) cu_cache[dev].store(c);
        }
        cus = c;
    }
    return cus;
}

int64_t ct_bitmask_compress_workspace_bytes(int64_t rows, int64_t cols) {
    if (rows <= 0 || cols <= 0) return 16;
    const Flat16Plan p = flat16_plan(rows, cols);
    int64_t flat = p.nblocks * 8 + p.nblocks * 4 * 4 + 8 * kMaxChunks;  // block totals (int64) + span totals (int32) + chunk totals
    const int64_t resident = (int64_t)(kResMaxWGs + 4 + 4 * kResStampWGs + kResRoundWords) * 8;  // count words + control words (+ time stamps of the first workgroups) + round words
    if (resident > flat) flat = resident;
    const int64_t generic = (rows + 1) * 8;                  // row counts of the count / scan / scatter form
    return (flat > generic ? flat : generic) + 16;
}

int ct_bitmask_compress(const void* x, int dt, int64_t rows, int64_t cols, void* values, int64_t values_capacity, uint8_t* bitmask,
                        int64_t* row_offsets, int64_t* total, void* workspace, int64_t workspace_bytes, ct_stream_t stream) {
    const int es = dt_size(dt);
    CT_REQUIRE(es == 1 || es == 2 || es == 4, "unsupported element dtype %d", dt);
    CT_REQUIRE(rows >= 0 && cols >= 0 && values_capacity >= 0, "negative shape");
    CT_REQUIRE(total != nullptr, "total must not be NULL");
    if (rows == 0 || cols == 0) {
        return hip_check(hipMemsetAsync(total, 0, sizeof(int64_t), as_stream(stream)), "ct_bitmask_compress memset");
    }
    const int64_t need = ct_bitmask_compress_workspace_bytes(rows, cols);
    CT_REQUIRE(workspace != nullptr && workspace_bytes >= need, "workspace too small: %lld < %lld bytes", (long long)workspace_bytes,
               (long long)need);
    CT_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "workspace must be 8-byte aligned");
    CT_REQUIRE(aligned16(values), "values buffer must be 16-byte aligned");
    // 16- and 32-bit elements (the latter as pairs of halves): the resident form (x read once).  CT_BITMASK_RESIDENT=0 selects the two kernels
    // below for 16-bit elements (x read twice; 32-bit ones then take the generic path), which
    // are also what the caller falls back to when the resident form reports -1; 3 = time stamps of the first 1024 workgroups in the
    // workspace (tools/exp_r04.py bmstamps)
#ifdef CT_DIAG
    static const int resident_mode = []() { const char* e = std::getenv("CT_BITMASK_RESIDENT"); return e ? std::atoi(e) : 1; }();
#else
    constexpr int resident_mode = 1;
#endif
    // (round 6: 8-bit payloads — FP8 / int8 weights viewed as bytes — ride the row form too: a unit is 16 elements, the counts are in bytes)
    if ((((es == 2 || es == 4) && cols % 8 == 0) || (es == 1 && cols % 16 == 0)) && aligned16(x) && resident_mode) {
        const int64_t upr = cols * es / 16;  // 16-byte units per row
        const int64_t units = rows * upr;
        const int64_t wts = cdiv64(units, kWT);
        int dev = 0, cus = kCUs;
        if (hipGetDevice(&dev) == hipSuccess) {
            static std::atomic<int> cu_cache[64];
            int c = dev >= 0 && dev < 64 ? cu_cache[dev].load() : 0;
            if (c <= 0) {
                if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = kCUs;
                if (dev >= 0 && dev < 64) cu_cache[dev].store(c);
            }
            cus = c;
        }
        // wave-tiles per wave: as many as the registers hold, fewer when the tensor would otherwise leave CUs without a workgroup
        // (two 8-wave workgroups fit a CU; 4-wave workgroups, four per CU, measured the same: 41.8-43.3 us against 42.0-42.3)
        // 8-bit payloads (round 6): ONE tile per wave and four workgroups per CU (55 VGPRs) — the row form's 64 rows per tile are instruction-bound, and
        // full occupancy hides more of it: 8192^2 int8 36.0 -> 32.0 us
        const int wgs_per_cu = es == 1 ? 4 : 2;
        int64_t tpw = cdiv64(wts, (int64_t)cus * wgs_per_cu * kResWaves);
        const int keep_cap = es == 1 ? 1 : kResKeep;
        if (tpw > keep_cap) tpw = keep_cap;
        if (es == 2 && tpw == kResKeep) {
            // several residency rounds: a nearly empty last round costs up to 6 % (18944 x 3584 = 2.02 rounds of 512 workgroups: 48.6 us; with two tiles per
            // wave — 78 VGPRs, three workgroups per CU — 2.7 rounds of 768: 45.8).  Two tiles when they fill their rounds clearly better; sweep in DESIGN 5.4
            auto fill = [&](int64_t t, int64_t slots) {
                const int64_t nw = cdiv64(wts, (int64_t)kResWaves * t);
                return (double)nw / (double)(cdiv64(nw, slots) * slots);
            };
            const double f4 = fill(kResKeep, 2 * (int64_t)cus), f2 = fill(2, 3 * (int64_t)cus);
            if (f4 < 0.8 && f2 > f4 + 0.15) tpw = 2;
        }
        if (tpw < 1) tpw = 1;
        const int64_t wg_wts = (int64_t)kResWaves * tpw;             // wave-tiles per workgroup (<= 128 KB)
#ifdef CT_DIAG
        static const int64_t max_wgs = []() {  // per launch: the count words of the workspace (the knob exists for the chunking tests)
            const char* e = std::getenv("CT_BITMASK_RESIDENT_MAX_WGS");
            const int64_t v = e ? (int64_t)std::atoll(e) : (int64_t)kResMaxWGs;
            return v < 1 ? (int64_t)1 : (v > kResMaxWGs ? (int64_t)kResMaxWGs : v);
        }();
#else
        constexpr int64_t max_wgs = kResMaxWGs;  // per launch: the count words of the workspace
#endif
        const int64_t chunk_wts = max_wgs * wg_wts;
        const int64_t nchunks = cdiv64(wts, chunk_wts);
        if (nchunks <= 4096) {
            std::atomic<uint32_t>& generation = resident_generation();
            // unique per launch in this process (random start: words left in recycled device memory by another process carry no
            // matching tag either)
            const uint32_t gen0 = generation.fetch_add((uint32_t)nchunks) + 1u;
            // the tag of chunk k: top bit always set and never all ones, so that neither zero-filled nor 0xff-filled fresh memory can
            // look like a published word; a recycled word that happens to carry the current tag (2^-31 per word) is the
            // residual risk of not clearing the 64 KB of slots before every launch
            auto tag_of = [](uint32_t c) { return 0x80000000u | (c % 0x7ffffffeu); };
            unsigned long long* slots = static_cast<unsigned long long*>(workspace);
            unsigned long long* ctl = slots + kResMaxWGs;  // [2] / [3] running totals (alternating between chunks)
#ifdef CT_DIAG
            unsigned long long* stamps = resident_mode == 3 ? ctl + 4 : nullptr;
            static const unsigned long long wait_ticks = []() {  // 100 MHz ticks; default 2 ms, then self-help
                const char* e = std::getenv("CT_BITMASK_RESIDENT_WAIT_US");
                return (unsigned long long)(e ? std::atoll(e) : 2000ll) * 100ull;
            }();
#else
            constexpr unsigned long long wait_ticks = 2000ull * 100ull;  // 100 MHz ticks: 2 ms, then self-help
#endif
            for (int64_t k = 0; k < nchunks; ++k) {
                const int64_t w0 = k * chunk_wts;
                const int64_t cw = (wts - w0) < chunk_wts ? (wts - w0) : chunk_wts;
                const int64_t u0 = w0 * kWT;
                const int64_t cu = (units - u0) < cw * kWT ? (units - u0) : cw * kWT;
                const int64_t nwg = cdiv64(cw, wg_wts);
                uint8_t* bm0 = bitmask + (es == 4 ? u0 / 2 : (es == 1 ? u0 * 2 : u0));  // a mask byte covers one 16-bit unit, two 32-bit units or half an 8-bit unit
                const int mask_dwords = (es == 4 || cu % 4 == 0) && ((reinterpret_cast<uintptr_t>(bm0) & 3u) == 0);
                // the chunk's running total: straight into *total for the last chunk, else into one of two alternating workspace words
                unsigned long long* run_out = k + 1 == nchunks ? reinterpret_cast<unsigned long long*>(total) : ctl + 2 + (k & 1);
                // stagger (see the kernel): only when the launch is at least two full residency rounds (two workgroups per CU each) — with a
                // single round a delayed start is pure loss (4096^2: 15.0 -> 17.4 us).  Round 4: a GRADIENT over the first residency round —
                // workgroup b of the first 2 x CUs starts b / (2 x CUs) of one load phase late, so that counts resolve, values leave and
                // slots free up progressively instead of in two lockstep halves (round 3: the second workgroup of every CU one whole load
                // phase late).  Spread swept at 8192^2 with the tile-by-tile kernel: none 43.7, 4 us 43.0, 7 us 42.3, 8 us 42.1, 9 us 42.0,
                // 10 us 42.7 (the round-3 form: 44.9).  9 us per 128 KB workgroup = the load phase of a workgroup at the CU's share of the HBM rate.
                const bool stagger = nwg >= 2 * wgs_per_cu * (int64_t)cus;
                const int stagger_lo = 0, stagger_hi = stagger ? wgs_per_cu * cus : 0;
                const unsigned stagger_ticks = 0;
                const unsigned long long spread_ticks = (unsigned long long)(wg_wts * kWT * 16) * 900ull / (128ull * 1024ull);  // 100 MHz ticks: 9 us per 128 KB
                const unsigned stagger_slope_q8 = (unsigned)((spread_ticks * 256ull) / (unsigned long long)(wgs_per_cu * cus));
                // one "inclusive count through residency round r" word per round: a round r >= 1 workgroup polls it instead of the r x 2 x CUs
                // raw count words of the earlier rounds (44.9 -> 44.6 us by itself; kept: it halves the polling of the later rounds)
                const int round_wgs = (wgs_per_cu * cus <= kResMaxWGs && cdiv64(nwg, wgs_per_cu * (int64_t)cus) <= kResRoundWords) ? wgs_per_cu * cus : 0;
                unsigned long long* round_words = ctl + 4 + 4 * kResStampWGs;
#define CT_RESIDENT_WK(ES_, W_, R_, K_)                                                                                                              \
    hipLaunchKernelGGL((flat16_resident_kernel<K_, W_, ES_, R_>), dim3((unsigned)nwg), dim3(W_ * 64), 0, as_stream(stream),                   \
                       static_cast<const u32x4*>(x) + u0, float_kind(dt), cu, upr, rows, (int)tpw, static_cast<uint16_t*>(values),                 \
                       values_capacity * (ES_ == 4 ? 2 : 1), bm0, mask_dwords, row_offsets, u0, k ? ctl + 2 + ((k - 1) & 1) : nullptr, slots, run_out, \
                       tag_of(gen0 + (uint32_t)k), wait_ticks, CT_STAMPS_ARG(k == 0 ? stamps : nullptr) stagger_lo, stagger_hi, stagger_ticks, stagger_slope_q8,       \
                       round_wgs, round_words)
                // the kernel is instantiated per tiles-per-wave (round 6): its loads are unconditional straight-line code, so a wave of the KEEP = 4 kernel
                // with tpw = 2 read two tiles it never used (4096^2 bf16: 18.3 -> 14.9 us with KEEP = 2)
#define CT_RESIDENT_K(ES_, R_)                                                                 \
    switch ((int)tpw) {                                                                         \
        case 1: CT_RESIDENT_WK(ES_, kResWaves, R_, 1); break;                                   \
        case 2: CT_RESIDENT_WK(ES_, kResWaves, R_, 2); break;                                   \
        case 3: CT_RESIDENT_WK(ES_, kResWaves, R_, 3); break;                                   \
        default: CT_RESIDENT_WK(ES_, kResWaves, R_, kResKeep); break;                           \
    }
                if (es == 4) CT_RESIDENT_K(4, 1)                     // the row form: 69 % against 65 % of the HBM peak at 8192^2 float32
                else if (es == 1) CT_RESIDENT_WK(1, kResWaves, 1, 1);  // 8-bit payloads: the row form, 64 rows per tile
                else CT_RESIDENT_K(2, 0)                             // the unit form: the row form is scalar-bound at 32 rows per tile
#undef CT_RESIDENT_K
#undef CT_RESIDENT_WK
            }
            CT_LAUNCH_CHECK("ct_bitmask_compress[resident]");
        }
    }
    if (es == 2 && cols % 8 == 0 && aligned16(x)) {
        const Flat16Plan p = flat16_plan(rows, cols);
        int64_t* block_tot = static_cast<int64_t*>(workspace);
        int32_t* span_tot = reinterpret_cast<int32_t*>(block_tot + p.nblocks);
        const int mask_dwords = (p.units % 4 == 0) && ((reinterpret_cast<uintptr_t>(bitmask) & 3u) == 0);
        // One chunk.  (Walking the tensor in cache-sized chunks — count chunk k, scatter chunk k, ... — so that the scatter's re-read
        // of x hits a cache was measured in round 2: 64 / 32 / 16 / 8 MB chunks -> 80 / 102 / 155 / 280 us against 66; the loop below
        // still carries a running total from chunk to chunk, and is exercised with one chunk.)
        const int64_t units_per_block = (int64_t)4 * p.span * kWT;
        const int64_t blocks_per_chunk = p.nblocks;
        // running totals: chunk k leaves its end in chunk_tot[k] (its own word: the next chunk's waves read it while nothing writes
        // it), the last chunk in `total`
        int64_t* chunk_tot = reinterpret_cast<int64_t*>(span_tot + 4 * p.nblocks);
        int chunk = 0;
        for (int64_t b0 = 0; b0 < p.nblocks; b0 += blocks_per_chunk, ++chunk) {
            const int64_t nb = (p.nblocks - b0) < blocks_per_chunk ? (p.nblocks - b0) : blocks_per_chunk;
            const int64_t u0 = b0 * units_per_block;
            const int64_t cu = (p.units - u0) < nb * units_per_block ? (p.units - u0) : nb * units_per_block;
            const u32x4* xc = static_cast<const u32x4*>(x) + u0;
            hipLaunchKernelGGL(flat16_count_kernel, dim3((unsigned)nb), dim3(kBlock), 0, as_stream(stream), xc, float_kind(dt), cu, p.span, bitmask + u0,
                               mask_dwords, block_tot + b0, span_tot + 4 * b0);
            hipLaunchKernelGGL(flat16_scatter_kernel, dim3((unsigned)nb), dim3(kBlock), 0, as_stream(stream), xc, float_kind(dt), cu, p.span, cols / 8, rows,
                           static_cast<uint16_t*>(values), values_capacity, row_offsets, (b0 + nb >= p.nblocks) ? total : chunk_tot + chunk,
                           block_tot + b0, span_tot + 4 * b0, u0, chunk ? chunk_tot + chunk - 1 : static_cast<const int64_t*>(nullptr));
        }
        CT_LAUNCH_CHECK("ct_bitmask_compress[flat16]");
    }
    // other element sizes / ragged rows: count, single-block scan, scatter (all on the stream)
    CT_REQUIRE(values_capacity >= rows * cols, "the generic path needs values_capacity >= numel (%lld)", (long long)(rows * cols));
    int64_t* counts = static_cast<int64_t*>(workspace);
    int rc = ct_bitmask_count(x, dt, rows, cols, bitmask, counts, stream);
    if (rc) return rc;
    rc = ct_exclusive_scan_i64(counts, rows, row_offsets, total, stream);
    if (rc) return rc;
    return ct_bitmask_scatter(x, dt, rows, cols, row_offsets, values, stream);
}

int64_t ct_bitmask_batch_plan(ct_bitmask_item* items, int n, int64_t* workspace_bytes) {
    if (n < 0 || (n > 0 && items == nullptr) || workspace_bytes == nullptr) {
        set_error("ct_bitmask_batch_plan: bad arguments");
        return -1;
    }
    const int cus = device_cus();
    int64_t blocks = 0, slots = 0;
    const int es0 = n > 0 ? dt_size(items[0].dt) : 2;
    auto tag_of = [](uint32_t c) { return 0x80000000u | (c % 0x7ffffffeu); };
    const uint32_t gen0 = resident_generation().fetch_add((uint32_t)(n > 0 ? n : 1)) + 1u;
    // wave-tiles per wave.  A single tensor spreads thin (fewer tiles per wave) so that every CU gets a workgroup; a TABLE fills the chip with its
    // tensors, and what limits it is bytes in flight per resident workgroup x 2 workgroups per CU / a workgroup's lifetime (a latency chain of
    // ~8-12 us whatever it carries): with 1-2 tiles per wave (the single-tensor rule on 1-23 MB tensors) 154 tensors ran at 3.4 TB/s.  If the whole
    // table still makes at least two residency rounds with full registers, every item takes kResKeep tiles per wave.
    int64_t all_wts = 0;
    for (int i = 0; i < n; ++i)
        if (items[i].rows > 0 && items[i].cols > 0) all_wts += cdiv64(items[i].rows * (items[i].cols * dt_size(items[i].dt) / 16), kWT);
    const bool full_tiles = all_wts >= (int64_t)2 * 2 * cus * kResWaves * kResKeep;
    for (int i = 0; i < n; ++i) {
        ct_bitmask_item& it = items[i];
        const int es = dt_size(it.dt);
        const bool ok = (es == 1 || es == 2 || es == 4) && es == es0 && it.rows > 0 && it.cols > 0 && it.cols % (es == 1 ? 16 : 8) == 0 && it.x && it.values && it.bitmask && it.row_offsets && it.total &&
                        aligned16(it.x) && aligned16(it.values) && it.values_capacity >= 0 && (reinterpret_cast<uintptr_t>(it.total) & 7u) == 0;
        if (!ok) {
            set_error("ct_bitmask_batch_plan: item %d (rows %lld, cols %lld, dtype %d) is not eligible for the batched sparse-bitmask compress (8-, 16- or 32-bit "
                      "payloads of ONE element size per table, cols x element size %% 16 == 0, non-empty, x / values 16-byte aligned, no NULL pointer)", i, (long long)it.rows,
                      (long long)it.cols, it.dt);
            return -1;
        }
        it.is_float = is_float_dt(it.dt) ? 1 : 0;
        it.upr = it.cols * es / 16;
        it.units = it.rows * it.upr;
        const int64_t wts = cdiv64(it.units, kWT);
        int64_t tpw = full_tiles ? kResKeep : cdiv64(wts, (int64_t)cus * 2 * kResWaves);  // else as the single-tensor launch: fewer tiles for a small tensor
        if (tpw > kResKeep) tpw = kResKeep;
        if (tpw < 1) tpw = 1;
        const int64_t nwg = cdiv64(wts, (int64_t)kResWaves * tpw);
        if (nwg > kResMaxWGs) {
            set_error("ct_bitmask_batch_plan: item %d holds more than 1 GiB of payload; compress it with ct_bitmask_compress", i);
            return -1;
        }
        it.tpw = (int32_t)tpw;
        it.nwg = (int32_t)nwg;
        it.mask_dwords = ((es == 4 || it.units % 4 == 0) && ((reinterpret_cast<uintptr_t>(it.bitmask) & 3u) == 0)) ? 1 : 0;
        it.gen = tag_of(gen0 + (uint32_t)i);
        it.first_block = blocks;
        it.slots_offset = slots;
        blocks += nwg;
        slots += (nwg + 15) / 16 * 16;  // 128-byte lines: two tensors never share one
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_bitmask_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    *workspace_bytes = slots * 8 + 16;
    return blocks;
}

int ct_bitmask_compress_batch(const ct_bitmask_item* items_dev, int n, int64_t total_blocks, int element_size, void* workspace, int64_t workspace_bytes,
                              ct_stream_t stream) {
    CT_REQUIRE(element_size == 1 || element_size == 2 || element_size == 4, "batched sparse-bitmask compress: 8-, 16- or 32-bit payloads, got element size %d", element_size);
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    CT_REQUIRE(items_dev != nullptr && workspace != nullptr && workspace_bytes >= 16 && (reinterpret_cast<uintptr_t>(workspace) & 7u) == 0, "table / workspace NULL or misaligned");
    constexpr unsigned long long wait_ticks = 2000ull * 100ull;  // 100 MHz ticks: 2 ms, then self-help (as the single-tensor launch)
    if (element_size == 1)
        hipLaunchKernelGGL((flat16_resident_batch_kernel<1>), dim3((unsigned)total_blocks), dim3(kResWaves * 64), 0, as_stream(stream), items_dev, n,
                           static_cast<unsigned long long*>(workspace), wait_ticks);
    else if (element_size == 4)
        hipLaunchKernelGGL((flat16_resident_batch_kernel<4>), dim3((unsigned)total_blocks), dim3(kResWaves * 64), 0, as_stream(stream), items_dev, n,
                           static_cast<unsigned long long*>(workspace), wait_ticks);
    else
        hipLaunchKernelGGL((flat16_resident_batch_kernel<2>), dim3((unsigned)total_blocks), dim3(kResWaves * 64), 0, as_stream(stream), items_dev, n,
                           static_cast<unsigned long long*>(workspace), wait_ticks);
    CT_LAUNCH_CHECK("ct_bitmask_compress_batch");
}

int64_t ct_bitmask_decompress_batch_plan(ct_bitmask_ditem* items, int n) {
    if (n < 0 || (n > 0 && items == nullptr)) {
        set_error("ct_bitmask_decompress_batch_plan: bad arguments");
        return -1;
    }
    int64_t blocks = 0;
    const int es0 = n > 0 ? dt_size(items[0].dt) : 2;
    for (int i = 0; i < n; ++i) {
        ct_bitmask_ditem& it = items[i];
        const int es = dt_size(it.dt);
        const bool ok = (es == 2 || es == 4) && es == es0 && it.rows > 0 && it.cols > 0 && (it.cols * es / 2) % 32 == 0 && it.values_len >= 0 && it.bitmask && it.row_offsets && it.out &&
                        (it.values != nullptr || it.values_len == 0) && aligned16(it.values) && aligned16(it.out) && (reinterpret_cast<uintptr_t>(it.bitmask) & 3u) == 0;
        if (!ok) {
            set_error("ct_bitmask_decompress_batch_plan: item %d (rows %lld, cols %lld, dtype %d) is not eligible for the batched sparse-bitmask decompress (16- or 32-bit "
                      "payloads of ONE element size per table, cols x element size %% 64 == 0, row_offsets given, values / out 16-byte and bitmask 4-byte aligned)", i,
                      (long long)it.rows, (long long)it.cols, it.dt);
            return -1;
        }
        const int64_t hcols = it.cols * es / 2;  // the row in 16-bit items
        // the tile form (the kernel's dispatch): rows of several tiles / the row is one tile / 16-bit rows shorter than a tile: flat tiles
        it.single = (es == 2 && hcols % kTile16 != 0) ? kDecFlat : (hcols > kTile16 ? kDecRows : kDecSingle);  // as the single-tensor launch
        it.first_block = blocks;
        blocks += it.single == kDecFlat ? cdiv64(it.rows * (hcols / 8), kTile16 / 8) : it.rows * cdiv64(hcols, kTile16);
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_bitmask_decompress_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    return blocks;
}

int ct_bitmask_decompress_batch(const ct_bitmask_ditem* items_dev, int n, int64_t total_blocks, int element_size, ct_stream_t stream) {
    CT_REQUIRE(element_size == 2 || element_size == 4, "batched sparse-bitmask decompress: 16- or 32-bit payloads, got element size %d", element_size);
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    CT_REQUIRE(items_dev != nullptr, "table is NULL");
    if (element_size == 4) hipLaunchKernelGGL((bitmask_decompress16_batch_kernel<4>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n);
    else hipLaunchKernelGGL((bitmask_decompress16_batch_kernel<2>), dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n);
    CT_LAUNCH_CHECK("ct_bitmask_decompress_batch");
}

int64_t ct_copy_batch_plan(ct_copy_item* items, int n) {
    if (n < 0 || (n > 0 && items == nullptr)) {
        set_error("ct_copy_batch_plan: bad arguments");
        return -1;
    }
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].bytes < 0 || (items[i].bytes > 0 && (!items[i].src || !items[i].dst))) {
            set_error("ct_copy_batch_plan: item %d has a negative size or a NULL pointer", i);
            return -1;
        }
        items[i].first_block = blocks;
        blocks += cdiv64(items[i].bytes, (int64_t)kBlock * 64);
    }
    if (blocks >= ((int64_t)1 << 31)) {
        set_error("ct_copy_batch_plan: %lld workgroups exceed one launch; split the batch", (long long)blocks);
        return -1;
    }
    return blocks;
}

int ct_copy_batch(const ct_copy_item* items_dev, int n, int64_t total_blocks, ct_stream_t stream) {
    CT_REQUIRE(n >= 0 && total_blocks >= 0 && total_blocks < ((int64_t)1 << 31), "bad batch size");
    if (n == 0 || total_blocks == 0) return CT_OK;
    CT_REQUIRE(items_dev != nullptr, "table is NULL");
    hipLaunchKernelGGL(copy_batch_kernel, dim3((unsigned)total_blocks), dim3(kBlock), 0, as_stream(stream), items_dev, n);
    CT_LAUNCH_CHECK("ct_copy_batch");
}

int ct_bitmask_decompress(const void* values, int64_t values_len, const uint8_t* bitmask, const int64_t* row_offsets, int64_t fixed_row_nnz,
                          int dt, int64_t rows, int64_t cols, void* out, ct_stream_t stream) {
    const int es = dt_size(dt);
    CT_REQUIRE(es == 1 || es == 2 || es == 4, "unsupported element dtype %d", dt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    CT_REQUIRE(row_offsets != nullptr || fixed_row_nnz >= 0, "row_offsets is NULL and fixed_row_nnz < 0");
    if (rows == 0 || cols == 0) return CT_OK;
    const int vec_out = (cols % 8 == 0) && aligned16(out);
    CT_REQUIRE(values_len >= 0, "negative values length");
    // aligned 16-byte loads of the value runs need a 16-byte aligned base; vectors that would
    // cross values_len are read element-wise in the kernel
    const int vec_in = aligned16(values);
    // (2:4-regular rows of 16-bit payloads measure the same on this kernel as on the general kernel's local-expand path: 33.9 / 34.8 us)
    if ((es == 2 || es == 4) && (cols * es / 2) % 32 == 0 && vec_out && vec_in && (reinterpret_cast<uintptr_t>(bitmask) & 3u) == 0 &&
        values_len < ((int64_t)1 << 61)) {
        const int64_t hcols = cols * es / 2;  // the row in 16-bit items
        const int64_t tiles = rows * cdiv64(hcols, kTile16);
        const unsigned grid = (unsigned)(tiles < ((int64_t)1 << 30) ? tiles : ((int64_t)1 << 30));  // exact grid measured best
#define CT_DEC16(FORM_, ES_)                                                                                                                        \
    hipLaunchKernelGGL((bitmask_decompress16_kernel<FORM_, ES_>), dim3(grid), dim3(kBlock), 0, as_stream(stream), static_cast<const uint16_t*>(values), \
                       values_len, bitmask, row_offsets, fixed_row_nnz, rows, cols, static_cast<uint16_t*>(out))
        if (es == 4) { if (hcols <= kTile16) CT_DEC16(kDecSingle, 4); else CT_DEC16(kDecRows, 4); }
        else if (hcols % kTile16 != 0) {
            // round 6: rows that are not whole tiles -> flat tiles of 1024 consecutive units (see the kernel).  Short rows first (2048 columns: one row per
            // workgroup moved 6 KB); long rows too — the last tile of an 11008-column row is a third full: 4096 x 11008 30.1 -> 26.5 us, 12288 30.0 -> 27.5,
            // 13824 39.5 -> 37.4, 28672 119.8 -> 116.2
            const int64_t ftiles = cdiv64(rows * (hcols / 8), kTile16 / 8);
            hipLaunchKernelGGL((bitmask_decompress16_kernel<kDecFlat, 2>), dim3((unsigned)(ftiles < ((int64_t)1 << 30) ? ftiles : ((int64_t)1 << 30))), dim3(kBlock), 0,
                               as_stream(stream), static_cast<const uint16_t*>(values), values_len, bitmask, row_offsets, fixed_row_nnz, rows, cols, static_cast<uint16_t*>(out));
        } else { if (hcols <= kTile16) CT_DEC16(kDecSingle, 2); else CT_DEC16(kDecRows, 2); }
#undef CT_DEC16
        CT_LAUNCH_CHECK("ct_bitmask_decompress[16]");
    }
    // 8-bit payloads: the byte-granular LDS-window kernel, for the unstructured codec (row offsets) and the 2:4 codec (fixed_row_nnz) alike — the
    // general kernel's 2:4-regular row path ran 8192^2 int8 at 33.4 us HBM-cold, this one at 22.2-23.0.
    if (es == 1 && cols % 16 == 0 && (cols <= 16384 || cols % 64 == 0) && vec_out && vec_in && (reinterpret_cast<uintptr_t>(bitmask) & 3u) == 0 &&
        values_len < ((int64_t)1 << 61)) {
#define CT_DEC8(SINGLE_, UPL_, FLAT_, GRID_)                                                                                                                     \
    hipLaunchKernelGGL((bitmask_decompress8_kernel<SINGLE_, UPL_, FLAT_>), dim3(GRID_), dim3(kBlock), 0, as_stream(stream), static_cast<const uint8_t*>(values), \
                       values_len, bitmask, row_offsets, fixed_row_nnz, rows, cols, static_cast<uint8_t*>(out))
        if ((cols < 8192 || (cols > 16384 && cols % 16384 != 0)) && cols % 32 == 0) {
            // flat tiles of 1024 units over the flattened tensor: rows shorter than half a tile, and rows of several tiles whose last one would be
            // partial (3584 x 18944 29.2 -> 24.8 us); between 8192 and 16384 columns the single partial tile wins (4096 x 11008 17.2 against 17.8)
            const int64_t ftiles = cdiv64(rows * (cols / 16), 1024);
            CT_DEC8(false, 4, true, (unsigned)(ftiles < ((int64_t)1 << 30) ? ftiles : ((int64_t)1 << 30)));
        } else {
            // units per lane: the smallest tile that holds a whole row (a row of <= 16384 columns is then ONE tile: SINGLE), else 16384-column tiles
            const int upl = cols <= 4096 ? 1 : (cols <= 8192 ? 2 : 4);
            const int64_t tiles = rows * cdiv64(cols, (int64_t)kBlock * upl * 16);
            const unsigned grid8 = (unsigned)(tiles < ((int64_t)1 << 30) ? tiles : ((int64_t)1 << 30));
            if (upl == 1) CT_DEC8(true, 1, false, grid8);
            else if (upl == 2) CT_DEC8(true, 2, false, grid8);
            else if (cols <= 16384) CT_DEC8(true, 4, false, grid8);
            else CT_DEC8(false, 4, false, grid8);
        }
#undef CT_DEC8
        CT_LAUNCH_CHECK("ct_bitmask_decompress[8]");
    }
    // 4096 resident-ish workgroups, grid-strided over rows: measured best on MI355X (tools/kbench)
    const unsigned grid = (unsigned)(rows < 4096 ? rows : 4096);
    CT_ES_SWITCH(es, hipLaunchKernelGGL((bitmask_decompress_kernel<ES>), dim3(grid), dim3(kBlock), 0, as_stream(stream), values,
                                        values_len, bitmask, row_offsets, fixed_row_nnz, rows, cols, out, vec_out, vec_in));
    CT_LAUNCH_CHECK("ct_bitmask_decompress");
}

int ct_sparse24_compress(const void* x, int dt, int64_t rows, int64_t cols, void* values, uint8_t* bitmask, ct_stream_t stream) {
    const int es = dt_size(dt);
    CT_REQUIRE(es == 1 || es == 2 || es == 4, "unsupported element dtype %d", dt);
    CT_REQUIRE(rows >= 0 && cols >= 0, "negative shape");
    CT_REQUIRE(cols % 8 == 0, "2:4 bitmask compression needs the row length to be a multiple of 8, got %lld", (long long)cols);
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t units = rows * (cols / 8);
    CT_REQUIRE(aligned16(values), "values buffer must be 16-byte aligned");
    const int vec = aligned16(x);
    if (vec && (es == 1 || es == 2) && units % 2 == 0 && (reinterpret_cast<uintptr_t>(bitmask) & 1u) == 0 && units / 2 < ((int64_t)1 << 38)) {
        const int64_t pairs = units / 2;
        dim3 g((unsigned)cdiv64(pairs, kBlock));
        if (es == 2) hipLaunchKernelGGL((sparse24_pair_kernel<2>), g, dim3(kBlock), 0, as_stream(stream), x, float_kind(dt), pairs, values, bitmask);
        else hipLaunchKernelGGL((sparse24_pair_kernel<1>), g, dim3(kBlock), 0, as_stream(stream), x, float_kind(dt), pairs, values, bitmask);
        CT_LAUNCH_CHECK("ct_sparse24_compress[pairs]");
    }
    CT_ES_SWITCH(es, hipLaunchKernelGGL((sparse24_compress_kernel<ES>), dim3(grid_1d(units)), dim3(kBlock), 0, as_stream(stream), x,
                                        float_kind(dt), units, values, bitmask, (uint8_t*)nullptr, vec));
    CT_LAUNCH_CHECK("ct_sparse24_compress");
}

int ct_sparse24_mask(const void* x, int dt, int64_t numel, uint8_t* mask, ct_stream_t stream) {
    const int es = dt_size(dt);
    CT_REQUIRE(es == 1 || es == 2 || es == 4, "unsupported element dtype %d", dt);
    CT_REQUIRE(numel >= 0 && numel % 8 == 0, "2:4 mask needs a multiple of 8 elements, got %lld", (long long)numel);
    if (numel == 0) return CT_OK;
    const int vec = aligned16(x);
    CT_ES_SWITCH(es, hipLaunchKernelGGL((sparse24_compress_kernel<ES>), dim3(grid_1d(numel / 8)), dim3(kBlock), 0, as_stream(stream), x,
                                        float_kind(dt), numel / 8, (void*)nullptr, (uint8_t*)nullptr, mask, vec));
    CT_LAUNCH_CHECK("ct_sparse24_mask");
}

}  // extern "C"
