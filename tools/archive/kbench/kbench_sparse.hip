// kbench_sparse.hip — developer micro-benchmark for sparse-bitmask decompress variants (bf16 payload).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 kbench_sparse.hip -o kbench_sparse
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int kBlock = 256;

__device__ __forceinline__ void wave_rank(uint32_t m, uint32_t& pre, int& wave_total) {
    pre = 0; wave_total = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned long long plane = __ballot((m >> k) & 1u);
        pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(plane >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)plane, pre));
        wave_total += __popcll(plane);
    }
}

// ---------------------------------------------------------------- V1: LUT-driven expansion, UPL units per lane, one row-chunk per block
// LUT in LDS: rank bytes (8 per mask value) and expand masks (4 dwords per mask value)
template <int UPL>
__global__ __launch_bounds__(kBlock) void dec_v1(const uint16_t* __restrict__ vin, int64_t values_len, const uint8_t* __restrict__ bitmask,
                                                 const int64_t* __restrict__ row_offsets, int64_t rows, int64_t cols, uint16_t* __restrict__ out) {
    constexpr int SUPER = kBlock * UPL * 8;
    __shared__ __attribute__((aligned(16))) uint16_t s_val[SUPER + 16];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut_mask[256 * 4];
    __shared__ __attribute__((aligned(8))) uint32_t s_lut_rank[256 * 2];
    __shared__ int s_tot[UPL][kBlock / 64];
    {   // LUT entry for mask value = threadIdx.x
        const uint32_t mv = threadIdx.x;
        uint32_t r_lo = 0, r_hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t rk = __popc(mv & ((1u << k) - 1u));
            if (k < 4) r_lo |= rk << (8 * k); else r_hi |= rk << (8 * (k - 4));
        }
        s_lut_rank[mv * 2] = r_lo; s_lut_rank[mv * 2 + 1] = r_hi;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            s_lut_mask[mv * 4 + j] = (((mv >> (2 * j)) & 1u) ? 0xffffu : 0u) | (((mv >> (2 * j + 1)) & 1u) ? 0xffff0000u : 0u);
    }
    __syncthreads();
    const int64_t bcols = cols >> 3;
    const int64_t chunks_per_row = (cols + SUPER - 1) / SUPER;
    const int64_t nwork = rows * chunks_per_row;
    const int wave = threadIdx.x >> 6;
    for (int64_t wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
        const int64_t row = wk / chunks_per_row, chunk = wk - row * chunks_per_row;
        const int64_t cbase = chunk * SUPER;
        uint32_t m[UPL];
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int64_t u = (cbase >> 3) + (int64_t)i * kBlock + threadIdx.x;
            m[i] = u < bcols ? bitmask[row * bcols + u] : 0u;
        }
        // start of this chunk's run: row offset + popcount of the row's mask bytes before the chunk
        int64_t run = row_offsets[row];
        int pre_chunk = 0;
        if (chunk > 0) {
            const int64_t nb = cbase >> 3;  // bytes before the chunk (multiple of kBlock*UPL)
            int c = 0;
            for (int64_t b = (int64_t)threadIdx.x * 4; b < nb; b += kBlock * 4) c += __popc(*reinterpret_cast<const uint32_t*>(bitmask + row * bcols + b));
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
            pre_chunk = c;  // per-wave partial; combined below through s_tot slot
        }
        uint32_t rank[UPL];
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            int wt;
            wave_rank(m[i], rank[i], wt);
            if ((threadIdx.x & 63) == 0) s_tot[i][wave] = wt;
        }
        __shared__ int s_pre[kBlock / 64];
        if ((threadIdx.x & 63) == 0) s_pre[wave] = pre_chunk;
        __syncthreads();
        int running = 0;
#pragma unroll
        for (int i = 0; i < UPL; ++i)
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) {
                const int t = s_tot[i][w];
                if (w == wave) rank[i] += running;
                running += t;
            }
        const int total = running;
        if (chunk > 0) run += s_pre[0] + s_pre[1] + s_pre[2] + s_pre[3];
        // stage the value run (aligned 16 B loads)
        const uintptr_t src = reinterpret_cast<uintptr_t>(vin + run);
        const int shift = (int)((src & 15u) >> 1);
        {
            const int nvec = (shift + total + 7) >> 3;
            const int64_t e0 = run - shift;
            const u32x4* g = reinterpret_cast<const u32x4*>(vin + e0);
            for (int v = threadIdx.x; v < nvec; v += kBlock) {
                if (e0 + (int64_t)(v + 1) * 8 <= values_len) reinterpret_cast<u32x4*>(s_val)[v] = g[v];
                else for (int j = 0; j < 8; ++j) { const int64_t gi = e0 + (int64_t)v * 8 + j; s_val[v * 8 + j] = gi < values_len ? vin[gi] : 0; }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int64_t u = (cbase >> 3) + (int64_t)i * kBlock + threadIdx.x;
            if (u >= bcols) continue;
            const uint32_t mv = m[i];
            const u32x2 rk = *reinterpret_cast<const u32x2*>(&s_lut_rank[mv * 2]);
            const u32x4 em = *reinterpret_cast<const u32x4*>(&s_lut_mask[mv * 4]);
            const uint16_t* sp = s_val + shift + rank[i];
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t rr = j < 2 ? rk.x : rk.y;
                const uint32_t r0 = (rr >> (16 * (j & 1))) & 0xffu, r1 = (rr >> (16 * (j & 1) + 8)) & 0xffu;
                w[j] = (uint32_t)sp[r0] | ((uint32_t)sp[r1] << 16);
            }
            __builtin_nontemporal_store(u32x4{w[0] & em.x, w[1] & em.y, w[2] & em.z, w[3] & em.w}, reinterpret_cast<u32x4*>(out + row * cols + (u << 3)));
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------- V2: as V1 (single chunk per row) but the value run is staged BEFORE the ranks
// (its length comes from the next row offset), so mask and value loads are in flight together
template <int UPL>
__global__ __launch_bounds__(kBlock) void dec_v2(const uint16_t* __restrict__ vin, int64_t values_len, const uint8_t* __restrict__ bitmask,
                                                 const int64_t* __restrict__ row_offsets, int64_t rows, int64_t cols, uint16_t* __restrict__ out) {
    constexpr int SUPER = kBlock * UPL * 8;
    __shared__ __attribute__((aligned(16))) uint16_t s_val[SUPER + 16];
    __shared__ __attribute__((aligned(16))) uint32_t s_lut_mask[256 * 4];
    __shared__ __attribute__((aligned(8))) uint32_t s_lut_rank[256 * 2];
    __shared__ int s_tot[UPL][kBlock / 64];
    {
        const uint32_t mv = threadIdx.x;
        uint32_t r_lo = 0, r_hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t rk = __popc(mv & ((1u << k) - 1u));
            if (k < 4) r_lo |= rk << (8 * k); else r_hi |= rk << (8 * (k - 4));
        }
        s_lut_rank[mv * 2] = r_lo; s_lut_rank[mv * 2 + 1] = r_hi;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            s_lut_mask[mv * 4 + j] = (((mv >> (2 * j)) & 1u) ? 0xffffu : 0u) | (((mv >> (2 * j + 1)) & 1u) ? 0xffff0000u : 0u);
    }
    const int64_t bcols = cols >> 3;
    const int wave = threadIdx.x >> 6;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int64_t run = row_offsets[row];
        const int64_t row_end = row + 1 < rows ? row_offsets[row + 1] : values_len;
        uint32_t m[UPL];
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int64_t u = (int64_t)i * kBlock + threadIdx.x;
            m[i] = u < bcols ? bitmask[row * bcols + u] : 0u;
        }
        const uintptr_t src = reinterpret_cast<uintptr_t>(vin + run);
        const int shift = (int)((src & 15u) >> 1);
        {
            const int len = (int)(row_end - run);
            const int nvec = (shift + len + 7) >> 3;
            const int64_t e0 = run - shift;
            const u32x4* g = reinterpret_cast<const u32x4*>(vin + e0);
            for (int v = threadIdx.x; v < nvec; v += kBlock) {
                if (e0 + (int64_t)(v + 1) * 8 <= values_len) reinterpret_cast<u32x4*>(s_val)[v] = g[v];
                else for (int j = 0; j < 8; ++j) { const int64_t gi = e0 + (int64_t)v * 8 + j; s_val[v * 8 + j] = gi < values_len ? vin[gi] : 0; }
            }
        }
        uint32_t rank[UPL];
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            int wt;
            wave_rank(m[i], rank[i], wt);
            if ((threadIdx.x & 63) == 0) s_tot[i][wave] = wt;
        }
        __syncthreads();
        int running = 0;
#pragma unroll
        for (int i = 0; i < UPL; ++i)
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) {
                const int t = s_tot[i][w];
                if (w == wave) rank[i] += running;
                running += t;
            }
#pragma unroll
        for (int i = 0; i < UPL; ++i) {
            const int64_t u = (int64_t)i * kBlock + threadIdx.x;
            if (u >= bcols) continue;
            const uint32_t mv = m[i];
            const u32x2 rk = *reinterpret_cast<const u32x2*>(&s_lut_rank[mv * 2]);
            const u32x4 em = *reinterpret_cast<const u32x4*>(&s_lut_mask[mv * 4]);
            const uint16_t* sp = s_val + shift + rank[i];
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t rr = j < 2 ? rk.x : rk.y;
                const uint32_t r0 = (rr >> (16 * (j & 1))) & 0xffu, r1 = (rr >> (16 * (j & 1) + 8)) & 0xffu;
                w[j] = (uint32_t)sp[r0] | ((uint32_t)sp[r1] << 16);
            }
            __builtin_nontemporal_store(u32x4{w[0] & em.x, w[1] & em.y, w[2] & em.z, w[3] & em.w}, reinterpret_cast<u32x4*>(out + row * cols + (u << 3)));
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------- V3: scan-based ranks + window/perm expansion
// * mask side: every lane loads ONE dword of the bitmask (4 consecutive units), popcounts it, the
//   block does a DPP wave scan + 4 wave totals; the owner lane publishes (rank << 8 | mask byte)
//   for its 4 units in LDS, the consumer lane (unit i*256+tid: contiguous 1 KiB stores per wave)
//   reads it back.  ~1 VALU op per element instead of 4 for the 8-ballot-plane ranks.
// * value side: output dword j of a unit (elements 2j, 2j+1) is a 32-bit WINDOW of the staged value
//   run starting at element q_j = rank + popc(m & ((1 << 2j) - 1)): two aligned LDS dwords and ONE
//   v_perm_b32 whose selector (LUT keyed by mask byte and rank parity) does the funnel shift, the
//   placement of a lone element and the zeroing.  3 VALU ops + 1 ds_read2_b32 per output dword.
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

template <bool NTST>
__global__ __launch_bounds__(kBlock) void dec_v3(const uint16_t* __restrict__ vin, int64_t values_len, const uint8_t* __restrict__ bitmask,
                                                 const int64_t* __restrict__ row_offsets, int64_t rows, int64_t cols, uint16_t* __restrict__ out) {
    constexpr int T = 8192;
    __shared__ __attribute__((aligned(16))) uint16_t s_val[T + 32];
    __shared__ __attribute__((aligned(16))) uint32_t s_sel[2 * 256 * 4];
    __shared__ uint32_t s_off[256];
    __shared__ __attribute__((aligned(16))) uint32_t s_unit[1024];
    __shared__ __attribute__((aligned(16))) int s_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const uint32_t mv = tid;
        uint32_t off = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t pj = __popc(mv & ((1u << (2 * j)) - 1u));
            off |= (2u * pj) << (8 * j);
            const uint32_t b0 = (mv >> (2 * j)) & 1u, b1 = (mv >> (2 * j + 1)) & 1u;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint32_t x = (p + pj) & 1u;                     // window starts at byte 2 of the low dword
                const uint32_t w01 = x ? 0x0302u : 0x0100u;           // selector of window bytes 0,1
                const uint32_t w23 = x ? 0x0504u : 0x0302u;           // window bytes 2,3
                uint32_t sel;
                if (b0 && b1) sel = w01 | (w23 << 16);
                else if (b0) sel = w01 | 0x0c0c0000u;
                else if (b1) sel = 0x0c0cu | (w01 << 16);
                else sel = 0x0c0c0c0cu;
                s_sel[(p * 256 + mv) * 4 + j] = sel;
            }
        }
        s_off[mv] = off;
    }
    const int64_t bcols = cols >> 3;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {   // single tile per row (cols <= T) in this harness
        const int nu = (int)bcols;
        const uint32_t md = (4 * tid < nu) ? *reinterpret_cast<const uint32_t*>(bitmask + row * bcols + 4 * tid) : 0u;
        const int64_t run = row_offsets[row];
        const int64_t row_end = row + 1 < rows ? row_offsets[row + 1] : values_len;
        const int shift = (int)(run & 7);
        const int total_known = (int)(row_end - run);
        const int nvec = (shift + total_known + 7) >> 3;
        const int64_t e0 = run - shift;
        const u32x4* g = reinterpret_cast<const u32x4*>(vin + e0);
        u32x4 vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = tid + k * kBlock;
            if (v < nvec && e0 + (int64_t)(v + 1) * 8 <= values_len) vv[k] = g[v];
        }
        // ranks
        const int c = __popc(md);
        const int incl = wave_incl_scan(c);
        if (lane == 63) s_tot[wave] = incl;
        const uint32_t r0 = (uint32_t)(incl - c);
        const uint32_t r1 = r0 + __popc(md & 0xffu), r2 = r0 + __popc(md & 0xffffu), r3 = r0 + __popc(md & 0xffffffu);
        reinterpret_cast<u32x4*>(s_unit)[tid] = u32x4{(r0 << 8) | (md & 0xffu), (r1 << 8) | ((md >> 8) & 0xffu), (r2 << 8) | ((md >> 16) & 0xffu), (r3 << 8) | (md >> 24)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = tid + k * kBlock;
            if (v < nvec) {
                if (e0 + (int64_t)(v + 1) * 8 <= values_len) reinterpret_cast<u32x4*>(s_val)[v] = vv[k];
                else for (int j = 0; j < 8; ++j) { const int64_t gi = e0 + (int64_t)v * 8 + j; s_val[v * 8 + j] = gi < values_len ? vin[gi] : 0; }
            }
        }
        if (tid == 0 && 4 * kBlock < nvec) {   // the one possible 1025th vector
            const int v = 4 * kBlock;
            for (int j = 0; j < 8; ++j) { const int64_t gi = e0 + (int64_t)v * 8 + j; s_val[v * 8 + j] = gi < values_len ? vin[gi] : 0; }
        }
        __syncthreads();
        const int t0 = s_tot[0], t1 = s_tot[1], t2 = s_tot[2];
        const int wbase[4] = {0, t0, t0 + t1, t0 + t1 + t2};
        const char* sv = reinterpret_cast<const char*>(s_val);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = i * kBlock + tid;
            if (u >= nu) continue;
            const uint32_t info = s_unit[u];
            const uint32_t mv = info & 0xffu;
            const uint32_t r = (info >> 8) + (uint32_t)(wbase[i] + shift);
            const u32x4 sel = *reinterpret_cast<const u32x4*>(&s_sel[((r & 1u) * 256 + mv) * 4]);
            const uint32_t off = s_off[mv];
            const uint32_t a0 = 2u * r;
            const uint32_t sl[4] = {sel.x, sel.y, sel.z, sel.w};
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t a = (a0 + ((off >> (8 * j)) & 0xffu)) & ~3u;
                const uint32_t lo = *reinterpret_cast<const uint32_t*>(sv + a), hi = *reinterpret_cast<const uint32_t*>(sv + a + 4);
                w[j] = __builtin_amdgcn_perm(hi, lo, sl[j]);
            }
            u32x4* o = reinterpret_cast<u32x4*>(out + row * cols + ((int64_t)u << 3));
            if (NTST) __builtin_nontemporal_store(u32x4{w[0], w[1], w[2], w[3]}, o); else *o = u32x4{w[0], w[1], w[2], w[3]};
        }
        __syncthreads();
    }
}

// traffic-shaped ceiling: same bytes in and out as the decompress (mask dword + value vectors in, dense nt out), no logic
__global__ __launch_bounds__(kBlock) void dec_ceiling(const uint16_t* __restrict__ vin, int64_t values_len, const uint8_t* __restrict__ bitmask,
                                                      const int64_t* __restrict__ row_offsets, int64_t rows, int64_t cols, uint16_t* __restrict__ out) {
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t bcols = cols >> 3;
    const uint32_t md = *reinterpret_cast<const uint32_t*>(bitmask + row * bcols + 4 * tid);
    const int64_t v0 = (row * (values_len >> 3)) / rows;  // this row's share of the value vectors
    const u32x4* g = reinterpret_cast<const u32x4*>(vin) + v0;
    u32x4 a = g[tid], b = g[tid + 256];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = i * kBlock + tid;
        u32x4 w = (i & 1) ? a : b;
        w.x ^= md + i;
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(out + row * cols + ((int64_t)u << 3)));
    }
}

// ---------------------------------------------------------------- host
int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 8192;
    const int64_t rows = N, cols = N, bcols = cols / 8;
    const int NSETS = 8;
    std::vector<uint16_t> hw(rows * cols), hv;
    std::vector<uint8_t> hm(rows * bcols, 0);
    std::vector<int64_t> hro(rows);
    srand(3);
    hv.reserve(rows * cols / 2 + 1024);
    for (int64_t r = 0; r < rows; ++r) {
        hro[r] = (int64_t)hv.size();
        for (int64_t c = 0; c < cols; ++c) {
            uint16_t v = (rand() & 1) ? (uint16_t)((rand() & 0x7fff) | 1) : 0;
            hw[r * cols + c] = v;
            if (v) { hv.push_back(v); hm[r * bcols + (c >> 3)] |= (uint8_t)(1u << (c & 7)); }
        }
    }
    const int64_t nnz = (int64_t)hv.size();
    struct Set { uint16_t *v, *out; uint8_t* m; int64_t* ro; };
    std::vector<Set> sets(NSETS);
    for (auto& s : sets) {
        CK(hipMalloc(&s.v, nnz * 2 + 64)); CK(hipMalloc(&s.out, rows * cols * 2)); CK(hipMalloc(&s.m, rows * bcols)); CK(hipMalloc(&s.ro, rows * 8));
        CK(hipMemcpy(s.v, hv.data(), nnz * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(s.m, hm.data(), rows * bcols, hipMemcpyHostToDevice));
        CK(hipMemcpy(s.ro, hro.data(), rows * 8, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double alg = 2.0 * rows * cols + 2.0 * nnz + rows * cols / 8.0 + 8.0 * rows;
    std::vector<uint16_t> back(rows * cols);
    auto run = [&](const char* name, std::function<void(const Set&)> fn) {
        CK(hipMemset(sets[0].out, 0xff, rows * cols * 2));
        fn(sets[0]); CK(hipDeviceSynchronize()); CK(hipGetLastError());
        CK(hipMemcpy(back.data(), sets[0].out, rows * cols * 2, hipMemcpyDeviceToHost));
        int64_t bad = 0; for (int64_t i = 0; i < rows * cols; ++i) bad += back[i] != hw[i];
        for (int i = 0; i < 8; ++i) fn(sets[i % NSETS]);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 16; ++i) fn(sets[i % NSETS]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        double us = best * 1000.0 / 16;
        printf("%-28s %8.2f us  %7.1f GB/s  (%.1f%% of 8 TB/s)  mismatches=%lld\n", name, us, alg / us / 1e3, alg / us / 1e3 / 80.0, (long long)bad);
    };
    printf("N=%lld nnz=%lld alg=%.0f\n", (long long)N, (long long)nnz, alg);
#define V1(UPL, CAP) run("dec_v1 UPL" #UPL " cap" #CAP, [&](const Set& s) { int64_t nw = rows * ((cols + 256 * UPL * 8 - 1) / (256 * UPL * 8)); int64_t g = CAP > 0 && nw > CAP ? CAP : nw; \
        hipLaunchKernelGGL((dec_v1<UPL>), dim3((unsigned)g), dim3(256), 0, 0, s.v, nnz, s.m, s.ro, rows, cols, s.out); })
#define V2(UPL, CAP) run("dec_v2 UPL" #UPL " cap" #CAP, [&](const Set& s) { int64_t g = CAP > 0 && rows > CAP ? CAP : rows; hipLaunchKernelGGL((dec_v2<UPL>), dim3((unsigned)g), dim3(256), 0, 0, s.v, nnz, s.m, s.ro, rows, cols, s.out); })
    run("ceiling (traffic only)", [&](const Set& s) { hipLaunchKernelGGL(dec_ceiling, dim3((unsigned)rows), dim3(256), 0, 0, s.v, nnz, s.m, s.ro, rows, cols, s.out); });
#define V3(NT, CAP) run("dec_v3 nt" #NT " cap" #CAP, [&](const Set& s) { int64_t g = CAP > 0 && rows > CAP ? CAP : rows; hipLaunchKernelGGL((dec_v3<NT>), dim3((unsigned)g), dim3(256), 0, 0, s.v, nnz, s.m, s.ro, rows, cols, s.out); })
    V3(true, 0); V3(false, 0); V3(true, 2048); V3(true, 4096);
    V2(4, 0); V2(4, 2048); V2(4, 4096);
    V1(4, 0); V1(2, 0); V1(1, 0); V1(4, 2048); V1(2, 2048); V1(1, 2048); V1(2, 4096);
    return 0;
}
