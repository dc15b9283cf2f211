// ct_qparams.hip — min/max observer + calculate_qparams for weight groups.
//
// Reference: quantization/utils/helpers.py:50-137 applied to torch.aminmax over each group of
// `cdiv` consecutive columns of a row (group / channel strategies; a tensor-wide range is the
// same call on the tensor viewed as one row).  Every arithmetic step is rounded to the weight
// dtype D exactly like the eager op sequence:
//   mn = min(mn, 0); mx = max(mx, 0)
//   symmetric : scale = rnd_D(max(|mn|, |mx|) / (range / 2)); zp = 0
//   asymmetric: scale = rnd_D(rnd_D(mx - mn) / range); zp = clamp(rnd_D(qmin - rnd_D(mn / scale)), qmin, qmax)
//   scale == 0 -> eps(D); zp -> clamp to int8, round half-even, cast
// One streaming read of the weight: 16-byte loads, one 8-element unit per lane, groups reduced
// across lanes with wave shuffles (a 128-wide group is 16 lanes).
#include "ct_common.h"
#include "ct_minmax.h"

#include <cstdlib>

namespace ct {

// FLOAT kinds: the scale is float32 for NVFP4 (fp8-representable values), x's dtype otherwise; no zero point (symmetric)
template <int XDT>
__device__ __forceinline__ void emit_qparams_float(MinMax m, int kind, const float* gscale, void* scale_out, int64_t idx) {
    const float s = compute_qparams_float<XDT>(m, kind, kind == QP_NVFP4 ? gscale[0] : 1.0f);
    if (kind == QP_NVFP4) static_cast<float*>(scale_out)[idx] = s;
    else store1<XDT>(scale_out, idx, s);
}

// groups of LPG lanes x Q units x 8 elements (cdiv = 64 * LPG for Q = 8 ...; LPG a power of two <= 64),
// cols % cdiv == 0.  A lane owns Q CONSECUTIVE units (Q x 16 B loads in flight: with one unit per lane
// a CU had only 32 KB outstanding and the kernel sat at 3.3 TB/s) and reduces them locally, the group is
// finished with DPP, and one lane in LPG runs the (divide-heavy, divergent) scale / zero-point math.
template <int XDT, int Q>
__global__ __launch_bounds__(kBlock) void qparams_subwave_kernel(const void* __restrict__ x, int64_t lanes_total, int lpg, int bits, int symmetric,
                                                                 void* __restrict__ scale_out, int8_t* __restrict__ zp_out, int kind,
                                                                 const float* __restrict__ gscale) {
    // lanes_total is a multiple of lpg, and kBlock is a multiple of lpg: groups never straddle waves
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t nloops = (lanes_total + stride - 1) / stride;
    int64_t l = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (int64_t it = 0; it < nloops; ++it, l += stride) {
        MinMax m;
        m.mn = __builtin_inff(); m.mx = -__builtin_inff(); m.nan = 0;
        const bool live = l < lanes_total;
        if (XDT != CT_F32 && symmetric) {
            // symmetric scheme, 16-bit weights: max |x| on the raw pairs (ct_minmax.h)
            if constexpr (XDT != CT_F32) {
                uint32_t acc = 0;
                if (live) {
                    u32x4 r[Q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) r[q] = reinterpret_cast<const u32x4*>(x)[l * Q + q];
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, r[q].x), r[q].y), r[q].z), r[q].w);
                }
                m = absmax_finish<XDT>(absmax_group_reduce(acc, lpg));
            }
        } else {
            if (live) {
                float v[Q][8];
#pragma unroll
                for (int q = 0; q < Q; ++q) load8<XDT>(x, (l * Q + q) << 3, v[q]);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        m.nan |= (v[q][k] != v[q][k]);
                        m.mn = __builtin_fminf(m.mn, v[q][k]);
                        m.mx = __builtin_fmaxf(m.mx, v[q][k]);
                    }
                }
            }
            m = group_reduce(m, lpg);
        }
        if (live && (threadIdx.x & (lpg - 1)) == 0) {
            if (kind == QP_INT) emit_qparams<XDT>(m, bits, symmetric, scale_out, zp_out, l / lpg);
            else emit_qparams_float<XDT>(m, kind, gscale, scale_out, l / lpg);
        }
    }
}

// symmetric schemes on 16-bit weights, groups of at most 64 units: lane = U units ONE BLOCK APART, so that every wave load
// instruction reads 1 KiB contiguous (the 64-bytes-per-lane form above reads 4 strided 16-byte pieces per instruction:
// 25.4 us against 21.3 us in the read-only calibration).  A group is `upg` adjacent lanes of one load; the
// integer abs-max makes the U reductions per lane cheap (one packed max per DPP step).
template <int XDT, int U>
__global__ __launch_bounds__(kBlock) void qparams_absmax_kernel(const u32x4* __restrict__ x, int64_t units, int upg, int bits, void* __restrict__ scale_out,
                                                                int8_t* __restrict__ zp_out, int kind, const float* __restrict__ gscale) {
    const int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x;
    // (round 4: unconditional loads — the index clamped, a unit beyond the tensor masked out of the maximum — so that hipcc's wait-count pass
    // can count them: behind `u < units ? x[u] : 0` every load sat in a branch and the first reduction waited for all U of them, vmcnt(0))
    u32x4 r[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int64_t u = base + (int64_t)i * kBlock;
        r[i] = x[u < units ? u : units - 1];
    }
    uint32_t acc[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const uint32_t live = 0u - (uint32_t)(base + (int64_t)i * kBlock < units);
        acc[i] = absmax_acc(absmax_acc(absmax_acc(absmax_acc(0u, r[i].x), r[i].y), r[i].z), r[i].w) & live;
        acc[i] = absmax_group_reduce(acc[i], upg);
    }
    // Round 3: the scale arithmetic (IEEE divides, the zero-point rounding: ~80 vector instructions) used to run once per load, with
    // one lane in `upg` active — for group 128 four passes per wave with 4 live lanes each, ~320 of the wave's ~400 instructions: the
    // kernel was issue-bound next to a 21.3 us read.  The wave's U * 64 / upg group maxima are first gathered into consecutive lanes
    // (lane L takes load L / gpl, group L % gpl; one ds_bpermute per load) and evaluated in ONE pass.
    const int lane = threadIdx.x & 63;
    const int gpl = 64 / upg;            // groups per load instruction and wave (upg is a power of two <= 64)
    const int ngr = U * gpl;             // groups of this wave: <= 4 * 64
    for (int g0 = 0; g0 < ngr; g0 += 64) {  // one trip unless upg == 1 with U > 1
        const int gi = g0 + lane;           // this lane's group within the wave
        const int li = gi / gpl, lk = gi - li * gpl;
        uint32_t mine = 0;
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint32_t v = (uint32_t)__shfl((int)acc[i], lk * upg, 64);
            mine = li == i ? v : mine;
        }
        const int64_t u = (int64_t)blockIdx.x * kBlock * U + (int64_t)li * kBlock + (threadIdx.x & ~63) + lk * upg;  // first unit of the group
        if (gi < ngr && u < units) {
            const MinMax m = absmax_finish<XDT>(mine);
            if (kind == QP_INT) emit_qparams<XDT>(m, bits, 1, scale_out, zp_out, u / upg);
            else emit_qparams_float<XDT>(m, kind, gscale, scale_out, u / upg);
        }
    }
}

// generic: one wave per group, lanes stride over the group's columns
template <int XDT>
__global__ __launch_bounds__(kBlock) void qparams_wave_kernel(const void* __restrict__ x, int64_t rows, int64_t cols, int64_t cdiv, int bits,
                                                              int symmetric, void* __restrict__ scale_out, int8_t* __restrict__ zp_out, int kind,
                                                              const float* __restrict__ gscale, int vec) {
    const int64_t ngroups = (cols + cdiv - 1) / cdiv;
    const int64_t total = rows * ngroups;
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * kBlock) >> 6;
    for (int64_t gi = wave0; gi < total; gi += nwaves) {
        const int64_t r = gi / ngroups, g = gi - r * ngroups;
        const int64_t c0 = g * cdiv, c1 = (c0 + cdiv < cols) ? c0 + cdiv : cols;
        MinMax m;
        m.mn = __builtin_inff(); m.mx = -__builtin_inff(); m.nan = 0;
        if constexpr (XDT != CT_F32) {
            // Round 6: symmetric schemes on 16-bit weights (channel-wise int8 / FP8 observers, generate_gparam's row maxima): max |x| on the raw pairs
            // (ct_minmax.h) with EIGHT unconditional 16-byte loads in flight per lane — the float path below (four loads behind a bounds test each, five
            // VALU per element) ran these rows at 0.41-0.69 of the HBM peak (profiles/r06_shape_sweep_rtn.txt)
            if (vec && symmetric) {
                const int64_t nu = (c1 - c0) >> 3;
                const u32x4* xin = reinterpret_cast<const u32x4*>(x) + ((r * cols + c0) >> 3);
                uint32_t acc = 0;
                int64_t ub = 0;  // wave-uniform
                for (; nu - ub >= 512; ub += 512) {
                    u32x4 v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = xin[ub + 64 * q + lane];
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, v[q].x), v[q].y), v[q].z), v[q].w);
                }
                // the row's tail: the index is clamped to the row's last unit (a unit read twice does not change a maximum)
                if (nu - ub > 256) {
                    u32x4 v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int64_t u = ub + 64 * q + lane; v[q] = xin[u < nu ? u : nu - 1]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, v[q].x), v[q].y), v[q].z), v[q].w);
                } else if (nu - ub > 0) {
                    u32x4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int64_t u = ub + 64 * q + lane; v[q] = xin[u < nu ? u : nu - 1]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = absmax_acc(absmax_acc(absmax_acc(absmax_acc(acc, v[q].x), v[q].y), v[q].z), v[q].w);
                }
                m = absmax_finish<XDT>(absmax_group_reduce(acc, 64));
                if (lane == 0) {
                    if (kind == QP_INT) emit_qparams<XDT>(m, bits, 1, scale_out, zp_out, gi);
                    else emit_qparams_float<XDT>(m, kind, gscale, scale_out, gi);
                }
                continue;
            }
        }
        if (vec) {
            // 16-byte units, four in flight per lane (a 5632- or 11008-wide row of a channel-wise scheme is not a power of two of
            // units and lands here: with one 2-byte load per lane and trip this kernel ran at 0.8 TB/s)
            const int64_t u0 = (r * cols + c0) >> 3, nu = (c1 - c0) >> 3;
            for (int64_t u = lane; u < nu; u += 256) {
                float v[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (u + 64 * q < nu) load8<XDT>(x, (u0 + u + 64 * q) << 3, v[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (u + 64 * q >= nu) continue;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        m.nan |= (v[q][k] != v[q][k]);
                        m.mn = __builtin_fminf(m.mn, v[q][k]);
                        m.mx = __builtin_fmaxf(m.mx, v[q][k]);
                    }
                }
            }
        } else {
            for (int64_t c = c0 + lane; c < c1; c += 64) {
                const float v = load_as_f<XDT>(x, r * cols + c);
                m.nan |= (v != v);
                m.mn = __builtin_fminf(m.mn, v);
                m.mx = __builtin_fmaxf(m.mx, v);
            }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            MinMax o;
            o.mn = __shfl_xor(m.mn, d, 64);
            o.mx = __shfl_xor(m.mx, d, 64);
            o.nan = __shfl_xor(m.nan, d, 64);
            m = mm_merge(m, o);
        }
        if (lane == 0) {
            if (kind == QP_INT) emit_qparams<XDT>(m, bits, symmetric, scale_out, zp_out, gi);
            else emit_qparams_float<XDT>(m, kind, gscale, scale_out, gi);
        }
    }
}

// generate_gparam's tail (helpers.py:308-337) on the device: the row maxima (kind 5) -> amax -> 448 * 6 / amax in x's dtype as the eager expression
// evaluates it (`float / tensor` = reciprocal, then product: two roundings), float32, non-finite -> 1.  One workgroup: a checkpoint's rows are few.
// (The host composed this from seven tiny tensor ops: 45-65 us of launches behind a 10-26 us reduction — profiles/r06_shape_sweep_rtn.txt.)
template <int XDT>
__global__ __launch_bounds__(kBlock) void gparam_finish_kernel(const void* __restrict__ row_amax, int64_t rows, float* __restrict__ gs_out) {
    __shared__ float s_mx[kBlock / 64];
    __shared__ int s_nan[kBlock / 64];
    float mx = 0.0f;
    int nan = 0;
    for (int64_t base = threadIdx.x; base < rows; base += 8 * kBlock) {  // eight loads in flight per lane (index clamped: a value read twice changes no maximum)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = base + (int64_t)k * kBlock;
            v[k] = load_as_f<XDT>(row_amax, i < rows ? i : rows - 1);  // >= 0, or NaN
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            nan |= (v[k] != v[k]);
            mx = __builtin_fmaxf(mx, v[k]);
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, d, 64));
        nan |= __shfl_xor(nan, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_nan[threadIdx.x >> 6] = nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) { mx = __builtin_fmaxf(mx, s_mx[w]); nan |= s_nan[w]; }
        const float tiny = XDT == CT_F16 ? 0x1p-14f : 0x1p-126f;  // torch.finfo(dtype).tiny
        float amax = nan ? __builtin_nanf("") : mx;
        amax = amax < tiny ? tiny : amax;                          // clamp(min=tiny): a NaN passes through
        const float recip = round_to<XDT>(1.0f / amax);
        const float gs = round_to<XDT>(recip * 2688.0f);           // FP8_E4M3_DATA.max * FP4_E2M1_DATA.max
        gs_out[0] = __builtin_isfinite(gs) ? gs : 1.0f;            // nan_to_num(nan=1, posinf=1, neginf=1)
    }
}

}  // namespace ct

using namespace ct;

extern "C" {

static int minmax_qparams_impl(const void* x, int xdt, int64_t rows, int64_t cols, int64_t cdiv, int bits, int symmetric, void* scale_out, int8_t* zp_out,
                               int kind, const float* gscale, ct_stream_t stream) {
    CT_REQUIRE(is_float_dt(xdt), "weight dtype code %d is not a float type", xdt);
    CT_REQUIRE(bits >= 1 && bits <= 8, "num_bits must be in [1, 8], got %d", bits);
    CT_REQUIRE(rows >= 0 && cols >= 0 && cdiv >= 1, "bad shape");
    if (rows == 0 || cols == 0) return CT_OK;
    const int64_t upg = cdiv / 8;  // units per group
    const bool subwave = (cdiv % 8 == 0) && (cols % cdiv == 0) && upg >= 1 && upg <= 256 && log2_exact(upg) >= 0 && aligned16(x);
    if (subwave && symmetric && xdt != CT_F32 && upg <= 64) {
        const int64_t units = rows * (cols / 8);
        // units per lane: 2 / 4 / 8 measured 25.0 / 24.4 / 27.0 us at 8192^2 (round 2): 4
#define CT_QPA(DT, U) do { const int64_t g = cdiv64(units, (int64_t)kBlock * U); CT_REQUIRE(g < ((int64_t)1 << 31), "tensor too large for one launch"); \
        hipLaunchKernelGGL((qparams_absmax_kernel<DT, U>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), static_cast<const u32x4*>(x), units, (int)upg, bits, scale_out, \
                           zp_out, kind, gscale); } while (0)
        if (xdt == CT_BF16) CT_QPA(CT_BF16, 4);
        else CT_QPA(CT_F16, 4);
#undef CT_QPA
        CT_LAUNCH_CHECK("ct_minmax_qparams[absmax]");
    }
    if (subwave) {
        const int64_t units = rows * (cols / 8);
        const int q = upg >= 4 ? 4 : 1;  // consecutive units per lane
        const int64_t lpg = upg / q, lanes = units / q;
        int64_t g = cdiv64(lanes, kBlock);  // exact grid: many small workgroups stream best
        if (g > ((int64_t)1 << 30)) g = (int64_t)1 << 30;
#define CT_QP(DT) do { if (q == 4) hipLaunchKernelGGL((qparams_subwave_kernel<DT, 4>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), x, lanes, (int)lpg, bits, symmetric, scale_out, zp_out, kind, gscale); \
                       else hipLaunchKernelGGL((qparams_subwave_kernel<DT, 1>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), x, lanes, (int)lpg, bits, symmetric, scale_out, zp_out, kind, gscale); } while (0)
        switch (xdt) {
            case CT_BF16: CT_QP(CT_BF16); break;
            case CT_F16: CT_QP(CT_F16); break;
            default: CT_QP(CT_F32); break;
        }
#undef CT_QP
        CT_LAUNCH_CHECK("ct_minmax_qparams[subwave]");
    }
    const int64_t total = rows * cdiv64(cols, cdiv);
    int64_t g = cdiv64(total, kBlock / 64);
    if (g > kCUs * 32) g = kCUs * 32;
    const int vec = (cols % 8 == 0) && (cdiv % 8 == 0 || cdiv >= cols) && aligned16(x);  // every group is a whole number of 16-byte units
    switch (xdt) {
        case CT_BF16: hipLaunchKernelGGL((qparams_wave_kernel<CT_BF16>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), x, rows, cols, cdiv, bits, symmetric, scale_out, zp_out, kind, gscale, vec); break;
        case CT_F16: hipLaunchKernelGGL((qparams_wave_kernel<CT_F16>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), x, rows, cols, cdiv, bits, symmetric, scale_out, zp_out, kind, gscale, vec); break;
        default: hipLaunchKernelGGL((qparams_wave_kernel<CT_F32>), dim3((unsigned)g), dim3(kBlock), 0, as_stream(stream), x, rows, cols, cdiv, bits, symmetric, scale_out, zp_out, kind, gscale, vec); break;
    }
    CT_LAUNCH_CHECK("ct_minmax_qparams");
}

int ct_minmax_qparams(const void* x, int xdt, int64_t rows, int64_t cols, int64_t cdiv, int bits, int symmetric, void* scale_out, int8_t* zp_out,
                      ct_stream_t stream) {
    return minmax_qparams_impl(x, xdt, rows, cols, cdiv, bits, symmetric, scale_out, zp_out, QP_INT, nullptr, stream);
}

int ct_generate_gparam(const void* x, int xdt, int64_t rows, int64_t cols, void* row_amax, float* global_scale_out, ct_stream_t stream) {
    CT_REQUIRE(row_amax != nullptr && global_scale_out != nullptr, "ct_generate_gparam needs its scratch (rows elements of x's dtype) and its output");
    CT_REQUIRE(rows >= 1 && cols >= 1, "generate_gparam of an empty tensor");
    int rc = minmax_qparams_impl(x, xdt, rows, cols, cols, 8, 1, row_amax, nullptr, QP_AMAX, nullptr, stream);
    if (rc) return rc;
    switch (xdt) {
        case CT_BF16: hipLaunchKernelGGL((gparam_finish_kernel<CT_BF16>), dim3(1), dim3(kBlock), 0, as_stream(stream), row_amax, rows, global_scale_out); break;
        case CT_F16: hipLaunchKernelGGL((gparam_finish_kernel<CT_F16>), dim3(1), dim3(kBlock), 0, as_stream(stream), row_amax, rows, global_scale_out); break;
        default: hipLaunchKernelGGL((gparam_finish_kernel<CT_F32>), dim3(1), dim3(kBlock), 0, as_stream(stream), row_amax, rows, global_scale_out); break;
    }
    CT_LAUNCH_CHECK("ct_generate_gparam");
}

int ct_minmax_qparams_float(const void* x, int xdt, int64_t rows, int64_t cols, int64_t cdiv, int kind, const float* global_scale, void* scale_out,
                            ct_stream_t stream) {
    CT_REQUIRE(kind >= QP_FP8 && kind <= QP_AMAX, "float qparams kind must be 1 (fp8), 2 (nvfp4), 3 (mxfp4), 4 (mxfp8) or 5 (amax), got %d", kind);
    CT_REQUIRE(kind != QP_NVFP4 || global_scale != nullptr, "nvfp4 scales need the global scale");
    return minmax_qparams_impl(x, xdt, rows, cols, cdiv, 8, 1, scale_out, nullptr, kind, global_scale, stream);
}

}  // extern "C"
