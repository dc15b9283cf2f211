/*
 * ct_hip.h — C ABI of libct_hip.so: the MI355X (gfx950) implementation of the
 * compressed-tensors compress/decompress hot path.
 *
 * Every entry point takes plain device pointers + sizes + a HIP stream (as void*), launches
 * asynchronously on that stream, performs no host synchronisation and no allocation, and
 * returns a ct_status.  ct_last_error() returns a thread-local message for the last
 * non-OK status.  The functions are re-entrant (no global scratch).
 *
 * Each declaration names the reference interface it replaces; paths are relative to
 * /root/reference/src/compressed_tensors/.  The Python host in compressed_tensors_amd/
 * binds these with ctypes (compressed_tensors_amd/_lib.py); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Tensors are row-major and contiguous unless a stride parameter says otherwise.
 */
#ifndef CT_HIP_H
#define CT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ct_stream_t; /* hipStream_t */

/* element type codes (shared with the oracle) */
enum ct_dtype {
    CT_F32 = 0,
    CT_F16 = 1,
    CT_BF16 = 2,
    CT_I8 = 3,
    CT_I32 = 4,
    CT_U8 = 5,
    CT_I16 = 6,
    CT_I64 = 7,
    CT_F8E4M3 = 8 /* float8_e4m3fn (OCP: max 448, 0x7f / 0xff = NaN, no inf) */
};

enum ct_status {
    CT_OK = 0,
    CT_ERR_INVALID_ARG = 1, /* maps to ValueError on the Python side */
    CT_ERR_UNSUPPORTED = 2, /* maps to NotImplementedError                */
    CT_ERR_HIP = 3          /* maps to RuntimeError (message has the HIP error string) */
};

const char* ct_last_error(void);
int ct_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Host mailbox — the only entries that wait.  Two points of the reference's interface hand a
 * DEVICE result to the HOST before the call may return: the number of kept values of the
 * sparse-bitmask codec (`tensor[mask]` sizes its result on the host; restated S1 over
 * utils/helpers.py:306-343) and the 2:4 structure verdict of the marlin-24 codec
 * (`tensor_follows_mask_structure(...)` raises from the call, utils/helpers.py:87-109).
 * With torch that is a D2H copy + a synchronisation per call (`.item()`, `.cpu()`: 20-30 us).
 * Here the kernels write the word straight into pinned, device-mapped host memory (one
 * system-scope store) and the host spins on it.
 *   ct_mailbox_alloc     `bytes` of zeroed pinned host memory; *dev_ptr is what the kernels get
 *   ct_mailbox_wait_i64  returns in *value the 64-bit word at host_word as soon as it differs
 *                        from `pending` (the host writes `pending` before the launch); if the
 *                        stream drains first, whatever the word holds then
 *   ct_stream_wait       returns when every launch queued on `stream` has completed (spins on
 *                        hipStreamQuery: ~1 us after completion, no interrupt, no copy) */
int ct_mailbox_alloc(int64_t bytes, void** host_ptr, void** dev_ptr);
int ct_mailbox_free(void* host_ptr);
int ct_mailbox_wait_i64(const int64_t* host_word, int64_t pending, ct_stream_t stream, int64_t* value);
int ct_stream_wait(ct_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Scale / zero-point addressing used by every quantization entry point:
 *     idx(r, c) = (r / rdiv) * scale_cols + (col_group ? col_group[c] : c / cdiv)
 * tensor:  rdiv = rows, cdiv = cols, scale_cols = 1
 * channel: rdiv = 1,    cdiv = cols, scale_cols = 1           (token: same)
 * group:   rdiv = 1 (or rows if the scale has a single row), cdiv = group_size,
 *          scale_cols = cols / group_size
 * block:   rdiv = block_h, cdiv = block_w, scale_cols = ceil(cols / block_w)
 * col_group (nullable, int32[cols], device) carries activation ordering:
 *          argsort(argsort(g_idx)) / group_size  (quantization/lifecycle/forward_helpers.py:147-175)
 * tdt is the torch result dtype of `x / scale` — every intermediate of the reference's
 * eager op sequence is rounded to it (forward_helpers.py:523-546, quant_args.py:460-496).
 * zp may be NULL (no zero point); zdt in {CT_I8, CT_I32, CT_F32, CT_F16, CT_BF16}.
 * ------------------------------------------------------------------------------------ */

/* pack_to_int32(value, num_bits, packed_dim=1)   compressors/pack_quantized/helpers.py:20-101
 * q: int8 (rows, cols); out: int32 (rows, out_row_stride >= ceil(cols*bits/32)) */
int ct_pack_int32(const int8_t* q, int64_t rows, int64_t cols, int bits, int32_t* out,
                  int64_t out_row_stride, ct_stream_t stream);

/* unpack_from_int32(value, num_bits, shape, packed_dim=1)   helpers.py:104-180
 * p: int32 (rows, words) with row stride p_row_stride; out: int8 (rows, cols) */
int ct_unpack_int32(const int32_t* p, int64_t rows, int64_t words, int64_t p_row_stride,
                    int64_t cols, int bits, int8_t* out, ct_stream_t stream);

/* pack_to_int32(zp, num_bits, packed_dim=0).contiguous()   compressors/pack_quantized/base.py:107-110
 * q: int8 (rows, cols) packed along ROWS; out: int32 (ceil(rows*bits/32), cols) contiguous */
int ct_pack_int32_dim0(const int8_t* q, int64_t rows, int64_t cols, int bits, int32_t* out,
                       ct_stream_t stream);

/* unpack_from_int32(zp, num_bits, shape, packed_dim=0)   base.py:147-153
 * p: int32 (words, cols) contiguous; out: int8 (rows, cols) */
int ct_unpack_int32_dim0(const int32_t* p, int64_t words, int64_t cols, int64_t rows, int bits,
                         int8_t* out, ct_stream_t stream);

/* quantize(x, scale, zero_point, args, dtype)   quantization/lifecycle/forward.py:36-73
 * odt in {CT_I8, CT_I32, CT_F32, CT_F16, CT_BF16} */
int ct_quantize(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt,
                int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                const int32_t* col_group, int bits, int tdt, void* out, int odt, ct_stream_t stream);

/* dequantize(x_q, scale, zero_point, ...)   forward.py:76-145; forward_helpers.py:549-572
 * xq: qdt in {CT_I8, CT_I32, CT_F8E4M3, float types}; arithmetic in sdt; odt in {CT_F32, CT_F16, CT_BF16} */
int ct_dequantize(const void* xq, int qdt, const void* scale, int sdt, const void* zp, int zdt,
                  int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                  const int32_t* col_group, void* out, int odt, ct_stream_t stream);

/* fake_quantize(x, scale, zero_point, args)   forward.py:148-181; forward_helpers.py:180-215 */
int ct_fake_quantize(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt,
                     int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                     const int32_t* col_group, int bits, int tdt, void* out, int odt,
                     ct_stream_t stream);

/* The same three for FLOAT 8-bit args (QuantizationType.FLOAT, num_bits 8; quant_args.py:463-486,
 * utils/helpers.py:212-214): clamp to +-448 and round to float8_e4m3fn instead of rint.  odt of
 * ct_quantize_fp8 in {CT_F8E4M3, CT_F32, CT_F16, CT_BF16}; zero points may be CT_F8E4M3 too.  Dequantize is
 * ct_dequantize with qdt = CT_F8E4M3.  These are the weight paths of float-quantized / naive-quantized(float)
 * (compressors/naive_quantized/base.py:48-126) and mxfp8-quantized (compressors/mxfp8/base.py:47-101). */
int ct_quantize_fp8(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt,
                    int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                    const int32_t* col_group, int tdt, void* out, int odt, ct_stream_t stream);
int ct_fake_quantize_fp8(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt,
                         int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                         const int32_t* col_group, int tdt, void* out, int odt, ct_stream_t stream);

/* FLOAT 4-bit args (E2M1; quant_args.py:463-486 -> cast_to_fp4, range +-6 utils/helpers.py:215-217), optionally
 * under a global scale (tensor_group strategy: effective scale = fl32(scale / global_scale[0]), the quotient and the
 * dequantize arithmetic are float32 then, forward_helpers.py:535-538,560-563; tdt must be CT_F32).  Values come back
 * in a float dtype (upstream has no 4-bit storage dtype; the packed weight path is ct_fp4_quant_pack).
 * global_scale: device float32[1] or NULL. */
int ct_quantize_fp4(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt,
                    int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                    const int32_t* col_group, const float* global_scale, int tdt, void* out, int odt,
                    ct_stream_t stream);
int ct_fake_quantize_fp4(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt,
                         int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                         const int32_t* col_group, const float* global_scale, int tdt, void* out, int odt,
                         ct_stream_t stream);
/* dequantize(x_q, scale, zero_point, global_scale=...): arithmetic in float32 */
int ct_dequantize_gs(const void* xq, int qdt, const void* scale, int sdt, const void* zp, int zdt,
                     int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                     const int32_t* col_group, const float* global_scale, void* out, int odt,
                     ct_stream_t stream);

/* Fused PackedQuantizationCompressor.compress weight path: quantize(dtype=int8) followed by
 * pack_to_int32, without the int8 intermediate.   compressors/pack_quantized/base.py:96-104
 * packed: int32 (rows, ceil(cols*bits/32)) contiguous */
int ct_quant_pack(const void* x, int xdt, const void* scale, int sdt, const void* zp, int zdt,
                  int64_t rows, int64_t cols, int64_t rdiv, int64_t cdiv, int64_t scale_cols,
                  const int32_t* col_group, int bits, int tdt, int32_t* packed, ct_stream_t stream);

/* Round-to-nearest W4 compress in one pass: the min-max observer + calculate_qparams
 * (quantization/utils/helpers.py:50-137) and PackedQuantizationCompressor's weight path
 * (compressors/pack_quantized/base.py:96-104) fused: x is read once, packed int32 (rows, cols/8), scale
 * (rows, cols/group) in x's dtype and zero point int8 (rows, cols/group) come out.  Bit-identical to
 * ct_minmax_qparams followed by ct_quant_pack.  group = 32 * 2^k <= 2048, cols % group == 0, int4. */
int ct_rtn_quant_pack_w4(const void* x, int xdt, int64_t rows, int64_t cols, int64_t group, int symmetric,
                         int32_t* packed, void* scale_out, int8_t* zp_out, ct_stream_t stream);

/* Channel-wise 8-bit round-to-nearest in one pass (W8A8 int8 / FP8 weights): per-row min-max observer +
 * calculate_qparams (helpers.py:50-137) + quantize(dtype = int8 resp. float8_e4m3fn) (forward.py:36-73).  out: one
 * byte per element; scale_out (rows, 1) in x's dtype; zp_out int8 (rows, 1), required when !symmetric (int8 only).
 * cols % 8 == 0, cols <= 16384.  Bit-identical to ct_minmax_qparams(_float) followed by ct_quantize(_fp8). */
int ct_rtn_quant_channel8(const void* x, int xdt, int64_t rows, int64_t cols, int fp8, int symmetric, void* out,
                          void* scale_out, int8_t* zp_out, ct_stream_t stream);

/* Fused PackedQuantizationCompressor.decompress weight path: unpack_from_int32 followed by
 * dequantize.   base.py:155-161.  zp is the UNPACKED zero point (int8) or NULL. */
int ct_unpack_dequant(const int32_t* packed, int64_t rows, int64_t words, int64_t cols, int bits,
                      const void* scale, int sdt, const void* zp, int zdt, int64_t rdiv,
                      int64_t cdiv, int64_t scale_cols, const int32_t* col_group, void* out, int odt,
                      ct_stream_t stream);

/* Activation ordering (quantization/lifecycle/forward_helpers.py:147-175): the `col_group` table the entries above take, from a module's `weight_g_idx` —
 * col_group[c] = (position of column c in the stable sort of g_idx) / group_size, or c / group_size while g_idx still holds a -1 (not initialised; the
 * reference tests `-1 in g_idx` on the host, here the choice is made on the device: no synchronisation).  `mode_word`: one int32 that receives what was
 * found (0: a -1 left, 1: balanced — every group exactly group_size columns, the usual case: no sort at all —, 2: the general stable rank).  One small launch. */
int ct_gidx_col_group(const int32_t* g_idx, int64_t cols, int64_t group_size, int32_t* col_group, int32_t* mode_word, ct_stream_t stream);

/* Batched W4A16 (int4, 16-bit weights and scales of one dtype, group or channel scales, int8 zero
 * points): one launch for a whole table of tensors — the per-module loop of
 * ModelCompressor.compress_model / decompress_model (compressors/model_compressors/
 * model_compressor.py:167-169,196-198) without a launch per module.
 * The caller fills src / scale / zp / dst / rows / cols / group of every item, calls
 * ct_w4_batch_plan on the HOST copy (fills the derived fields, returns the workgroup count or -1),
 * copies the table to the device and launches.  direction: 0 = compress (src = weights,
 * dst = packed int32), 1 = decompress (src = packed, dst = weights). */
typedef struct ct_w4_item {
    const void* src;
    const void* scale;
    const void* zp; /* int8, shape of scale, or NULL (with zp_packed on the decompress side: an OUTPUT, see there) */
    void* dst;
    int64_t rows, cols, group; /* group <= 0 or >= cols: one scale per row */
    int64_t first_block;       /* derived */
    int64_t units;             /* derived */
    int32_t upg_shift, upg;    /* derived */
    /* W4 batches only (ct_zp4_* / ct_q8_* ignore it; NULL = none).  The zero points of an asymmetric scheme in their STORED form,
     * int32 (ceil(rows * 4 / 32), cols / group) = pack_to_int32(zp, 4, packed_dim=0) (compressors/pack_quantized/base.py:107-110),
     * handled by the SAME launch as the weights (round 6):
     *   compress   (direction 0): OUTPUT.  `zp` (required) is quantized against as always, and tail workgroups of the item's block range
     *              also write its packed form here.  Any item the batch takes may carry it.
     *   decompress (direction 1): INPUT.  The kernel takes every zero point straight from the packed words — no unpacked copy has to
     *              exist first — and, when `zp` is non-NULL, tail workgroups also WRITE the unpacked int8 (rows, cols / group) zero
     *              points there (what decompress stores back into the state dict, base.py:147-153).  Needs group == 128,
     *              cols % 512 == 0, a 16-byte aligned zp_packed and an 8-byte aligned scale (ct_w4_batch_plan refuses the item
     *              otherwise: unpack with ct_zp4_pack_dim0_batch first and pass `zp`). */
    void* zp_packed;
    int64_t main_blocks;       /* derived: the item's blocks in front of its zero-point tail */
    uint32_t g_magic;          /* derived: n / (cols / group) == (n * g_magic) >> g_shift for n < 2^31 */
    int32_t g_shift;           /* derived */
} ct_w4_item;                  /* 13 64-bit words */
int64_t ct_w4_batch_plan(ct_w4_item* items_host, int n, int direction);
int ct_quant_pack_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, ct_stream_t stream);
int ct_unpack_dequant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, ct_stream_t stream);

/* One asymmetric W4A16 tensor, zero points in their stored (packed) form, ONE launch per direction — the single-module
 * PackedQuantizationCompressor.compress / .decompress of an asymmetric scheme (base.py:96-110 resp. 147-161) without the separate
 * pack_to_int32(zp, 4, packed_dim=0) / unpack_from_int32(..., packed_dim=0) launch.  The kernels are the batch kernels on a table of
 * one item handed over by value (no table in device memory, no plan call): same bits as the batch and as ct_quant_pack +
 * ct_pack_int32_dim0 resp. ct_unpack_int32_dim0 + ct_unpack_dequant.
 *   ct_quant_pack_w4_zp      x (rows, cols) bf16 / fp16, scale (rows, cols / group) of x's dtype, zp int8 (rows, cols / group) ->
 *                            packed int32 (rows, cols / 8) and zp_packed int32 (ceil(rows / 8), cols / group).
 *                            cols % 32 == 0, group % 32 == 0, cols % group == 0 (group <= 0 or >= cols: channel-wise).
 *   ct_unpack_dequant_w4_zp  packed + scale + zp_packed -> out (rows, cols) of the scale's dtype sdt and, if zp_out != NULL, the
 *                            unpacked int8 zero points.  group == 128 and cols % 512 == 0 (CT_ERR_UNSUPPORTED otherwise: unpack the
 *                            zero points with ct_unpack_int32_dim0 and call ct_unpack_dequant). */
int ct_quant_pack_w4_zp(const void* x, int xdt, const void* scale, const int8_t* zp, int64_t rows, int64_t cols, int64_t group,
                        int32_t* packed, int32_t* zp_packed, ct_stream_t stream);
int ct_unpack_dequant_w4_zp(const int32_t* packed, const void* scale, int sdt, const int32_t* zp_packed, int64_t rows, int64_t cols,
                            int64_t group, void* out, int8_t* zp_out, ct_stream_t stream);

/* Batched 4-bit zero-point packing along rows: pack_to_int32(zp, 4, packed_dim=0) / unpack_from_int32(..., packed_dim=0) of
 * PackedQuantizationCompressor (compressors/pack_quantized/base.py:107-110,147-153) for the zero points of many asymmetric
 * modules in one launch.  Items: src / dst + rows, cols of the UNPACKED int8 matrix (the other fields are ignored);
 * direction 0: int8 (rows, cols) -> int32 (ceil(rows * 4 / 32), cols); 1: the inverse.  Plan on the host copy first. */
int64_t ct_zp4_batch_plan(ct_w4_item* items_host, int n);
int ct_zp4_pack_dim0_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int direction, ct_stream_t stream);

/* Batched 8-bit codecs (Naive / Int / FloatQuantizationCompressor.compress / decompress, compressors/naive_quantized/
 * base.py:48-126, looped per module by model_compressor.py:167-169,196-198): quantize to int8 (num_bits <= 8, clamped to the
 * num_bits range) or float8_e4m3fn, and the inverse, for a whole table of 16-bit tensors in one launch.  Same table type and
 * protocol as the W4 batch; `group` counts the consecutive elements that share one scale: 0 or cols (one per row), a
 * divisor of cols, or >= rows * cols (one per tensor); a NEGATIVE group is the block strategy (forward.py:198-216, the FP8-block
 * checkpoints): -group = (rows per block << 24) | columns per block, both powers of two, the width >= 16 and a divisor of cols,
 * scale / zero point of shape (ceil(rows / block rows), cols / block columns).  Needs cols % 16 == 0 and group % 16 == 0; zero points int8 or NULL —
 * `fp8` = 2 (round 6): float8 codes whose zero points are float8_e4m3fn bytes (what a calibrated FLOAT scheme carries; 1: int8 or none);
 * `fp8` = 3 (round 6): int8 codes stored + 128, four to an int32 word — the 8-bit words of pack_to_int32 (pack_quantized/helpers.py:39-75;
 * PackedQuantizationCompressor with num_bits = 8, e.g. the W8A16 preset): dst / src = int32 (rows, cols / 4), cols % 32 == 0.
 * direction: 0 = quantize (src = weights, dst = codes), 1 = dequantize (src = codes, dst = weights). */
int64_t ct_q8_batch_plan(ct_w4_item* items_host, int n, int direction);
int ct_q8_quant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, int fp8, int bits,
                      ct_stream_t stream);
int ct_q8_dequant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int dt, int fp8, ct_stream_t stream);

/* Min/max observer + calculate_qparams for weight groups (rows x ceil(cols/cdiv) groups of
 * cdiv consecutive columns).   quantization/utils/helpers.py:50-137
 * scale_out has x's dtype; zp_out is int8 (may be NULL for symmetric). */
int ct_minmax_qparams(const void* x, int xdt, int64_t rows, int64_t cols, int64_t cdiv, int bits,
                      int symmetric, void* scale_out, int8_t* zp_out, ct_stream_t stream);

/* The same reduction for the symmetric FLOAT schemes (helpers.py:50-137, mxfp_utils.py:37-143); amax = max(|min(mn, 0)|,
 * |max(mx, 0)|) of each group:
 *   kind 1 FP8    scale = rnd_X(amax / 448), zero -> eps(X)                                     scale_out: x's dtype
 *   kind 2 NVFP4  scale = float8_e4m3fn(clamp(global_scale * rnd_X(amax / 6), +-448)), 0 -> 1/8  scale_out: float32
 *   kind 3 MXFP4 / 4 MXFP8 (group 32): amax's significand rounded at a quarter and masked off, E8M0 exponent
 *          127 + log2 - 2 (resp. - 8) through uint8, scale = 2^(e - 127) in x's dtype, zero -> 1 scale_out: x's dtype
 *   kind 5 the raw amax of each group in x's dtype (the input of generate_gparam, helpers.py:308-337)
 * Zero points of these schemes are all-zero tensors of the scheme's zp_dtype (made by the host). */
int ct_minmax_qparams_float(const void* x, int xdt, int64_t rows, int64_t cols, int64_t cdiv, int kind,
                            const float* global_scale, void* scale_out, ct_stream_t stream);

/* generate_gparam of a whole weight (quantization/utils/helpers.py:308-337, the NVFP4 global scale): amax = max |x| (NaN if any
 * element is), clamped from below to finfo(x dtype).tiny; global_scale = rnd_X(rnd_X(1 / amax) * 2688) as float32 — `float / tensor`
 * is evaluated by torch as reciprocal times float, two roundings to x's dtype; a non-finite result becomes 1.  Two launches: the
 * row maxima (kind 5 above) into `row_amax` (scratch: rows elements of x's dtype), then one workgroup.  global_scale_out: device float32[1]. */
int ct_generate_gparam(const void* x, int xdt, int64_t rows, int64_t cols, void* row_amax, float* global_scale_out,
                       ct_stream_t stream);

/* ---------------------------------------------------------------------------- FP4 (E2M1) codecs
 * nvfp4-pack-quantized / mxfp4-pack-quantized weight paths (compressors/nvfp4/base.py:68-139,
 * mxfp4/base.py:27-65): quantize(x, scale, global_scale) -> cast_to_fp4 -> pack_fp4_to_uint8 fused, and the
 * inverse.  group: 16 (nvfp4) or 32 (mxfp4); cols % group == 0.  global_scale: device float32[1] or NULL.
 * packed: uint8 (rows, cols/2), element 2i in the low nibble of byte i.
 * compress: scale is the float scale tensor (rows, cols/group) the reference passes to quantize(); x: bf16 / fp16 weights (the lean kernels) or
 * float32 ones (round 6: a plain one-unit-per-lane kernel, IEEE float32 quotient).
 * decompress: scale_kind 0 = float tensor of dtype sdt, 1 = the stored fp8-e4m3fn bytes (nvfp4), 2 = the stored
 * E8M0 exponent bytes (mxfp4); odt in {CT_BF16, CT_F16} (the reference always produces bf16). */
int ct_fp4_quant_pack(const void* x, int xdt, const void* scale, int sdt, const float* global_scale,
                      int64_t rows, int64_t cols, int64_t group, uint8_t* packed, ct_stream_t stream);
int ct_fp4_unpack_dequant(const uint8_t* packed, int64_t rows, int64_t cols, const void* scale, int scale_kind,
                          int sdt, const float* global_scale, int64_t group, void* out, int odt,
                          ct_stream_t stream);
/* The same two launches with the scale tensors of the compressors' state dicts written by the kernel (round 6: the class calls composed them from 1-4
 * tensor ops per module — 5-20 us of launches beside a 10-30 us kernel):
 *   ct_fp4_quant_pack_stored: also writes the STORED scale — NVFP4: `scale.to(float8_e4m3fn)` bytes (rows, cols/16) (nvfp4/base.py:96-100; torch's
 *     conversion: RNE, a magnitude rounding beyond 448 or a NaN -> 0x7f | sign); MXFP4: `compress_mx_scale(scale, uint8)` codes (rows, cols/32)
 *     (mx_utils.py:18-31), read from `mx_code_table`, a device uint8[65536] the caller fills ONCE per scale dtype with that expression evaluated over
 *     every 16-bit pattern (so the codes are the reference's for every input, log2's rounding in the scale dtype included); NULL for NVFP4.
 *     CT_ERR_UNSUPPORTED (nothing launched) unless rows * cols % 32 == 0, the scale is 8-byte aligned and, for MXFP4, 16-bit with a table.
 *   ct_fp4_unpack_dequant_scale: also writes the decompressed scale as bfloat16 (rows, cols/group) — `scale.to(bfloat16)` of the float8 bytes
 *     (nvfp4/base.py:133-137) resp. `decompress_mx_scale` = 2 ** (code - 127) (mx_utils.py:34-44). */
int ct_fp4_quant_pack_stored(const void* x, int xdt, const void* scale, int sdt, const float* global_scale,
                             int64_t rows, int64_t cols, int64_t group, uint8_t* packed, uint8_t* scale_stored,
                             const uint8_t* mx_code_table, ct_stream_t stream);
int ct_fp4_unpack_dequant_scale(const uint8_t* packed, int64_t rows, int64_t cols, const void* scale, int scale_kind,
                                int sdt, const float* global_scale, int64_t group, void* out, int odt,
                                void* scale_bf16_out, ct_stream_t stream);
/* Tables of FP4 tensors: ONE launch per direction for the modules of an NVFP4 / MXFP4 checkpoint (the per-module loop of ModelCompressor.compress_model /
 * decompress_model, model_compressor.py:167-169,196-198, over NVFP4PackedCompressor / MXFP4PackedCompressor, nvfp4/base.py:68-139, mxfp4/base.py:27-65) —
 * ct_fp4_quant_pack_stored / ct_fp4_unpack_dequant_scale for every item.  The table is `struct ct_w4_item` read this way: src / dst = weights and packed
 * bytes in the direction's order, scale = the float scale (compress) resp. the stored float8 / E8M0 bytes (decompress), zp = the item's global scale (one
 * float32 on the device; group 16) or NULL (group 32), group = 16 or 32 (one value per table), zp_packed = the stored-scale output (compress: float8 bytes
 * resp. E8M0 codes, (rows, cols / group)) resp. the bfloat16 scale output (decompress).  ct_fp4_batch_plan fills the derived fields on the HOST copy and
 * returns the workgroup count, or -1 (error set) for an item outside the layout: rows * cols % 32 == 0, cols % group == 0, 16-byte aligned weights / packed
 * bytes (decompress: 4-byte aligned packed bytes), 8-byte aligned float scale.  xdt / sdt: the weights' and the float scales' dtype (one per table; MXFP4:
 * 16-bit scales and `mx_code_table` as in ct_fp4_quant_pack_stored).  odt: bfloat16 (what the reference returns) or float16. */
int64_t ct_fp4_batch_plan(ct_w4_item* items_host, int n, int direction);
int ct_fp4_quant_pack_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int xdt, int sdt, int group,
                            const uint8_t* mx_code_table, ct_stream_t stream);
int ct_fp4_unpack_dequant_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int group, int odt, ct_stream_t stream);
/* compress_mx_scale / decompress_mx_scale (compressors/mx_utils.py:18-44) on their own — the MXFP8 codec's scale conversions and upstream's helpers:
 * codes_out[i] = code_table[bits of scale[i]] (16-bit scales; the table as above) resp. scale_bf16_out[i] = 2 ** (codes[i] - 127) as bfloat16
 * (code 0: the bfloat16 subnormal 2^-127, code 255: inf). */
int ct_mx_scale_compress(const void* scale, int sdt, int64_t n, const uint8_t* code_table, uint8_t* codes_out, ct_stream_t stream);
/* ... for a table of scale tensors in one launch (the MXFP8 modules of a checkpoint: ModelCompressor's loop over MXFP8QuantizationCompressor.compress /
 * .decompress, compressors/mxfp8/base.py:47-101; the weights ride ct_q8_quant_batch / ct_q8_dequant_batch with groups of 32).  Items: src, dst and `rows` = the
 * element count (the other fields are ignored); direction 0: 16-bit scales (dtype `sdt`, one per table) -> uint8 codes through `code_table`; 1: codes ->
 * bfloat16.  Plan on the host copy first. */
int64_t ct_mx_scale_batch_plan(ct_w4_item* items_host, int n);
int ct_mx_scale_batch(const ct_w4_item* items_dev, int n, int64_t total_blocks, int direction, int sdt, const uint8_t* code_table, ct_stream_t stream);
int ct_mx_scale_decompress(const uint8_t* codes, int64_t n, void* scale_bf16_out, ct_stream_t stream);

/* Round-to-nearest MXFP4 in one pass: per 32-element group the min-max observer, calculate_qparams' MX branch
 * (helpers.py:50-137, mxfp_utils.py:118-143), quantize -> cast_to_fp4 -> pack (nvfp4/base.py:88-95) and
 * compress_mx_scale (mx_utils.py:18-31).  packed: uint8 (rows, cols/2); scale_e8m0: uint8 (rows, cols/32), the
 * stored form of the scale; scale_out (nullable): the float scale (x's dtype) calculate_qparams would return.
 * Bit-identical to ct_minmax_qparams_float(kind 3) + ct_fp4_quant_pack + the E8M0 encoding. */
int ct_rtn_mxfp4_quant_pack(const void* x, int xdt, int64_t rows, int64_t cols, uint8_t* packed,
                            uint8_t* scale_e8m0, void* scale_out, ct_stream_t stream);
/* The NVFP4 counterpart (groups of 16 under `global_scale`, device float32[1], e.g. generate_gparam of the weight):
 * scale_f8: float8_e4m3fn bytes (rows, cols/16), the stored form; scale_out (nullable): the float32 scales. cols % 32 == 0. */
int ct_rtn_nvfp4_quant_pack(const void* x, int xdt, int64_t rows, int64_t cols, const float* global_scale,
                            uint8_t* packed, uint8_t* scale_f8, float* scale_out, ct_stream_t stream);

/* The stand-alone primitives the reference exposes as ImplBackend entry points (utils/impl_backend.py:50-79):
 * cast_to_fp4 (quantization/utils/fp4_utils.py:77-98): n float elements -> the nearest E2M1 value in the same
 *   dtype; ties as upstream's thresholds, -0.0 -> +0.0, negative-rounds-to-zero -> -0.0, NaN -> NaN.
 * pack_fp4_to_uint8 (compressors/nvfp4/helpers.py:108-150): n (even) E2M1-valued elements -> n / 2 bytes,
 *   bit 3 of a nibble = the IEEE sign bit.  Values off the E2M1 grid are rounded to it (upstream leaves them
 *   undefined).
 * unpack_fp4_from_uint8 (helpers.py:153-193): n / 2 bytes -> n elements of dtype odt; code 8 -> -0.0. */
int ct_fp4_cast(const void* x, int xdt, void* out, int64_t n, ct_stream_t stream);
int ct_fp4_pack(const void* x, int xdt, uint8_t* packed, int64_t n, ct_stream_t stream);
int ct_fp4_unpack(const uint8_t* packed, int64_t n, void* out, int odt, ct_stream_t stream);

/* Proof obligation of ct_fp4_quant_pack's reciprocal quotient (ct_fp4.hip): counts the (x, s) pairs, s = 1.m for
 * every 23-bit mantissa m in [m_lo, m_hi) and x = every mantissa of dtype xdt (CT_BF16 / CT_F16) in [1, 2), for which
 * it differs from the IEEE fp32 quotient.  *mismatches (device uint64) must come back 0. */
int ct_selftest_fp4_div(int xdt, uint32_t m_lo, uint32_t m_hi, unsigned long long* mismatches, ct_stream_t stream);

/* ---------------------------------------------------------------------------- sparse codecs
 * The compressor classes for these formats were removed from the reference snapshot
 * (compressors/base.py:43-44); the formats and primitives remain: config/base.py:17-18,23,
 * utils/helpers.py:306-343 (pack_bitmasks / unpack_bitmasks, numpy little-endian packbits),
 * utils/semi_structured_conversions.py:33-298, utils/permutations_24.py:20-53.
 * `dt` describes the element for the `!= 0` test and its size; values are copied bit-for-bit.
 */

/* pack_bitmasks(mask)   utils/helpers.py:306-318.  mask: uint8/bool (rows, cols) */
int ct_pack_bitmasks(const uint8_t* mask, int64_t rows, int64_t cols, uint8_t* out, ct_stream_t stream);
/* unpack_bitmasks(packed, shape)   utils/helpers.py:321-343 */
int ct_unpack_bitmasks(const uint8_t* packed, int64_t rows, int64_t cols, uint8_t* mask, ct_stream_t stream);

/* sparse-bitmask compress, pass 1: bitmask = pack_bitmasks(x != 0), row_counts[r] = nnz of row r */
int ct_bitmask_count(const void* x, int dt, int64_t rows, int64_t cols, uint8_t* bitmask,
                     int64_t* row_counts, ct_stream_t stream);
/* exclusive scan of int64 counts -> row_offsets (n) and the grand total (total[0]) */
int ct_exclusive_scan_i64(const int64_t* counts, int64_t n, int64_t* offsets, int64_t* total,
                          ct_stream_t stream);
/* sparse-bitmask compress, pass 2: values[row_offsets[r] + rank] = x[r, c] for non-zero x */
int ct_bitmask_scatter(const void* x, int dt, int64_t rows, int64_t cols, const int64_t* row_offsets,
                       void* values, ct_stream_t stream);
/* sparse-bitmask compress, fused form: bitmask, row_offsets, values and total[0] = nnz with no host
 * round trip.  16- and 32-bit payloads with cols % 8 == 0 (a 32-bit element travels as its two
 * 16-bit halves; the bitmask, the row offsets and the total are per ELEMENT): ONE pass over x — every workgroup keeps its share
 * of the tensor in registers, compacts it while the loads land, publishes its count as one 64-bit
 * word and stores once it has the counts of the workgroups before it; a count that does not arrive
 * within 2 ms is recomputed by the waiting workgroup, so the call cannot fail or deadlock.
 * CT_BITMASK_RESIDENT=0 selects count + scatter (x read twice, no inter-workgroup waiting).
 * 8-bit payloads / other column counts: count, scan, scatter.  `values` must hold `values_capacity` elements
 * (numel is always enough; the one-pass paths never write beyond the capacity and still report the
 * needed size in total).  `workspace` = ct_bitmask_compress_workspace_bytes(rows, cols) bytes, 8-byte
 * aligned, need not be initialised.  (HIP graphs: every compute entry of this header can be captured; this one and its batch form bake a
 * per-launch generation tag into the launch, so a graph that contains them clears `workspace` inside the graph before the launch.) */
int64_t ct_bitmask_compress_workspace_bytes(int64_t rows, int64_t cols);
int ct_bitmask_compress(const void* x, int dt, int64_t rows, int64_t cols, void* values,
                        int64_t values_capacity, uint8_t* bitmask, int64_t* row_offsets, int64_t* total,
                        void* workspace, int64_t workspace_bytes, ct_stream_t stream);
/* The same compress for a TABLE of tensors in ONE launch (round 6): a checkpoint's sparse weights, e.g. what the restated
 * BitmaskCompressor.compress_state_dict loops over.  One ct_bitmask_compress launch of a checkpoint-sized tensor is a latency chain
 * (9 us for 1 MB, 10 us for 8 MB, 17 us for 23 MB of bf16: 2-27 % of the HBM rate); in one grid the tensors' chains run side by
 * side.  16- or 32-bit payloads (ONE element size per table), cols % 8 == 0, x and values 16-byte aligned, at most 1 GiB of payload
 * per tensor.  Protocol as the W4 batch: fill x / dt / rows / cols / values / values_capacity / bitmask / row_offsets / total of every
 * item, call ct_bitmask_batch_plan on the HOST copy (derived fields; returns the workgroup count or -1 and, in *workspace_bytes, the
 * size of the one workspace the launch needs: 8-byte aligned device memory, need not be initialised), copy the table to the device,
 * launch.  total[0] of every item receives its nnz (a word of a ct_mailbox_alloc block, for a host that waits; or device memory).
 * Outputs are bit-identical to ct_bitmask_compress item by item. */
typedef struct ct_bitmask_item {
    const void* x;
    void* values;
    uint8_t* bitmask;
    int64_t* row_offsets;
    int64_t* total;
    int64_t rows, cols, values_capacity; /* capacity in elements; numel is always enough */
    int32_t dt;                          /* element type code */
    int32_t is_float;                    /* derived */
    int64_t first_block, units, upr, slots_offset; /* derived */
    int32_t nwg, tpw, mask_dwords;                 /* derived */
    uint32_t gen;                                  /* derived */
} ct_bitmask_item;                       /* 15 64-bit words */
int64_t ct_bitmask_batch_plan(ct_bitmask_item* items_host, int n, int64_t* workspace_bytes);
int ct_bitmask_compress_batch(const ct_bitmask_item* items_dev, int n, int64_t total_blocks, int element_size, void* workspace,
                              int64_t workspace_bytes, ct_stream_t stream);

/* The decompress side of the same loop — loading a sparse checkpoint (the restated BitmaskCompressor.decompress_state_dict) — for a TABLE of
 * tensors in one launch: out = zeros; out[mask] = values per item, bit-identical to ct_bitmask_decompress item by item.  16- or 32-bit payloads
 * (ONE element size per table), cols x element size % 64 == 0, row_offsets given, values / out 16-byte and bitmask 4-byte aligned.
 * ct_bitmask_decompress_batch_plan fills the derived fields on the host copy and returns the workgroup count or -1. */
typedef struct ct_bitmask_ditem {
    const void* values;
    const uint8_t* bitmask;
    const int64_t* row_offsets;
    void* out;
    int64_t rows, cols, values_len; /* values_len in elements */
    int32_t dt;                     /* element type code */
    int32_t single;                 /* derived */
    int64_t first_block;            /* derived */
} ct_bitmask_ditem;                 /* 9 64-bit words */
int64_t ct_bitmask_decompress_batch_plan(ct_bitmask_ditem* items_host, int n);
int ct_bitmask_decompress_batch(const ct_bitmask_ditem* items_dev, int n, int64_t total_blocks, int element_size, ct_stream_t stream);

/* n byte ranges copied by one launch (the exact-size `values` of a batch leave the worst-case arena the kernels wrote into:
 * tensor[mask] owns nnz elements, restated S1 over utils/helpers.py:306-343).  ct_copy_batch_plan fills first_block on the host
 * copy and returns the workgroup count (16 KiB per workgroup) or -1. */
typedef struct ct_copy_item {
    const void* src;
    void* dst;
    int64_t bytes;
    int64_t first_block; /* derived */
} ct_copy_item;
int64_t ct_copy_batch_plan(ct_copy_item* items_host, int n);
int ct_copy_batch(const ct_copy_item* items_dev, int n, int64_t total_blocks, ct_stream_t stream);

/* sparse-bitmask decompress: out = zeros; out[mask] = values.  row_offsets may be NULL only if
 * fixed_row_nnz >= 0 (every row holds exactly that many values: the 2:4 codec).  16-bit payloads with
 * cols % 32 == 0 and 32-bit payloads with cols % 16 == 0 (as pairs of halves) take the LDS-window
 * kernel; everything else the general one. */
int ct_bitmask_decompress(const void* values, int64_t values_len, const uint8_t* bitmask,
                          const int64_t* row_offsets, int64_t fixed_row_nnz, int dt, int64_t rows,
                          int64_t cols, void* out, ct_stream_t stream);
/* popcount of each bitmask row -> row_counts (to rebuild row_offsets when a checkpoint lacks them) */
int ct_bitmask_row_popcount(const uint8_t* bitmask, int64_t rows, int64_t cols, int64_t* row_counts,
                            ct_stream_t stream);

/* sparse-24-bitmask compress: keep the 2 largest-|x| of every 4 consecutive elements (ties:
 * lower index first; NaN largest); values (rows, cols/2); bitmask (rows, ceil(cols/8)) */
int ct_sparse24_compress(const void* x, int dt, int64_t rows, int64_t cols, void* values,
                         uint8_t* bitmask, ct_stream_t stream);
/* the 2:4 magnitude mask alone (mask_creator, utils/semi_structured_conversions.py:301-330) */
int ct_sparse24_mask(const void* x, int dt, int64_t numel, uint8_t* mask, ct_stream_t stream);

/* sparse_semi_structured_from_dense_cutlass   utils/semi_structured_conversions.py:66-197
 * dense (m, k) of CT_F16/CT_BF16 (meta int16) or CT_I8 (meta int32); sparse (m, k/2);
 * meta (m, k/(4*Q)) reordered, Q = 4 (int16) or 8 (int32) */
int ct_cutlass24_from_dense(const void* dense, int dt, int64_t m, int64_t k, void* sparse, void* meta,
                            ct_stream_t stream);
/* sparse_semi_structured_to_dense_cutlass   :204-298.  sparse (m, k) -> dense (m, 2k) */
int ct_cutlass24_to_dense(const void* sparse, int dt, const void* meta, int meta_itemsize, int64_t m,
                          int64_t k, void* dense, ct_stream_t stream);

/* marlin-24 front end, fused (restated Marlin24Compressor.compress): fp16 quantize
 * of a 2:4-sparse 16-bit weight (every op rounded to fp16, like quantize() on weight.to(fp16)),
 * 2:4 compression of the codes (utils/semi_structured_conversions.py:66-197 on the 16-bit flavour:
 * int16 metadata, reordered) without any full-size intermediate.  comp: int8 (m, k/2) kept codes
 * (no offset); meta: int16 (m, k/16) reordered; bad[0] != 0 afterwards if some quad had more than
 * two non-zero codes (the weight is not 2:4).  scale: (m, k/cdiv), zp nullable. */
/* the same front end with the marlin-24 tile permutation + int4 packing fused in (rows % 64 == 0,
 * cols % 256 == 0): weight -> packed int32 (cols/32, rows*2) + reordered int16 metadata in ONE pass. */
int ct_marlin24_compress_w4(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt,
                            int64_t m, int64_t k, int64_t cdiv, int32_t* packed, int16_t* meta, int* bad,
                            ct_stream_t stream);
int ct_marlin24_quant_compress(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt,
                               int64_t m, int64_t k, int64_t cdiv, int bits, int8_t* comp, int16_t* meta,
                               int* bad, ct_stream_t stream);
/* the whole restated Marlin24Compressor.compress for int4 in one host call: ct_marlin24_compress_w4 followed by
 * ct_marlin24_pack_scales_f16 (scale (m, k/cdiv) bf16 / fp16 -> scale_packed fp16 (k/cdiv, m); group_perm != 0 selects
 * scale_perm, 0 scale_perm_single).  clear_bad == 0 leaves *bad alone (the violation is OR-ed in): the caller owns a
 * pre-zeroed flag, e.g. one slot of a ring that is read back once per batch instead of once per tensor. */
int ct_marlin24_compress_w4_full(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt,
                                 int64_t m, int64_t k, int64_t cdiv, int group_perm, int32_t* packed, int16_t* meta,
                                 void* scale_packed, int* bad, int clear_bad, ct_stream_t stream);

/* the same one-launch compress for a caller that must RAISE from the call when the weight is not 2:4 (upstream's
 * Marlin24Compressor.compress -> validate_sparsity_structure, historical sparse_quantized_compressors/marlin_24.py; the mask test is
 * utils/semi_structured_conversions.py / tensor_follows_mask_structure, utils/helpers.py:87-109): instead of a flag that is final only
 * when the stream has drained, the launch's last-reporting workgroup stores 1 (every quad of every row keeps at most two non-zero
 * codes) or 3 (violated) into *verdict_word at system scope as soon as every workgroup has evaluated its tiles — the caller zeroes
 * the word (pinned, device-mapped host memory: ct_mailbox_alloc) before the call and spins on it; outputs follow on `stream` as usual.
 *   workspace        CT_M24_VERDICT_WORKSPACE_BYTES of device memory on the stream's device, 128-byte aligned, owned by the CALLER
 *                    (as ct_bitmask_compress's): the workgroups count themselves in through it.  It must be all-zero when the launch
 *                    starts and is all-zero again once the verdict has been stored (the kernel orders its resets before that store),
 *                    so one zeroed allocation serves any number of consecutive calls.  One launch per workspace at a time: launches on
 *                    ONE stream may share a workspace without waiting (stream order separates them); launches that can overlap — other
 *                    streams, other host threads, other devices — need a workspace each, or must wait for the verdict in between.
 *                    The library keeps no device state: the entry is re-entrant like every other one (convert_checkpoint's worker
 *                    threads, entrypoints/convert/convert_checkpoint.py:129-132).
 *   clear_workspace  != 0: the call zeroes the workspace first (one hipMemsetAsync on `stream`) — the first use of an allocation
 *                    that is not known to be zero, or the use after a launch that did not finish.
 * CT_ERR_UNSUPPORTED for a layout the one-launch kernel does not take (use ct_marlin24_compress_w4_full + a stream wait). */
#define CT_M24_VERDICT_WORKSPACE_BYTES 8320
int ct_marlin24_compress_w4_verdict(const void* w, int wdt, const void* scale, int sdt, const void* zp, int zdt,
                                    int64_t m, int64_t k, int64_t cdiv, int group_perm, int32_t* packed, int16_t* meta,
                                    void* scale_packed, int64_t* verdict_word, void* workspace, int clear_workspace,
                                    ct_stream_t stream);

/* marlin-24 weight packing (historical Marlin24Compressor.pack_weight_24 with the table of
 * utils/permutations_24.py:20-45).  q: codes of dtype dt (CT_I32 / CT_I8 / float holding
 * integers), laid out (size_k, size_n) [transposed == 0] or as the un-transposed 2:4-compressed
 * matrix (size_n, size_k) [transposed == 1]; add_offset != 0 adds 2^(bits-1) to make the codes
 * unsigned.  packed: int32 (size_k/16, size_n*16*bits/32) */
int ct_marlin24_pack_weights(const void* q, int dt, int transposed, int add_offset, int64_t size_k,
                             int64_t size_n, int bits, int32_t* packed, ct_stream_t stream);
/* marlin-24 scale packing: scale (size_n, groups) 16-bit float -> transposed, permuted with
 * scale_perm (single == 0) or scale_perm_single (single != 0) of utils/permutations_24.py:46-53
 * -> (groups, size_n) */
int ct_marlin24_pack_scales(const void* scale, int dt, int64_t size_n, int64_t groups, int single,
                            void* out, ct_stream_t stream);
/* the same with the reference's `scale.to(torch.float16)` folded in: scale may be bfloat16, out is float16 */
int ct_marlin24_pack_scales_f16(const void* scale, int dt, int64_t size_n, int64_t groups, int single,
                                void* out, ct_stream_t stream);

/* ---------------------------------------------------------------------------- diagnostics
 * Exhaustive device-side check of the reciprocal fast path used by the bf16 fused kernels:
 * for every bf16 (x, s) pair in [s_lo_bits, s_hi_bits) x all 65536 x, compares
 * rnd_bf16(x * rcp(s)) with rnd_bf16(x / s).  mismatches[0] receives the count. */
/* the same for the fp16 quotient shortcut of ct_marlin24_quant_compress (reciprocal + one Newton step) */
int ct_selftest_f16_div(uint32_t s_lo_bits, uint32_t s_hi_bits, unsigned long long* mismatches,
                        ct_stream_t stream);
int ct_selftest_bf16_div(uint32_t s_lo_bits, uint32_t s_hi_bits, unsigned long long* mismatches,
                         ct_stream_t stream);
/* the quotients of the lean marlin-24 front end (ct_marlin24_compress_w4): mode 0 = fp16 x / fp16 scale by reciprocal +
 * Newton step, mode 1 = bf16 x / bf16 scale by ONE multiply with the reciprocal, mode 2 = the kernel's reciprocal (v_rcp_f32 + two
 * Newton steps) against the IEEE 1.0f / s, bit for bit; s bits in [s_lo_bits, s_hi_bits) */
int ct_selftest_m24_div(int mode, uint32_t s_lo_bits, uint32_t s_hi_bits, unsigned long long* mismatches,
                        ct_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CT_HIP_H */
