"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Loads oracle/libct_oracle.so (built from ct_oracle.c by oracle/Makefile) and exposes the
reference's hot-path functions on *CPU* torch tensors.  torch is used only to hold the
buffers (numpy has no bfloat16); all arithmetic happens in the C restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (compressed_tensors_amd) must never import it.

Reference functions restated (paths relative to /root/reference/src/compressed_tensors):
  pack_to_int32 / unpack_from_int32      compressors/pack_quantized/helpers.py:20-180
  quantize / dequantize / fake_quantize  quantization/lifecycle/forward.py:36-241,
                                         quantization/lifecycle/forward_helpers.py:118-215,523-572
  calculate_qparams                      quantization/utils/helpers.py:50-137
  pack_bitmasks / unpack_bitmasks        utils/helpers.py:306-343
  cutlass 2:4 from/to dense              utils/semi_structured_conversions.py:33-298
  marlin-24 permutations                 utils/permutations_24.py:20-53
  PackedQuantizationCompressor           compressors/pack_quantized/base.py:62-163
  NaiveQuantizationCompressor            compressors/naive_quantized/base.py:48-126
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libct_oracle.so")

F32, F16, BF16, I8, I32, U8, I16, I64, F8 = range(9)
_DT = {
    torch.float32: F32,
    torch.float16: F16,
    torch.bfloat16: BF16,
    torch.int8: I8,
    torch.int32: I32,
    torch.uint8: U8,
    torch.int16: I16,
    torch.int64: I64,
    torch.bool: U8,
    torch.float8_e4m3fn: F8,
}
_FLOAT_CODE_TO_TORCH = {F32: torch.float32, F16: torch.float16, BF16: torch.bfloat16}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ct_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libct_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.cto_bitmask_compress.restype = ctypes.c_int64
    return _lib


def num_threads() -> int:
    return int(lib().cto_num_threads())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _i64(v):
    return ctypes.c_int64(int(v))


def _cpu(t):
    assert t.device.type == "cpu", "the oracle works on CPU tensors only"
    return t.contiguous()


# --------------------------------------------------------------------------- pack/unpack
def pack_to_int32(value: torch.Tensor, num_bits: int, packed_dim: int = 1) -> torch.Tensor:
    if value.dtype is not torch.int8:
        raise ValueError("Tensor must be quantized to torch.int8 before packing")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Packing is only supported for num_bits in [1, 8], got {num_bits}")
    if value.ndim > 2:
        return torch.stack([pack_to_int32(v, num_bits, packed_dim) for v in value])
    v = value.t() if packed_dim == 0 else value
    v = _cpu(v)
    rows, cols = v.shape
    out = torch.empty((rows, math.ceil(cols * num_bits / 32)), dtype=torch.int32)
    rc = lib().cto_pack_int32(_p(v), _i64(rows), _i64(cols), num_bits, _p(out))
    assert rc == 0
    return out.t() if packed_dim == 0 else out


def unpack_from_int32(value: torch.Tensor, num_bits: int, shape, packed_dim: int = 1) -> torch.Tensor:
    if value.dtype is not torch.int32:
        raise ValueError(f"Expected {torch.int32} but got {value.dtype}, Aborting unpack.")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Unpacking is only supported for num_bits in [1, 8], got {num_bits}")
    shape = tuple(int(s) for s in shape)
    if value.ndim > 2:
        return torch.stack([unpack_from_int32(v, num_bits, shape[1:], packed_dim) for v in value])
    v = value.t() if packed_dim == 0 else value
    v = _cpu(v)
    rows, words = v.shape
    cols = shape[packed_dim]
    out = torch.empty((rows, cols), dtype=torch.int8)
    rc = lib().cto_unpack_int32(_p(v), _i64(rows), _i64(words), _i64(cols), num_bits, _p(out))
    assert rc == 0
    return out.t() if packed_dim == 0 else out


# --------------------------------------------------------------------------- quantization
def _resolve(x, scale, strategy, group_size, block_structure):
    """Returns (rows, cols, rdiv, cdiv, scale_cols, scale_is_0dim_after_reshape)."""
    cols = x.shape[-1]
    rows = x.numel() // cols if cols else 0
    strategy = str(getattr(strategy, "value", strategy))
    if strategy in ("group", "tensor_group"):
        if cols >= group_size and cols % group_size != 0:
            raise ValueError(
                "tensor column shape must be divisble "
                f"by the given group_size {group_size} but got {cols}"
            )
        s2 = scale
        while s2.ndim < 2:
            s2 = s2.unsqueeze(1)
        srows = 1
        for d in s2.shape[:-1]:  # an N-D weight (experts, rows, cols) carries an N-D scale (experts, rows, groups): rows are flattened
            srows *= int(d)
        rdiv = 1 if srows == rows else max(rows, 1)
        return rows, cols, rdiv, group_size, s2.shape[-1], False
    if strategy == "block":
        bh, bw = block_structure
        return rows, cols, bh, bw, scale.shape[-1], scale.ndim == 0
    if strategy in ("channel", "token") or (scale.ndim >= 1 and scale.numel() == rows and rows > 1):
        return rows, cols, 1, max(cols, 1), 1, scale.ndim == 0
    # tensor (or any single-value scale)
    if scale.numel() != 1:
        raise ValueError(f"cannot broadcast scale of shape {tuple(scale.shape)} for strategy {strategy}")
    return rows, cols, max(rows, 1), max(cols, 1), 1, scale.ndim == 0


def _col_group(g_idx, group_size):
    if g_idx is None or g_idx.device.type == "meta" or bool((g_idx == -1).any()):
        return None
    perm = torch.argsort(g_idx)
    inv = torch.argsort(perm)
    return (inv // group_size).to(torch.int32).contiguous()


def _result_code(x, scale, zero_dim):
    s = scale if zero_dim else scale.reshape(-1)[:1]
    return _DT[torch.result_type(x, s)]


def _quant_common(fn_name, x, scale, zero_point, num_bits, strategy, group_size, block_structure,
                  g_idx, out_dtype, with_bits=True, global_scale=None):
    """`out_dtype` may be a torch dtype or a callable mapping T (torch dtype) -> torch dtype."""
    x = _cpu(x)
    scale = _cpu(scale)
    zp = _cpu(zero_point) if zero_point is not None else None
    rows, cols, rdiv, cdiv, scols, zero_dim = _resolve(x, scale, strategy, group_size, block_structure)
    is_group = str(getattr(strategy, "value", strategy)) in ("group", "tensor_group")
    cg = _col_group(g_idx, group_size) if is_group else None
    tdt = _result_code(x, scale, zero_dim) if with_bits in (True, "fl", "tdt") else None
    gs = _cpu(global_scale).to(torch.float32).reshape(-1)[:1].contiguous() if global_scale is not None else None
    if gs is not None and with_bits:
        tdt = F32  # scale / global_scale is float32, and so is x / that
    if callable(out_dtype):
        out_dtype = out_dtype(_FLOAT_CODE_TO_TORCH[tdt])
    if out_dtype not in _DT:
        raise NotImplementedError(out_dtype)
    out = torch.empty(x.shape, dtype=out_dtype)
    args = [_p(x), _DT[x.dtype], _p(scale), _DT[scale.dtype], _p(zp), _DT[zp.dtype] if zp is not None else -1,
            _i64(rows), _i64(cols), _i64(rdiv), _i64(cdiv), _i64(scols), _p(cg)]
    if with_bits == "fl":  # (fkind, global scale, tdt)
        args += [num_bits, _p(gs), tdt]
    elif with_bits == "gs":
        args += [_p(gs)]
    elif with_bits == "tdt":
        args += [tdt]
    elif with_bits:
        args += [num_bits, tdt]
    args += [_p(out), _DT[out_dtype]]
    rc = getattr(lib(), fn_name)(*args)
    assert rc == 0
    return out


def quantize(x, scale, zero_point, *, num_bits, strategy, group_size=None, block_structure=None,
             dtype=None, g_idx=None, qtype="int", global_scale=None):
    """forward.py:36-73.  Output dtype: `dtype`; else x.dtype for group strategies
    (forward_helpers.py:134,171) and the promoted float type T of x / scale otherwise."""
    is_group = str(getattr(strategy, "value", strategy)) in ("group", "tensor_group")

    def out_dtype(T):
        if dtype is not None:
            return dtype
        return x.dtype if is_group else T

    if qtype == "float":  # FLOAT 8-bit: clamp to +-448, round to float8_e4m3fn; 4-bit: +-6, cast_to_fp4 (quant_args.py:463-486)
        assert num_bits in (8, 4)
        return _quant_common("cto_quantize_fl", x, scale, zero_point, 1 if num_bits == 8 else 2, strategy, group_size,
                             block_structure, g_idx, out_dtype, with_bits="fl", global_scale=global_scale)
    return _quant_common("cto_quantize", x, scale, zero_point, num_bits, strategy, group_size,
                         block_structure, g_idx, out_dtype)


def dequantize(x_q, scale, zero_point=None, *, strategy=None, group_size=None, block_structure=None,
               dtype=None, g_idx=None, global_scale=None):
    """forward.py:76-145 (strategy inferred from the scale shape when not given)."""
    if strategy is None:
        if scale.ndim in (0, 1):
            strategy = "tensor"
        elif scale.ndim == 2:
            if scale.shape[1] == 1:
                strategy = "channel"
            elif scale.shape[0] == 1 or scale.shape[0] == x_q.shape[0]:
                strategy, group_size = "group", int(x_q.shape[1] / scale.shape[1])
            else:
                strategy = "block"
                block_structure = [x_q.shape[-2] // scale.shape[0], x_q.shape[-1] // scale.shape[1]]
        else:
            raise ValueError(
                f"Could not infer a quantization strategy from scale with {scale.ndim} "
                "dimmensions. Expected 0 or 2 dimmensions."
            )
    out_dtype = dtype if dtype is not None else scale.dtype
    if global_scale is not None:
        return _quant_common("cto_dequantize_gs", x_q, scale, zero_point, 0, strategy, group_size,
                             block_structure, g_idx, out_dtype, with_bits="gs", global_scale=global_scale)
    return _quant_common("cto_dequantize", x_q, scale, zero_point, 0, strategy, group_size,
                         block_structure, g_idx, out_dtype, with_bits=False)


def fake_quantize(x, scale, zero_point, *, num_bits, strategy, group_size=None, block_structure=None,
                  g_idx=None, qtype="int", global_scale=None):
    """forward.py:148-181.  Group strategies cast back to x.dtype (forward_helpers.py:134,171);
    the others return scale.dtype (dequant * scale)."""
    st = str(getattr(strategy, "value", strategy))
    out_dtype = x.dtype if st in ("group", "tensor_group") else scale.dtype
    if qtype == "float":
        assert num_bits in (8, 4)
        return _quant_common("cto_fake_quantize_fl", x, scale, zero_point, 1 if num_bits == 8 else 2, strategy, group_size,
                             block_structure, g_idx, out_dtype, with_bits="fl", global_scale=global_scale)
    return _quant_common("cto_fake_quantize", x, scale, zero_point, num_bits, strategy, group_size,
                         block_structure, g_idx, out_dtype)


def calculate_qparams_minmax(x, *, num_bits, group_size=None, symmetric=True):
    """min/max over each group (or row when group_size is None) -> reference scale/zp
    (quantization/utils/helpers.py:50-137 applied to torch.aminmax of every group)."""
    x = _cpu(x)
    rows, cols = x.shape
    cdiv = group_size or cols
    ng = math.ceil(cols / cdiv)
    scale = torch.empty((rows, ng), dtype=x.dtype)
    zp = torch.empty((rows, ng), dtype=torch.int8)
    rc = lib().cto_calculate_qparams(_p(x), _DT[x.dtype], _i64(rows), _i64(cols), _i64(cdiv), num_bits,
                                     int(bool(symmetric)), _p(scale), _p(zp))
    assert rc == 0
    return scale, zp


_QP_KIND = {"fp8": 1, "nvfp4": 2, "mxfp4": 3, "mxfp8": 4}


def calculate_qparams_float(x, *, kind, group_size=None, global_scale=None):
    """calculate_qparams of the symmetric FLOAT schemes over each group's min / max (helpers.py:50-137,
    mxfp_utils.py:37-143): kind "fp8" (scale in x.dtype), "nvfp4" (fp8-representable float32 scale under the global
    scale), "mxfp4" / "mxfp8" (power-of-two scale in x.dtype from the E8M0 exponent)."""
    x = _cpu(x)
    rows, cols = x.shape
    cdiv = group_size or cols
    ng = math.ceil(cols / cdiv)
    out_dtype = torch.float32 if kind == "nvfp4" else x.dtype
    scale = torch.empty((rows, ng), dtype=out_dtype)
    gs = _cpu(global_scale).to(torch.float32).reshape(-1)[:1].contiguous() if global_scale is not None else None
    rc = lib().cto_calculate_qparams_float(_p(x), _DT[x.dtype], _i64(rows), _i64(cols), _i64(cdiv), _QP_KIND[kind], _p(gs), _p(scale), _DT[out_dtype])
    assert rc == 0
    return scale


def generate_gparam(x):
    """helpers.py:308-337 for a whole tensor: 448 * 6 / amax evaluated in x's dtype, returned as float32; non-finite -> 1.
    `python_float / tensor` is evaluated by torch as tensor.reciprocal() * float: TWO roundings to x's dtype."""
    x = _cpu(x)
    xf = x.float()
    mn, mx = min(float(xf.min()), 0.0), max(float(xf.max()), 0.0)
    amax = torch.tensor([max(abs(mn), abs(mx))], dtype=x.dtype).clamp(min=torch.finfo(x.dtype).tiny)
    recip = (torch.ones(1, dtype=torch.float32) / amax.float()).to(x.dtype)
    gs = (recip.float() * (448.0 * 6.0)).to(x.dtype).float()
    return torch.nan_to_num(gs, nan=1.0, posinf=1.0, neginf=1.0)


# --------------------------------------------------------------------------- bitmask
def pack_bitmasks(bytemasks: torch.Tensor) -> torch.Tensor:
    m = _cpu(bytemasks.to(torch.uint8))
    cols = m.shape[-1]
    rows = math.prod(m.shape[:-1])
    out = torch.empty((*m.shape[:-1], math.ceil(cols / 8)), dtype=torch.uint8)
    lib().cto_pack_bitmasks(_p(m), _i64(rows), _i64(cols), _p(out))
    return out


def unpack_bitmasks(packed: torch.Tensor, original_shape) -> torch.Tensor:
    p = _cpu(packed)
    shape = tuple(int(s) for s in original_shape)
    cols = shape[-1]
    rows = math.prod(shape[:-1])
    out = torch.empty(shape, dtype=torch.uint8)
    lib().cto_unpack_bitmasks(_p(p), _i64(rows), _i64(cols), _p(out))
    return out.bool()


def _bits_view(t):
    """fp8 tensors travel as raw int8 bits through the copy codecs."""
    if t.dtype in (getattr(torch, "float8_e4m3fn", None), getattr(torch, "float8_e5m2", None)):
        return t.view(torch.int8)
    return t


def bitmask_compress(tensor: torch.Tensor):
    """Returns (values, bitmask uint8 (R, ceil(C/8)), row_offsets int64 (R,))."""
    t = _cpu(_bits_view(tensor))
    cols = t.shape[-1]
    rows = math.prod(t.shape[:-1])
    values = torch.empty(t.numel(), dtype=t.dtype)
    bitmask = torch.empty((rows, math.ceil(cols / 8)), dtype=torch.uint8)
    row_offsets = torch.empty(rows, dtype=torch.int64)
    nnz = lib().cto_bitmask_compress(_p(t), _DT[t.dtype], _i64(rows), _i64(cols), _p(values), _p(bitmask),
                                     _p(row_offsets))
    return values[:nnz].clone().view(tensor.dtype), bitmask, row_offsets


def bitmask_decompress(values, bitmask, shape):
    v = _cpu(_bits_view(values))
    shape = tuple(int(s) for s in shape)
    cols = shape[-1]
    rows = math.prod(shape[:-1])
    out = torch.empty(shape, dtype=v.dtype)
    lib().cto_bitmask_decompress(_p(v), _p(_cpu(bitmask)), _DT[v.dtype], _i64(rows), _i64(cols), _p(out))
    return out.view(values.dtype)


def sparse24_mask(tensor):
    t = _cpu(_bits_view(tensor))
    if t.numel() % 4:
        raise ValueError("Tensor size must be a multiple of 4 for TWO_FOUR sparsity")
    mask = torch.empty(t.shape, dtype=torch.uint8)
    rc = lib().cto_sparse24_mask(_p(t), _DT[t.dtype], _i64(t.numel()), _p(mask))
    assert rc == 0
    return mask.bool()


def sparse24_bitmask_compress(tensor):
    t = _cpu(_bits_view(tensor))
    rows, cols = t.shape
    values = torch.empty((rows, cols // 2), dtype=t.dtype)
    bitmask = torch.empty((rows, math.ceil(cols / 8)), dtype=torch.uint8)
    rc = lib().cto_sparse24_compress(_p(t), _DT[t.dtype], _i64(rows), _i64(cols), _p(values), _p(bitmask))
    assert rc == 0
    return values.view(tensor.dtype), bitmask


def sparse24_bitmask_decompress(values, bitmask, shape):
    return bitmask_decompress(values.reshape(-1), bitmask, shape)


# --------------------------------------------------------------------------- 2:4 cutlass + marlin
def cutlass24_from_dense(dense):
    d = _cpu(dense)
    m, k = d.shape
    meta_dtype = torch.int32 if d.dtype == torch.int8 else torch.int16
    q = meta_dtype.itemsize * 2
    sparse = torch.empty((m, k // 2), dtype=d.dtype)
    meta = torch.empty((m, k // (4 * q)), dtype=meta_dtype)
    rc = lib().cto_cutlass24_from_dense(_p(d), _DT[d.dtype], _i64(m), _i64(k), _p(sparse), _p(meta))
    if rc != 0:
        raise RuntimeError(f"cutlass24_from_dense: unsupported shape/dtype (rc={rc})")
    return sparse, meta


def cutlass24_to_dense(sparse, meta):
    s = _cpu(sparse)
    m, k = s.shape
    dense = torch.empty((m, 2 * k), dtype=s.dtype)
    rc = lib().cto_cutlass24_to_dense(_p(s), _DT[s.dtype], _p(_cpu(meta)), meta.dtype.itemsize, _i64(m), _i64(k),
                                      _p(dense))
    assert rc == 0
    return dense


def marlin24_perm(num_bits):
    perm = torch.empty(1024, dtype=torch.int32)
    rc = lib().cto_marlin24_perm(num_bits, _p(perm))
    if rc != 0:
        raise ValueError("num_bits must be 4 or 8, got {}".format(num_bits))
    return perm


def marlin24_scale_perms():
    scale_perm = [i * 8 + j for i in range(8) for j in (0, 4, 1, 5, 2, 6, 3, 7)]
    scale_perm_single = [8 * i + j for i in range(8) for j in range(8)]
    return scale_perm, scale_perm_single


def marlin24_pack_weights(qw_int32, num_bits):
    q = _cpu(qw_int32.to(torch.int32))
    k, n = q.shape
    pf = 32 // num_bits
    out = torch.empty((k // 16, n * 16 // pf), dtype=torch.int32)
    rc = lib().cto_marlin24_pack_weights(_p(q), _i64(k), _i64(n), num_bits, _p(out))
    if rc != 0:
        raise ValueError(f"marlin24_pack_weights: bad shape (rc={rc})")
    return out


def marlin24_compress(weight, scale, zero_point, *, num_bits, strategy, group_size=None):
    """Restated historical Marlin24Compressor.compress_weight (absent from the reference
    snapshot).  Returns dict(weight_packed, scale_packed, meta)."""
    w16 = weight.to(torch.float16)
    s16 = scale.to(torch.float16)
    q = quantize(w16, s16, zero_point, num_bits=num_bits, strategy=strategy, group_size=group_size)
    comp, meta = cutlass24_from_dense(q)
    comp_t = comp.t().contiguous()
    s_t = s16.t().contiguous()
    size_k, size_n = comp_t.shape
    codes = comp_t.to(torch.int32) + (1 << num_bits) // 2
    packed = marlin24_pack_weights(codes, num_bits)
    sp, sps = marlin24_scale_perms()
    st = str(getattr(strategy, "value", strategy))
    if st == "group" and group_size is not None and group_size < size_k:  # size_k = in_features / 2 (w_shape of the compressed, transposed weight)
        s_p = s_t.reshape(-1, len(sp))[:, sp]
    else:
        s_p = s_t.reshape(-1, len(sps))[:, sps]
    s_p = s_p.reshape(-1, size_n).contiguous()
    meta2 = meta.reshape(-1).reshape(meta.shape[1] // 2, meta.shape[0] * 2)
    return {"weight_packed": packed, "scale_packed": s_p, "meta": meta2}


# --------------------------------------------------------------------------- state-dict codecs
def pack_quantized_compress(state_dict, *, num_bits, strategy, group_size=None, symmetric=True):
    """compressors/pack_quantized/base.py:62-114"""
    sd = dict(state_dict)
    weight = sd.pop("weight")
    scale = sd.get("weight_scale")
    zp = sd.get("weight_zero_point")
    g_idx = sd.get("weight_g_idx")
    q = quantize(weight, scale, zp, num_bits=num_bits, strategy=strategy, group_size=group_size,
                 dtype=torch.int8, g_idx=g_idx)
    sd["weight_packed"] = pack_to_int32(q, num_bits)
    sd["weight_shape"] = torch.tensor(weight.shape)
    st = str(getattr(strategy, "value", strategy))
    if not symmetric and st in ("group", "channel"):
        assert zp is not None, "Asymmetric quant requires zero-point values"
        sd["weight_zero_point"] = pack_to_int32(zp, num_bits, packed_dim=0).contiguous()
    if symmetric:
        sd.pop("weight_zero_point", None)
    return sd


def pack_quantized_decompress(state_dict, *, num_bits, strategy, symmetric=True):
    """compressors/pack_quantized/base.py:116-163"""
    sd = dict(state_dict)
    packed = sd.pop("weight_packed")
    scale = sd.get("weight_scale")
    zp = sd.get("weight_zero_point")
    g_idx = sd.get("weight_g_idx")
    shape = sd.get("weight_shape")
    st = str(getattr(strategy, "value", strategy))
    if not symmetric and st in ("group", "channel"):
        assert zp is not None, "Asymmetric quant requires zero-point values"
        zp_shape = (*[int(s) for s in shape[:-1]], scale.shape[-1])
        zp = unpack_from_int32(zp, num_bits, zp_shape, packed_dim=0)
        sd["weight_zero_point"] = zp
    unpacked = unpack_from_int32(packed, num_bits, shape)
    sd["weight"] = dequantize(unpacked, scale, zp, g_idx=g_idx)
    return sd


# --------------------------------------------------------------------------- FP4 (E2M1) codecs, SURVEY §8f N4
# Restated with index arithmetic instead of the reference's masked assignments; pinned against
# tests/golden/fp4.safetensors (generated by running the reference: oracle/gen_golden.py fp4).
_E2M1 = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)


def _fp4_index(a: torch.Tensor) -> torch.Tensor:
    """magnitude index of |value| after the clamp to 6 (quant_args.py:478): ties go to the even mantissa, exactly
    cast_to_fp4's thresholds (utils/fp4_utils.py:88-98): <= 0.25 -> 0, < 0.75 -> 0.5, <= 1.25 -> 1, < 1.75 -> 1.5,
    <= 2.5 -> 2, < 3.5 -> 3, <= 5 -> 4, else 6"""
    a = a.double().clamp(max=6.0)
    return ((a > 0.25).to(torch.uint8) + (a >= 0.75).to(torch.uint8) + (a > 1.25).to(torch.uint8) + (a >= 1.75).to(torch.uint8)
            + (a > 2.5).to(torch.uint8) + (a >= 3.5).to(torch.uint8) + (a > 5.0).to(torch.uint8))


def fp4_values(nib: torch.Tensor, dtype=torch.bfloat16) -> torch.Tensor:
    """nvfp4/helpers.py:157-193: magnitude table, negated (also the zero) when bit 3 is set"""
    mag = torch.tensor(_E2M1, dtype=torch.float32)[(nib & 7).long()]
    return torch.where((nib & 8) != 0, -mag, mag).to(dtype)


def cast_to_fp4(x: torch.Tensor) -> torch.Tensor:
    """quant_args.py:49-68 / fp4_utils.py:77-98: |x| rounded to the grid, times torch.sign(x) — so a negative input
    that rounds to zero gives -0.0, while -0.0 itself gives +0.0 (sign(-0.0) == 0)"""
    xf = x.float()
    mag = torch.tensor(_E2M1, dtype=torch.float32)[_fp4_index(xf.abs()).long()]
    return torch.where(xf < 0, -mag, mag).to(x.dtype)


def fp4_nibbles_of_values(q: torch.Tensor) -> torch.Tensor:
    """pack_fp4_to_uint8's index of an E2M1-valued tensor (nvfp4/helpers.py:139-145): sign from torch.signbit"""
    return _fp4_index(q.float().abs()) | (torch.signbit(q).to(torch.uint8) << 3)


def fp4_nibbles(t: torch.Tensor) -> torch.Tensor:
    """cast_to_fp4 followed by the pack index: the code of the scaled, float tensor `t`"""
    return fp4_nibbles_of_values(cast_to_fp4(t.float()))


def pack_fp4(nib: torch.Tensor) -> torch.Tensor:
    """helpers.py:147-150: element 2i in the low nibble of byte i"""
    n = nib.reshape(*nib.shape[:-1], nib.shape[-1] // 2, 2)
    return (n[..., 0] | (n[..., 1] << 4)).contiguous()


def unpack_fp4(packed: torch.Tensor) -> torch.Tensor:
    return torch.stack((packed & 0xF, packed >> 4), dim=-1).reshape(*packed.shape[:-1], packed.shape[-1] * 2)


def e8m0_encode(scale: torch.Tensor) -> torch.Tensor:
    """mx_utils.py:18-31: 127 + floor(log2(scale)) as uint8.  log2 is evaluated in the scale's own dtype (float32
    math rounded to that dtype), so a bfloat16 scale just below a power of two at a large exponent rounds UP to the
    next integer before the floor; MX scales are powers of two, where it is exact."""
    l = torch.from_numpy(np.log2(scale.float().numpy().astype(np.float64)).astype(np.float32)).to(scale.dtype)
    return (127 + torch.floor(l.float()).to(torch.int32)).to(torch.uint8)


def e8m0_decode(code: torch.Tensor) -> torch.Tensor:
    """mx_utils.py:34-44: 2^(code - 127) as bfloat16"""
    return torch.ldexp(torch.ones(code.shape, dtype=torch.float64), code.to(torch.int32) - 127).to(torch.bfloat16)


def _group_expand(s: torch.Tensor, cols: int) -> torch.Tensor:
    return s.repeat_interleave(cols // s.shape[-1], dim=-1)


def fp4_compress(weight, scale, global_scale=None, *, fmt):
    """NVFP4PackedCompressor.compress / MXFP4PackedCompressor.compress (nvfp4/base.py:68-104, mxfp4/base.py:47-51).
    forward_helpers.py:535-538: s_eff = scale / global_scale (float32 by promotion), t = x / s_eff in the promoted
    dtype T (float32 with a float32 scale; x.dtype when the scale has x.dtype, as for MXFP4)."""
    w, s = _cpu(weight), _cpu(scale)
    cols = w.shape[-1]
    if global_scale is not None:
        s_eff = (s.float() / _cpu(global_scale).float())  # float32 division, rounded to float32
        t = w.float() / _group_expand(s_eff, cols)
    else:
        T = torch.promote_types(w.dtype, s.dtype)
        t = (w.to(T) / _group_expand(s.to(T), cols)).float()  # rounded to T by torch, then widened exactly
    out = {"weight_packed": pack_fp4(fp4_nibbles(t))}
    out["weight_scale"] = s.to(torch.float8_e4m3fn) if fmt == "nvfp4-pack-quantized" else e8m0_encode(s)
    if global_scale is not None:
        out["weight_global_scale"] = _cpu(global_scale)
    return out


def fp4_decompress(state, *, fmt):
    """nvfp4/base.py:106-139 / mxfp4/base.py:53-55: the result is always bfloat16 (unpack's default dtype)"""
    nib = unpack_fp4(_cpu(state["weight_packed"]))
    v = fp4_values(nib, torch.bfloat16)
    cols = v.shape[-1]
    if fmt == "nvfp4-pack-quantized":
        s = _cpu(state["weight_scale"]).to(torch.bfloat16)
        s_eff = s.float() / _cpu(state["weight_global_scale"]).float()
        w = (v.float() * _group_expand(s_eff, cols)).to(torch.bfloat16)
    else:
        s = e8m0_decode(_cpu(state["weight_scale"]))
        w = (v.float() * _group_expand(s.float(), cols)).to(torch.bfloat16)
    out = {"weight": w, "weight_scale": s}
    if "weight_global_scale" in state:
        out["weight_global_scale"] = _cpu(state["weight_global_scale"])
    return out
