"""Tensor-level entry points of the hot path: thin wrappers that validate arguments, allocate
outputs with torch and launch the HIP kernels of libct_hip.so on the caller's current stream.

Every function mirrors a reference function (cited per function; paths relative to
/root/reference/src/compressed_tensors/).  There is no eager/CPU implementation here: CPU
tensors are staged through the GPU (H2D, kernel, D2H) so the arithmetic is always the HIP
path; without a GPU the calls raise.
"""
import math
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import DT, call, ptr, stream_of

__all__ = [
    "pack_to_int32",
    "unpack_from_int32",
    "quantize_tensor",
    "dequantize_tensor",
    "fake_quantize_tensor",
    "quantize_and_pack",
    "unpack_and_dequantize",
    "quantize_and_pack_many",
    "unpack_and_dequantize_many",
    "minmax_qparams",
    "minmax_qparams_float",
    "generate_gparam",
    "rtn_quantize_and_pack",
    "rtn_mxfp4_quantize_and_pack",
    "rtn_quantize_channel8",
    "rtn_nvfp4_quantize_and_pack",
    "pack_bitmasks",
    "unpack_bitmasks",
    "W4Batch",
    "launch_w4_words",
    "launch_zp4_words",
    "w4_batch_eligible",
    "q8_batch_group",
    "zp4_batch",
    "cast_to_fp4",
    "pack_fp4_to_uint8",
    "unpack_fp4_from_uint8",
    "compress_mx_scale",
    "decompress_mx_scale",
    "fp4_quantize_and_pack",
    "fp4_quantize_and_pack_stored",
    "fp4_unpack_and_dequantize",
    "marlin24_quant_compress",
    "marlin24_compress_w4",
    "marlin24_compress_w4_full",
    "selftest_m24_div",
    "bitmask_compress",
    "bitmask_decompress",
    "sparse24_mask",
    "sparse24_bitmask_compress",
    "sparse24_bitmask_decompress",
    "cutlass24_from_dense",
    "cutlass24_to_dense",
    "marlin24_pack_weights",
    "marlin24_pack_scales",
    "selftest_bf16_div",
    "selftest_f16_div",
    "selftest_fp4_div",
    "QuantLayout",
]

_FLOATS = (torch.float32, torch.float16, torch.bfloat16)


# --------------------------------------------------------------------------- staging helpers
def _dev(t: Optional[torch.Tensor], device: torch.device) -> Optional[torch.Tensor]:
    """contiguous, 16-byte aligned tensor on `device` (views into other storage are cloned)"""
    if t is None:
        return None
    if t.device != device:
        t = t.to(device)
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


def _compute_device(*tensors) -> torch.device:
    for t in tensors:
        if t is not None and t.is_cuda:
            return t.device
    return _lib.require_device()


def _home(out: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """results live on the device of the input they were derived from (reference behaviour)"""
    return out if out.device == like.device else out.to(like.device)


def _strategy_name(strategy) -> Optional[str]:
    if strategy is None:
        return None
    return str(getattr(strategy, "value", strategy)).lower()


# --------------------------------------------------------------------------- pack / unpack
def pack_to_int32(value: torch.Tensor, num_bits: int, packed_dim: int = 1) -> torch.Tensor:
    """compressors/pack_quantized/helpers.py:20-101 (same arguments, same errors)."""
    if value.dtype is not torch.int8:
        raise ValueError("Tensor must be quantized to torch.int8 before packing")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Packing is only supported for num_bits in [1, 8], got {num_bits}")
    if value.ndim > 2 and packed_dim == 0:
        return torch.stack([pack_to_int32(value[i], num_bits, packed_dim) for i in range(value.shape[0])])
    if value.ndim < 2:
        raise ValueError(f"pack_to_int32 expects a tensor with at least 2 dims, got {value.ndim}")
    dev = _compute_device(value)
    v = _dev(value, dev)
    if packed_dim == 0:
        rows, cols = v.shape
        out = torch.empty((math.ceil(rows * num_bits / 32), cols), dtype=torch.int32, device=dev)
        call("ct_pack_int32_dim0", ptr(v), rows, cols, num_bits, ptr(out), stream_of(v))
        return _home(out, value)
    # packing is row-wise, so N-D (MoE) tensors are a stack of rows (helpers.py:45-51)
    cols = v.shape[-1]
    rows = v.numel() // cols if cols else math.prod(v.shape[:-1])
    packed_cols = math.ceil(cols * num_bits / 32)
    out = torch.empty((*v.shape[:-1], packed_cols), dtype=torch.int32, device=dev)
    call("ct_pack_int32", ptr(v), rows, cols, num_bits, ptr(out), packed_cols, stream_of(v))
    return _home(out, value)


def unpack_from_int32(value: torch.Tensor, num_bits: int, shape: Sequence[int], packed_dim: int = 1) -> torch.Tensor:
    """compressors/pack_quantized/helpers.py:104-180 (same arguments, same errors)."""
    if value.dtype is not torch.int32:
        raise ValueError(f"Expected {torch.int32} but got {value.dtype}, Aborting unpack.")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Unpacking is only supported for num_bits in [1, 8], got {num_bits}")
    shape = tuple(int(s) for s in shape)
    if value.ndim > 2 and packed_dim == 0:
        return torch.stack(
            [unpack_from_int32(value[i], num_bits, shape[1:], packed_dim) for i in range(value.shape[0])]
        )
    dev = _compute_device(value)
    v = _dev(value, dev)
    if packed_dim == 0:
        words, cols = v.shape
        rows = shape[0]
        out = torch.empty((rows, cols), dtype=torch.int8, device=dev)
        call("ct_unpack_int32_dim0", ptr(v), words, cols, rows, num_bits, ptr(out), stream_of(v))
        return _home(out, value)
    words = v.shape[-1]
    rows = math.prod(v.shape[:-1])
    cols = shape[-1]
    out = torch.empty((*v.shape[:-1], cols), dtype=torch.int8, device=dev)
    call("ct_unpack_int32", ptr(v), rows, words, words, cols, num_bits, ptr(out), stream_of(v))
    return _home(out, value)


# --------------------------------------------------------------------------- quantization layout
def _col_group_of(g_idx: torch.Tensor, group_size: int) -> torch.Tensor:
    """activation ordering (forward_helpers.py:147-175): column c uses the group of its position in the g_idx-sorted order; a g_idx
    that still holds a -1 (not initialised) means plain column order.  The reference decides that with `-1 in g_idx`, a host read
    per call; here the choice is a device-side select — no synchronisation.  Nothing is cached: `param.data.copy_(...)` rewrites a
    g_idx in place without moving its pointer or bumping its version counter, so no key derived from the tensor is safe
    (ADVICE r03), and the two argsorts of a (cols,) vector are noise next to the weight pass."""
    flat = g_idx.detach().reshape(-1)
    if flat.is_cuda and flat.dtype is torch.int32 and flat.is_contiguous() and flat.numel() and int(group_size) > 0:
        # two small launches instead of two argsorts and five more tensor ops per call (ct_gidx_col_group: a balanced g_idx needs no sort at all) — a
        # module with activation ordering spent 86 us of host time per direction, most of it here
        out = torch.empty(flat.numel() + 1, dtype=torch.int32, device=flat.device)  # the last word: the kernel pair's mode scratch
        call("ct_gidx_col_group", ptr(flat), flat.numel(), int(group_size), ptr(out), out.data_ptr() + 4 * flat.numel(), stream_of(flat))
        return out[:-1]
    inv = torch.argsort(torch.argsort(flat))
    plain = torch.arange(flat.numel(), device=flat.device)
    return (torch.where((flat == -1).any(), plain, inv) // int(group_size)).to(torch.int32)


class QuantLayout:
    """How scale / zero-point entries map onto the elements of x:
    idx(r, c) = (r // rdiv) * scale_cols + (col_group[c] if col_group is not None else c // cdiv)
    (see include/ct_hip.h).  Built from the reference's strategy vocabulary."""

    __slots__ = ("rows", "cols", "rdiv", "cdiv", "scale_cols", "col_group", "is_group", "scale_zero_dim")

    def __init__(self, x_shape, scale: torch.Tensor, strategy, group_size=None, block_structure=None,
                 g_idx: Optional[torch.Tensor] = None):
        st = _strategy_name(strategy)
        cols = int(x_shape[-1]) if len(x_shape) else 1
        rows = math.prod(x_shape[:-1]) if len(x_shape) > 1 else 1
        self.rows, self.cols = rows, cols
        self.col_group = None
        self.is_group = st in ("group", "tensor_group")
        self.scale_zero_dim = scale.ndim == 0 and not self.is_group
        if self.is_group:
            if group_size is None:
                raise ValueError("group strategy requires group_size")
            group_size = int(group_size)
            # forward_helpers.py:141-145
            if cols >= group_size and cols % group_size != 0:
                raise ValueError(
                    "tensor column shape must be divisble "
                    f"by the given group_size {group_size} but got {cols}"
                )
            srows = math.prod(scale.shape[:-1]) if scale.ndim >= 2 else scale.numel()
            scols = scale.shape[-1] if scale.ndim >= 2 else 1
            if srows not in (1, rows):
                raise ValueError(f"scale of shape {tuple(scale.shape)} does not match {rows} rows")
            self.rdiv = 1 if srows == rows else max(rows, 1)
            self.cdiv = group_size
            self.scale_cols = scols
            if scols * group_size < cols and scols != math.ceil(cols / group_size):
                raise ValueError(f"scale of shape {tuple(scale.shape)} does not cover {cols} columns with groups of {group_size}")
            if g_idx is not None and g_idx.device.type != "meta":
                self.col_group = _col_group_of(g_idx, group_size)
        elif st == "block":
            if len(x_shape) != 2:
                raise NotImplementedError("block quantization expects a 2-D weight")
            bh, bw = (int(b) for b in block_structure)
            self.rdiv, self.cdiv, self.scale_cols = bh, bw, int(scale.shape[-1])
            if scale.shape[0] != math.ceil(rows / bh) or scale.shape[1] != math.ceil(cols / bw):
                raise ValueError(f"scale of shape {tuple(scale.shape)} does not match blocks {bh}x{bw} of {tuple(x_shape)}")
        elif st in ("channel", "token"):
            if scale.numel() != rows:
                if scale.numel() == 1:
                    self.rdiv, self.cdiv, self.scale_cols = max(rows, 1), max(cols, 1), 1
                else:
                    raise ValueError(f"scale of shape {tuple(scale.shape)} does not match {rows} channels")
            else:
                self.rdiv, self.cdiv, self.scale_cols = 1, max(cols, 1), 1
        elif st in ("tensor", None):
            if scale.numel() != 1:
                raise ValueError(f"per-tensor quantization expects a single scale, got shape {tuple(scale.shape)}")
            self.rdiv, self.cdiv, self.scale_cols = max(rows, 1), max(cols, 1), 1
        else:
            raise NotImplementedError(f"quantization strategy {st!r} is not supported by the MI355X path")

    def args(self, dev):
        cg = _dev(self.col_group, dev) if self.col_group is not None else None
        return (self.rows, self.cols, self.rdiv, self.cdiv, self.scale_cols, ptr(cg)), cg


def infer_dequant_layout(x_shape, scale: torch.Tensor):
    """strategy inferred from the scale shape when no args are given (lifecycle/forward.py:99-130)"""
    if scale.ndim in (0, 1):
        return "tensor", None, None
    if scale.ndim == 2:
        if scale.shape[1] == 1:
            return "channel", None, None
        if scale.shape[0] == 1 or scale.shape[0] == x_shape[0]:
            return "group", int(x_shape[1] / scale.shape[1]), None
        rows, cols = x_shape[-2], x_shape[-1]
        return "block", None, [rows // scale.shape[0], cols // scale.shape[1]]
    raise ValueError(
        f"Could not infer a quantization strategy from scale with {scale.ndim} "
        "dimmensions. Expected 0 or 2 dimmensions."
    )


def _result_dtype(x: torch.Tensor, scale: torch.Tensor, zero_dim: bool) -> torch.dtype:
    """torch result dtype of `x / scale` as the reference evaluates it (a 0-dim scale does not
    promote a dimensioned x; group strategies always unsqueeze the scale first)"""
    probe = scale if zero_dim else scale.reshape(-1)[:1]
    return torch.result_type(x, probe)


def _check_float(x, what):
    if x.dtype not in _FLOATS:
        raise NotImplementedError(f"{what} dtype {x.dtype} is not supported by the MI355X path")


def _zp_arg(zp, dev):
    if zp is None:
        return None, -1
    if zp.dtype not in DT or zp.dtype in (torch.bool,):
        zp = zp.to(torch.int32)
    z = _dev(zp, dev)
    return z, DT[z.dtype]


def _check_sz(layout: QuantLayout, scale, zp):
    need = (max(layout.rows, 1) + layout.rdiv - 1) // layout.rdiv * layout.scale_cols if layout.rows else 0
    if layout.rows and scale.numel() < need:
        raise ValueError(f"scale has {scale.numel()} entries, layout needs {need}")
    if zp is not None and zp.numel() != scale.numel():
        raise ValueError(f"zero_point shape {tuple(zp.shape)} does not match scale shape {tuple(scale.shape)}")


_F8 = torch.float8_e4m3fn


def _check_qtype(qtype, num_bits):
    qtype = getattr(qtype, "value", qtype)
    if qtype not in ("int", "float"):
        raise ValueError(f"Invalid quantization type {qtype}")
    if qtype == "float" and int(num_bits) not in (4, 8):
        raise NotImplementedError("Only num_bits in (4, 8) are supported")  # quant_args.py:479
    return qtype


def _gs_arg(global_scale, dev):
    if global_scale is None:
        return None
    return _dev(global_scale, dev).to(torch.float32).reshape(-1)[:1].contiguous()


def quantize_tensor(x, scale, zero_point, *, num_bits, strategy, group_size=None, block_structure=None,
                    dtype=None, g_idx=None, qtype="int", global_scale=None) -> torch.Tensor:
    """quantization/lifecycle/forward.py:36-73: returns `dtype` (int8/int32, float8_e4m3fn for FLOAT args, or a float
    type) or, when dtype is None, x.dtype for group strategies and the promoted type otherwise.  qtype "float": 8 bits
    clamp to +-448 and round to float8_e4m3fn, 4 bits clamp to +-6 and cast_to_fp4 (values stay in a float dtype), instead
    of rint.  global_scale (FLOAT 4-bit tensor_group): the effective scale is scale / global_scale in float32."""
    qtype = _check_qtype(qtype, num_bits)
    if global_scale is not None and not (qtype == "float" and int(num_bits) == 4):
        raise NotImplementedError("global_scale is implemented for FLOAT 4-bit quantization")
    _check_float(x, "x")
    _check_float(scale, "scale")
    layout = QuantLayout(x.shape, scale, strategy, group_size, block_structure, g_idx)
    _check_sz(layout, scale, zero_point)
    T = _result_dtype(x, scale, layout.scale_zero_dim) if global_scale is None else torch.float32
    out_dtype = dtype if dtype is not None else (x.dtype if layout.is_group else T)
    fp4 = qtype == "float" and int(num_bits) == 4
    allowed = _FLOATS if fp4 else (_F8, *_FLOATS) if qtype == "float" else (torch.int8, torch.int32, *_FLOATS)
    if out_dtype not in allowed:
        raise NotImplementedError(f"quantize ({qtype}, {num_bits} bits) to {out_dtype} is not supported by the MI355X path")
    dev = _compute_device(x, scale)
    xd, sd = _dev(x, dev), _dev(scale, dev)
    zd, zdt = _zp_arg(zero_point, dev)
    out = torch.empty(x.shape, dtype=out_dtype, device=dev)
    largs, _keep = layout.args(dev)
    if fp4:
        gs = _gs_arg(global_scale, dev)
        call("ct_quantize_fp4", ptr(xd), DT[xd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, ptr(gs), DT[T],
             ptr(out), DT[out_dtype], stream_of(xd))
    elif qtype == "float":
        call("ct_quantize_fp8", ptr(xd), DT[xd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, DT[T],
             ptr(out), DT[out_dtype], stream_of(xd))
    else:
        call("ct_quantize", ptr(xd), DT[xd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, int(num_bits), DT[T],
             ptr(out), DT[out_dtype], stream_of(xd))
    return _home(out, x)


def fake_quantize_tensor(x, scale, zero_point, *, num_bits, strategy, group_size=None, block_structure=None,
                         g_idx=None, qtype="int", global_scale=None) -> torch.Tensor:
    """quantization/lifecycle/forward.py:148-181 (forward_helpers.py:180-215); qtype / global_scale as in quantize_tensor."""
    qtype = _check_qtype(qtype, num_bits)
    if global_scale is not None and not (qtype == "float" and int(num_bits) == 4):
        raise NotImplementedError("global_scale is implemented for FLOAT 4-bit quantization")
    _check_float(x, "x")
    _check_float(scale, "scale")
    layout = QuantLayout(x.shape, scale, strategy, group_size, block_structure, g_idx)
    _check_sz(layout, scale, zero_point)
    T = _result_dtype(x, scale, layout.scale_zero_dim) if global_scale is None else torch.float32
    out_dtype = x.dtype if layout.is_group else (scale.dtype if global_scale is None else torch.float32)
    dev = _compute_device(x, scale)
    xd, sd = _dev(x, dev), _dev(scale, dev)
    zd, zdt = _zp_arg(zero_point, dev)
    out = torch.empty(x.shape, dtype=out_dtype, device=dev)
    largs, _keep = layout.args(dev)
    if qtype == "float" and int(num_bits) == 4:
        gs = _gs_arg(global_scale, dev)
        call("ct_fake_quantize_fp4", ptr(xd), DT[xd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, ptr(gs), DT[T],
             ptr(out), DT[out_dtype], stream_of(xd))
    elif qtype == "float":
        call("ct_fake_quantize_fp8", ptr(xd), DT[xd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, DT[T],
             ptr(out), DT[out_dtype], stream_of(xd))
    else:
        call("ct_fake_quantize", ptr(xd), DT[xd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, int(num_bits), DT[T],
             ptr(out), DT[out_dtype], stream_of(xd))
    return _home(out, x)


def dequantize_tensor(x_q, scale, zero_point=None, *, strategy=None, group_size=None, block_structure=None,
                      dtype=None, g_idx=None, global_scale=None) -> torch.Tensor:
    """quantization/lifecycle/forward.py:76-145 (forward_helpers.py:549-572)."""
    _check_float(scale, "scale")
    if x_q.dtype not in (torch.int8, torch.int32, _F8, *_FLOATS):
        raise NotImplementedError(f"dequantize from {x_q.dtype} is not supported by the MI355X path")
    if strategy is None:
        strategy, group_size, block_structure = infer_dequant_layout(x_q.shape, scale)
    layout = QuantLayout(x_q.shape, scale, strategy, group_size, block_structure, g_idx)
    _check_sz(layout, scale, zero_point)
    out_dtype = dtype if dtype is not None else scale.dtype
    if out_dtype not in _FLOATS:
        raise NotImplementedError(f"dequantize to {out_dtype} is not supported by the MI355X path")
    dev = _compute_device(x_q, scale)
    qd, sd = _dev(x_q, dev), _dev(scale, dev)
    zd, zdt = _zp_arg(zero_point, dev)
    out = torch.empty(x_q.shape, dtype=out_dtype, device=dev)
    largs, _keep = layout.args(dev)
    if global_scale is not None:
        gs = _gs_arg(global_scale, dev)
        call("ct_dequantize_gs", ptr(qd), DT[qd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, ptr(gs), ptr(out), DT[out_dtype],
             stream_of(qd))
    else:
        call("ct_dequantize", ptr(qd), DT[qd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, ptr(out), DT[out_dtype],
             stream_of(qd))
    return _home(out, x_q)


def quantize_and_pack(x, scale, zero_point, *, num_bits, strategy, group_size=None, block_structure=None,
                      g_idx=None) -> torch.Tensor:
    """Fused quantize(dtype=int8) + pack_to_int32 (compressors/pack_quantized/base.py:96-104):
    one pass over the weight, no int8 intermediate.  Returns int32 (*x.shape[:-1], ceil(cols*bits/32))."""
    _check_float(x, "weight")
    _check_float(scale, "scale")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Packing is only supported for num_bits in [1, 8], got {num_bits}")
    layout = QuantLayout(x.shape, scale, strategy, group_size, block_structure, g_idx)
    _check_sz(layout, scale, zero_point)
    T = _result_dtype(x, scale, layout.scale_zero_dim)
    dev = _compute_device(x, scale)
    xd, sd = _dev(x, dev), _dev(scale, dev)
    zd, zdt = _zp_arg(zero_point, dev)
    packed_cols = math.ceil(layout.cols * num_bits / 32)
    out = torch.empty((*x.shape[:-1], packed_cols), dtype=torch.int32, device=dev)
    largs, _keep = layout.args(dev)
    call("ct_quant_pack", ptr(xd), DT[xd.dtype], ptr(sd), DT[sd.dtype], ptr(zd), zdt, *largs, int(num_bits), DT[T],
         ptr(out), stream_of(xd))
    return _home(out, x)


def unpack_and_dequantize(packed, shape, scale, zero_point=None, *, num_bits, strategy=None, group_size=None,
                          block_structure=None, dtype=None, g_idx=None) -> torch.Tensor:
    """Fused unpack_from_int32 + dequantize (compressors/pack_quantized/base.py:155-161).
    `zero_point` is the UNPACKED int8 zero point (or None)."""
    if packed.dtype is not torch.int32:
        raise ValueError(f"Expected {torch.int32} but got {packed.dtype}, Aborting unpack.")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Unpacking is only supported for num_bits in [1, 8], got {num_bits}")
    _check_float(scale, "scale")
    shape = tuple(int(s) for s in shape)
    if strategy is None:
        strategy, group_size, block_structure = infer_dequant_layout(shape, scale)
    layout = QuantLayout(shape, scale, strategy, group_size, block_structure, g_idx)
    _check_sz(layout, scale, zero_point)
    out_dtype = dtype if dtype is not None else scale.dtype
    if out_dtype not in _FLOATS:
        raise NotImplementedError(f"dequantize to {out_dtype} is not supported by the MI355X path")
    dev = _compute_device(packed, scale)
    pd, sd = _dev(packed, dev), _dev(scale, dev)
    zd, zdt = _zp_arg(zero_point, dev)
    words = pd.shape[-1]
    if math.prod(pd.shape[:-1]) != layout.rows:
        raise ValueError(f"packed shape {tuple(pd.shape)} does not match weight shape {shape}")
    out = torch.empty(shape, dtype=out_dtype, device=dev)
    largs, _keep = layout.args(dev)
    rows, cols, rdiv, cdiv, scols, cg = largs
    call("ct_unpack_dequant", ptr(pd), rows, words, cols, int(num_bits), ptr(sd), DT[sd.dtype], ptr(zd), zdt, rdiv, cdiv,
         scols, cg, ptr(out), DT[out_dtype], stream_of(pd))
    return _home(out, packed)


def _w4_zp_fused_ok(x_or_packed, scale, zero_point, rows, cols, group) -> bool:
    """the layout ct_quant_pack_w4_zp / ct_unpack_dequant_w4_zp take: 2-D, 16-bit scale of shape (rows, cols / group), everything
    contiguous, 16-byte aligned and on one GPU"""
    if scale is None or scale.dtype not in (torch.bfloat16, torch.float16) or not scale.is_cuda or not x_or_packed.is_cuda:
        return False
    if cols % 32 or group % 32 or cols % group or rows <= 0:
        return False
    if tuple(scale.shape) != (rows, cols // group) or not scale.is_contiguous() or not x_or_packed.is_contiguous():
        return False
    if zero_point is None or zero_point.device != scale.device or not zero_point.is_contiguous() or x_or_packed.device != scale.device:
        return False
    return _aligned16(x_or_packed, scale, zero_point)


def quantize_and_pack_with_zp(x, scale, zero_point, *, num_bits, strategy, group_size=None):
    """PackedQuantizationCompressor.compress of an ASYMMETRIC int4 scheme in ONE launch (compressors/pack_quantized/base.py:96-110):
    returns (weight_packed int32 (R, C / 8), weight_zero_point int32 (ceil(R / 8), G) = pack_to_int32(zp, 4, packed_dim=0)), or None
    when the tensors are not the layout `ct_quant_pack_w4_zp` takes (the caller then composes quantize_and_pack + pack_to_int32)."""
    st = _strategy_name(strategy)
    if int(num_bits) != 4 or x.dim() != 2 or st not in ("group", "channel") or x.dtype not in (torch.bfloat16, torch.float16):
        return None
    rows, cols = int(x.shape[0]), int(x.shape[1])
    group = cols if st == "channel" else int(group_size or 0)
    if group <= 0 or zero_point is None or zero_point.dtype is not torch.int8 or scale is None or scale.dtype is not x.dtype:
        return None
    if not _w4_zp_fused_ok(x, scale, zero_point, rows, cols, group) or tuple(zero_point.shape) != tuple(scale.shape):
        return None
    packed = torch.empty((rows, cols // 8), dtype=torch.int32, device=x.device)
    zpp = torch.empty(((rows * 4 + 31) // 32, cols // group), dtype=torch.int32, device=x.device)
    call("ct_quant_pack_w4_zp", ptr(x), DT[x.dtype], ptr(scale), ptr(zero_point), rows, cols, group, ptr(packed), ptr(zpp), stream_of(x))
    return packed, zpp


def unpack_and_dequantize_with_zp(packed, shape, scale, zp_packed, *, num_bits, want_zero_point=True):
    """PackedQuantizationCompressor.decompress of an ASYMMETRIC int4 scheme in ONE launch (base.py:147-161): the zero points are read in
    their stored form.  Returns (weight of the scale's dtype, unpacked int8 zero point or None), or None when the tensors are not the
    layout `ct_unpack_dequant_w4_zp` takes (groups of 128, cols % 512 == 0, ...): the caller then unpacks the zero points first."""
    shape = tuple(int(v) for v in shape)
    if int(num_bits) != 4 or len(shape) != 2 or scale is None or scale.dim() != 2 or scale.shape[1] == 0 or packed.dtype is not torch.int32:
        return None
    rows, cols = shape
    if cols % scale.shape[1]:
        return None
    group = cols // scale.shape[1]
    if not w4_packed_zp_readable(cols, group) or rows * (cols // 8) >= 1 << 31 or zp_packed is None or zp_packed.dtype is not torch.int32:
        return None
    if tuple(zp_packed.shape) != ((rows * 4 + 31) // 32, scale.shape[1]) or tuple(packed.shape) != (rows, cols // 8):
        return None
    if not _w4_zp_fused_ok(packed, scale, zp_packed, rows, cols, group):
        return None
    out = torch.empty(shape, dtype=scale.dtype, device=packed.device)
    zp = torch.empty((rows, scale.shape[1]), dtype=torch.int8, device=packed.device) if want_zero_point else None
    call("ct_unpack_dequant_w4_zp", ptr(packed), ptr(scale), DT[scale.dtype], ptr(zp_packed), rows, cols, group, ptr(out), ptr(zp), stream_of(packed))
    return out, zp


def minmax_qparams(x, *, num_bits, group_size=None, symmetric=True):
    """Min-max observer + calculate_qparams (quantization/utils/helpers.py:50-137) over groups of
    `group_size` consecutive columns (None: the whole row).  Returns (scale x.dtype (R, G),
    zero_point int8 (R, G))."""
    _check_float(x, "weight")
    if x.ndim != 2:
        raise ValueError("minmax_qparams expects a 2-D weight")
    dev = _compute_device(x)
    xd = _dev(x, dev)
    rows, cols = xd.shape
    cdiv = int(group_size) if group_size else max(cols, 1)
    ng = math.ceil(cols / cdiv) if cols else 0
    scale = torch.empty((rows, ng), dtype=xd.dtype, device=dev)
    zp = torch.empty((rows, ng), dtype=torch.int8, device=dev)
    call("ct_minmax_qparams", ptr(xd), DT[xd.dtype], rows, cols, cdiv, int(num_bits), int(bool(symmetric)), ptr(scale),
         ptr(zp), stream_of(xd))
    return _home(scale, x), _home(zp, x)


# --------------------------------------------------------------------------- batched W4A16
_QP_KIND = {"fp8": 1, "nvfp4": 2, "mxfp4": 3, "mxfp8": 4}


def minmax_qparams_float(x, *, kind: str, group_size=None, global_scale=None) -> torch.Tensor:
    """Min-max observer + calculate_qparams for the symmetric FLOAT schemes (quantization/utils/helpers.py:50-137,
    mxfp_utils.py:37-143): kind "fp8" (scale in x.dtype), "nvfp4" (fp8-representable float32 scale under `global_scale`),
    "mxfp4" / "mxfp8" (power-of-two scale in x.dtype, group 32).  Returns the scale (R, G); the zero points of these schemes
    are zeros of the scheme's zp_dtype."""
    _check_float(x, "weight")
    if x.ndim != 2:
        raise ValueError("minmax_qparams_float expects a 2-D weight")
    if kind not in _QP_KIND:
        raise ValueError(f"unknown float qparams kind {kind!r}")
    if kind == "nvfp4" and global_scale is None:
        raise ValueError("nvfp4 scales need the global scale")
    dev = _compute_device(x)
    xd = _dev(x, dev)
    rows, cols = xd.shape
    cdiv = int(group_size) if group_size else max(cols, 1)
    ng = math.ceil(cols / cdiv) if cols else 0
    scale = torch.empty((rows, ng), dtype=torch.float32 if kind == "nvfp4" else xd.dtype, device=dev)
    gs = _gs_arg(global_scale, dev) if kind == "nvfp4" else None
    call("ct_minmax_qparams_float", ptr(xd), DT[xd.dtype], rows, cols, cdiv, _QP_KIND[kind], ptr(gs), ptr(scale), stream_of(xd))
    return _home(scale, x)


def generate_gparam(x: torch.Tensor) -> torch.Tensor:
    """generate_gparam (quantization/utils/helpers.py:308-337) of a whole weight: 448 * 6 / amax evaluated in x's dtype
    exactly as the eager expression does on the CPU (`float / tensor` = reciprocal, then product: two roundings), float32
    (1,), non-finite -> 1.  The tensor-wide amax is the row-wise min-max kernel followed by a reduction of one value per row."""
    _check_float(x, "weight")
    x2 = x.reshape(-1, x.shape[-1]) if x.ndim != 2 else x
    dev = _compute_device(x2)
    xd = _dev(x2, dev)
    rows, cols = xd.shape
    if rows == 0 or cols == 0:
        raise ValueError("generate_gparam of an empty tensor")
    row_amax = torch.empty((rows, 1), dtype=xd.dtype, device=dev)
    gs = torch.empty(1, dtype=torch.float32, device=dev)
    # the row maxima, then ONE workgroup: amax -> clamp(min=tiny) -> reciprocal -> x 2688 -> non-finite = 1 (seven tiny tensor ops cost 45-65 us of launches here)
    call("ct_generate_gparam", ptr(xd), DT[xd.dtype], rows, cols, ptr(row_amax), ptr(gs), stream_of(xd))
    return _home(gs, x)


def rtn_quantize_and_pack(x: torch.Tensor, *, group_size: Optional[int] = None, symmetric: bool = True):
    """Round-to-nearest int4 compress in ONE pass over the weight: min-max observer + calculate_qparams
    (quantization/utils/helpers.py:50-137) + quantize + pack_to_int32 (compressors/pack_quantized/base.py:96-104).
    Returns (packed int32 (R, C/8), scale (R, C/group) in x.dtype, zero_point int8 (R, C/group)) — bit-identical to
    `minmax_qparams` followed by `quantize_and_pack`.  group_size None = one group per row (channel)."""
    if x.dim() != 2:
        raise ValueError("rtn_quantize_and_pack expects a 2-D weight")
    if x.dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"the one-pass compress takes 16-bit float weights, got {x.dtype}")
    rows, cols = x.shape
    group = int(group_size) if group_size else cols
    if group <= 0 or cols % group != 0:
        raise ValueError(f"tensor column shape must be divisble by the given group_size {group} but got {cols}")
    if group % 32 != 0 or group > 2048 or (group // 32) & (group // 32 - 1):
        raise NotImplementedError(f"the one-pass compress needs group sizes 32 * 2^k <= 2048, got {group}; use minmax_qparams + quantize_and_pack")
    dev = _compute_device(x)
    xd = _dev(x, dev).contiguous()
    packed = torch.empty((rows, cols // 8), dtype=torch.int32, device=dev)
    scale = torch.empty((rows, cols // group), dtype=x.dtype, device=dev)
    zp = torch.empty((rows, cols // group), dtype=torch.int8, device=dev)
    call("ct_rtn_quant_pack_w4", ptr(xd), DT[xd.dtype], rows, cols, group, int(bool(symmetric)), ptr(packed), ptr(scale), ptr(zp), stream_of(xd))
    return _home(packed, x), _home(scale, x), _home(zp, x)


def rtn_mxfp4_quantize_and_pack(x: torch.Tensor, *, return_scale: bool = False):
    """Round-to-nearest MXFP4 compress in ONE pass over the weight (group 32): min-max observer + calculate_qparams' MX
    branch + quantize + cast_to_fp4 + pack + compress_mx_scale.  Returns (packed uint8 (R, C/2), scale uint8 (R, C/32))
    [, the float scale in x.dtype] — bit-identical to minmax_qparams_float("mxfp4") -> fp4_quantize_and_pack ->
    compress_mx_scale."""
    if x.dim() != 2:
        raise ValueError("rtn_mxfp4_quantize_and_pack expects a 2-D weight")
    if x.dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"the one-pass MXFP4 compress takes 16-bit float weights, got {x.dtype}")
    rows, cols = x.shape
    if cols % 32 != 0:
        raise ValueError(f"tensor column shape must be divisble by the given group_size 32 but got {cols}")
    dev = _compute_device(x)
    xd = _dev(x, dev).contiguous()
    packed = torch.empty((rows, cols // 2), dtype=torch.uint8, device=dev)
    code = torch.empty((rows, cols // 32), dtype=torch.uint8, device=dev)
    scale = torch.empty((rows, cols // 32), dtype=x.dtype, device=dev) if return_scale else None
    call("ct_rtn_mxfp4_quant_pack", ptr(xd), DT[xd.dtype], rows, cols, ptr(packed), ptr(code), ptr(scale), stream_of(xd))
    if return_scale:
        return _home(packed, x), _home(code, x), _home(scale, x)
    return _home(packed, x), _home(code, x)


def rtn_nvfp4_quantize_and_pack(x: torch.Tensor, global_scale: Optional[torch.Tensor] = None, *, return_scale: bool = False):
    """Round-to-nearest NVFP4 compress (groups of 16): generate_gparam of the weight unless `global_scale` is given, then ONE
    pass: min-max observer + calculate_qparams (float8 scales under the global scale) + quantize + cast_to_fp4 + pack.
    Returns (packed uint8 (R, C/2), scale float8_e4m3fn (R, C/16), global_scale float32 (1,)) [, the float32 scales]."""
    if x.dim() != 2:
        raise ValueError("rtn_nvfp4_quantize_and_pack expects a 2-D weight")
    if x.dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"the one-pass NVFP4 compress takes 16-bit float weights, got {x.dtype}")
    rows, cols = x.shape
    if cols % 32 != 0:
        raise NotImplementedError(f"the one-pass NVFP4 compress needs cols % 32 == 0, got {cols}; use minmax_qparams_float + fp4_quantize_and_pack")
    dev = _compute_device(x)
    xd = _dev(x, dev).contiguous()
    gs = generate_gparam(xd) if global_scale is None else _gs_arg(global_scale, dev)
    packed = torch.empty((rows, cols // 2), dtype=torch.uint8, device=dev)
    s8 = torch.empty((rows, cols // 16), dtype=torch.float8_e4m3fn, device=dev)
    scale = torch.empty((rows, cols // 16), dtype=torch.float32, device=dev) if return_scale else None
    call("ct_rtn_nvfp4_quant_pack", ptr(xd), DT[xd.dtype], rows, cols, ptr(gs), ptr(packed), ptr(s8), ptr(scale), stream_of(xd))
    out = (_home(packed, x), _home(s8, x), _home(gs, x))
    return out + (_home(scale, x),) if return_scale else out


def rtn_quantize_channel8(x: torch.Tensor, *, qtype: str = "int", symmetric: bool = True):
    """Channel-wise 8-bit round-to-nearest in ONE pass over the weight: per-row min-max observer + calculate_qparams +
    quantize to int8 (qtype "int") or float8_e4m3fn ("float").  Returns (codes (R, C), scale (R, 1) in x.dtype, zero_point):
    the zero point is int8 (R, 1) for INT, float8 zeros for FLOAT — bit-identical to the observer kernel followed by
    quantize_tensor."""
    if x.dim() != 2:
        raise ValueError("rtn_quantize_channel8 expects a 2-D weight")
    qtype = getattr(qtype, "value", qtype)
    if x.dtype not in (torch.bfloat16, torch.float16) or x.shape[1] % 8 != 0 or x.shape[1] > 16384 or (qtype == "float" and not symmetric):
        raise NotImplementedError("the one-pass channel-wise compress takes 16-bit weights with cols % 8 == 0 and cols <= 16384 (symmetric for FLOAT)")
    rows, cols = x.shape
    dev = _compute_device(x)
    xd = _dev(x, dev).contiguous()
    fp8 = qtype == "float"
    out = torch.empty((rows, cols), dtype=_F8 if fp8 else torch.int8, device=dev)
    scale = torch.empty((rows, 1), dtype=x.dtype, device=dev)
    zp = torch.zeros((rows, 1), dtype=_F8, device=dev) if fp8 else torch.empty((rows, 1), dtype=torch.int8, device=dev)
    call("ct_rtn_quant_channel8", ptr(xd), DT[xd.dtype], rows, cols, int(fp8), int(bool(symmetric)), ptr(out), ptr(scale), None if fp8 else ptr(zp), stream_of(xd))
    return _home(out, x), _home(scale, x), _home(zp, x)


def _aligned16(*tensors) -> bool:
    return all(t is None or t.data_ptr() % 16 == 0 for t in tensors)


def w4_batch_eligible(weight_shape, w_dtype, scale, zero_point, *, num_bits, strategy, group_size, g_idx=None, device=None) -> bool:
    """can this tensor join a one-launch W4 batch (`ct_quant_pack_batch` / `ct_unpack_dequant_batch`)?
    int4, 2-D 16-bit weights and scales of the same dtype on the same GPU (`device`: where the weight / packed words
    live), group or channel scales with cols % 32 == 0 and group % 32 == 0, int8 (or no) zero point, 16-byte aligned
    parameters, no activation ordering.  Anything else takes the per-module path."""
    if num_bits != 4 or g_idx is not None or len(weight_shape) != 2:
        return False
    if w_dtype not in (torch.bfloat16, torch.float16) or scale is None or scale.dtype != w_dtype or not scale.is_cuda:
        return False
    if device is not None and (scale.device != device or (zero_point is not None and zero_point.device != device)):
        return False
    if not _aligned16(scale, zero_point):
        return False
    rows, cols = int(weight_shape[0]), int(weight_shape[1])
    st = _strategy_name(strategy)
    if st == "channel":
        g = cols
    elif st == "group" and group_size:
        g = int(group_size)
    else:
        return False
    if rows <= 0 or cols % 32 or g % 32 or cols % g:
        return False
    if tuple(scale.shape) != (rows, cols // g) or not scale.is_contiguous():
        return False
    if zero_point is not None and (zero_point.dtype != torch.int8 or tuple(zero_point.shape) != tuple(scale.shape)
                                   or not zero_point.is_contiguous() or not zero_point.is_cuda):
        return False
    return True


def _upload_table(words, dev) -> torch.Tensor:
    """a host table of int64 words (an `array.array("q")` or a CPU int64 tensor) -> device bytes.  The staging buffer is pinned
    (PyTorch's caching host allocator hands the block out again only after the copy's stream event has completed), so the copy is
    asynchronous: no blocking pageable H2D (25-30 us per table before) on the path of a launch"""
    import ctypes

    if isinstance(words, torch.Tensor):
        src, nbytes = words.data_ptr(), 8 * words.numel()
    else:
        src, nbytes = words.buffer_info()[0], 8 * len(words)
    host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) if dev.type == "cuda" else torch.empty(nbytes, dtype=torch.uint8)
    ctypes.memmove(host.data_ptr(), src, nbytes)
    return host.to(dev, non_blocking=True)


def launch_w4_words(words: torch.Tensor, n: int, direction: str, dtype: torch.dtype, device: torch.device) -> None:
    """plan, upload and launch a W4 table that already exists as a flat CPU int64 tensor of `struct ct_w4_item` words (built by the
    C++ host loop, csrc/host/ct_hostpath.cpp) on `device`'s current stream.  The caller keeps the tensors the table points at alive."""
    if not n:
        return
    d = 0 if direction == "compress" else 1
    blocks = int(_lib.load().ct_w4_batch_plan(words.data_ptr(), n, d))
    if blocks < 0:
        raise ValueError(_lib.last_error())
    table = _upload_table(words, device)
    call("ct_quant_pack_batch" if d == 0 else "ct_unpack_dequant_batch", table.data_ptr(), n, blocks, DT[dtype], _lib.stream_on(device))


_Q8_KINDS = ("int8", "fp8", "fp8z")


def launch_q8_words(words: torch.Tensor, n: int, direction: str, dtype: torch.dtype, device: torch.device, kind: int, bits: int = 8) -> None:
    """`launch_w4_words` for a table of the 8-bit codecs (`ct_q8_quant_batch` / `ct_q8_dequant_batch`); kind 0 int8, 1 fp8, 2 fp8 with float8 zero points"""
    if not n:
        return
    d = 0 if direction == "compress" else 1
    blocks = int(_lib.load().ct_q8_batch_plan(words.data_ptr(), n, d))
    if blocks < 0:
        raise ValueError(_lib.last_error())
    table = _upload_table(words, device)
    if d == 0:
        call("ct_q8_quant_batch", table.data_ptr(), n, blocks, DT[dtype], kind, bits, _lib.stream_on(device))
    else:
        call("ct_q8_dequant_batch", table.data_ptr(), n, blocks, DT[dtype], kind, _lib.stream_on(device))


def launch_fp4_words(words: torch.Tensor, n: int, direction: str, device: torch.device, group: int, x_dtype=None, scale_dtype=None) -> None:
    """`launch_w4_words` for a table of FP4 tensors (`ct_fp4_quant_pack_batch` / `ct_fp4_unpack_dequant_batch`): group 16 = NVFP4 (every item carries its global
    scale), 32 = MXFP4; compress: the weights' and the float scales' dtype (one per table); decompress writes bfloat16"""
    if not n:
        return
    d = 0 if direction == "compress" else 1
    blocks = int(_lib.load().ct_fp4_batch_plan(words.data_ptr(), n, d))
    if blocks < 0:
        raise ValueError(_lib.last_error())
    table = _upload_table(words, device)
    if d == 0:
        lut = _mx_code_table(scale_dtype, device) if group == 32 else None
        call("ct_fp4_quant_pack_batch", table.data_ptr(), n, blocks, DT[x_dtype], DT[scale_dtype], int(group), ptr(lut), _lib.stream_on(device))
    else:
        call("ct_fp4_unpack_dequant_batch", table.data_ptr(), n, blocks, int(group), DT[torch.bfloat16], _lib.stream_on(device))


def launch_mx_scale_words(words: torch.Tensor, n: int, direction: str, device: torch.device, scale_dtype=None) -> None:
    """a table of MX scale tensors in ONE launch (`ct_mx_scale_batch`): "compress" 16-bit scales -> E8M0 codes, "decompress" codes -> bfloat16"""
    if not n:
        return
    blocks = int(_lib.load().ct_mx_scale_batch_plan(words.data_ptr(), n))
    if blocks < 0:
        raise ValueError(_lib.last_error())
    table = _upload_table(words, device)
    if direction == "compress":
        call("ct_mx_scale_batch", table.data_ptr(), n, blocks, 0, DT[scale_dtype], ptr(_mx_code_table(scale_dtype, device)), _lib.stream_on(device))
    else:
        call("ct_mx_scale_batch", table.data_ptr(), n, blocks, 1, -1, None, _lib.stream_on(device))


def launch_zp4_words(words: torch.Tensor, n: int, direction: str, device: torch.device) -> None:
    """`zp4_batch` for a table that already exists as a flat CPU int64 tensor (src, 0, 0, dst, unpacked rows, cols, 0 ... per item;
    built by the C++ host loop): plan, upload, ONE `ct_zp4_pack_dim0_batch` launch on `device`'s current stream"""
    if not n:
        return
    blocks = int(_lib.load().ct_zp4_batch_plan(words.data_ptr(), n))
    if blocks < 0:
        raise ValueError(_lib.last_error())
    table = _upload_table(words, device)
    call("ct_zp4_pack_dim0_batch", table.data_ptr(), n, blocks, 0 if direction == "pack" else 1, _lib.stream_on(device))


_ITEM_WORDS = _lib.ITEM_WORDS  # struct ct_w4_item of include/ct_hip.h in 64-bit words: 4 pointers, rows, cols, group, first_block, units,
#                                {upg_shift, upg}, zp_packed, main_blocks, {g_magic, g_shift}
_ITEM_TAIL = (0,) * (_ITEM_WORDS - 11)  # the derived words behind zp_packed


def w4_packed_zp_readable(cols: int, group: int) -> bool:
    """can the W4 decompress kernels take this tensor's zero points in their STORED (packed) form? (include/ct_hip.h, ct_w4_item.zp_packed:
    groups of 128, whole waves per row)"""
    return group == 128 and cols % 512 == 0


class W4Batch:
    """A table of tensors processed by ONE kernel launch per direction — the per-module loop of ModelCompressor without a
    launch (and a ~5 us host call) per module.

    entries: (src, scale, zero_point or None, dst, rows, cols, group[, zp_packed or None]) with src / dst in the order the direction needs
    ("compress": weight -> codes).  `zp_packed` ("w4" only; round 6): the int32 (ceil(rows / 8), cols / group) STORED form of an
    asymmetric scheme's zero points, written ("compress", beside the int8 `zero_point` that is quantized against) or read
    ("decompress": the kernel takes the zero points from it and fills `zero_point`, if given, with the unpacked int8) by the SAME launch.  kind "w4": W4A16 pack-quantized (`ct_quant_pack_batch` / `ct_unpack_dequant_batch`,
    dst / src = packed int32 words); kind "int8" / "fp8": the 8-bit codecs (`ct_q8_quant_batch` / `ct_q8_dequant_batch`, one
    byte per element, `bits` = the INT scheme's num_bits; group may be rows * cols for a per-tensor scale).

    The host table is one flat array of 64-bit words (13 per item, the layout of `struct ct_w4_item`) planned in place by the
    library and uploaded asynchronously from pinned memory."""

    def __init__(self, entries, direction: str, dtype: torch.dtype, kind: str = "w4", bits: int = 8):
        assert direction in ("compress", "decompress") and kind in ("w4", "int8", "fp8", "fp8z")  # fp8z: float8 codes with float8 zero points
        import array

        self.direction = 0 if direction == "compress" else 1
        self.kind, self.bits = kind, int(bits)
        self.dt = DT[dtype]
        self.keep = entries if isinstance(entries, list) else list(entries)  # the table holds raw pointers: keep the tensors alive
        n = self.n = len(self.keep)
        flat = []
        dev = None
        for e in self.keep:
            src, scale, zp, dst, rows, cols, group = e[:7]
            zpp = e[7] if len(e) > 7 else None
            flat += (src.data_ptr(), scale.data_ptr(), 0 if zp is None else zp.data_ptr(), dst.data_ptr(), rows, cols, group, 0, 0, 0,
                     0 if zpp is None else zpp.data_ptr(), *_ITEM_TAIL)
        self.blocks, self.table, self.device = 0, None, None
        if n:
            dev = self.keep[0][0].device
            words = array.array("q", flat)
            lib = _lib.load()
            plan = lib.ct_w4_batch_plan if kind == "w4" else lib.ct_q8_batch_plan
            self.blocks = int(plan(words.buffer_info()[0], n, self.direction))
            if self.blocks < 0:
                raise ValueError(_lib.last_error())
            self.device = dev
            self.table = _upload_table(words, dev)

    def launch(self, stream=None):
        """`stream`: a raw hipStream_t of self.device (default: the caller's current stream there)"""
        if not self.n:
            return
        s = _lib.stream_on(self.device, stream)
        if self.kind == "w4":
            call("ct_quant_pack_batch" if self.direction == 0 else "ct_unpack_dequant_batch", self.table.data_ptr(), self.n, self.blocks, self.dt, s)
        elif self.direction == 0:
            call("ct_q8_quant_batch", self.table.data_ptr(), self.n, self.blocks, self.dt, {"int8": 0, "fp8": 1, "fp8z": 2}[self.kind], self.bits, s)
        else:
            call("ct_q8_dequant_batch", self.table.data_ptr(), self.n, self.blocks, self.dt, {"int8": 0, "fp8": 1, "fp8z": 2}[self.kind], s)


def quantize_and_pack_many(items, *, num_bits, strategy, group_size=None):
    """`quantize_and_pack` for a LIST of (weight, scale, zero_point or None) on one GPU: the tensors a W4 batch takes (`w4_batch_eligible`:
    int4, 2-D 16-bit weights, group / channel scales) leave in ONE `ct_quant_pack_batch` launch — what a caller holding q / k / v or
    gate / up pairs should use instead of one call per tensor (a small tensor's single launch is bound by its ramp and its launch
    boundary: 4096 x 4096 alone reaches 58-61 % of the HBM peak, bench.py reports the pair next to it) — and the others one by one.
    Returns the packed int32 tensors in order.  Mirrors the loop body of compressors/pack_quantized/base.py:96-104 per item."""
    items = [(x, s, z) for x, s, z in items]
    out = [None] * len(items)
    entries, where = [], []
    for i, (x, s, z) in enumerate(items):
        if (x.is_cuda and x.is_contiguous() and _aligned16(x)
                and w4_batch_eligible(x.shape, x.dtype, s, z, num_bits=num_bits, strategy=strategy, group_size=group_size, device=x.device)
                and (not entries or (x.device == entries[0][0].device and x.dtype == entries[0][0].dtype))):
            rows, cols = int(x.shape[0]), int(x.shape[1])
            group = cols if _strategy_name(strategy) == "channel" else int(group_size)
            dst = torch.empty((rows, cols // 8), dtype=torch.int32, device=x.device)
            entries.append((x, s, z, dst, rows, cols, group))
            where.append(i)
        else:
            out[i] = quantize_and_pack(x, s, z, num_bits=num_bits, strategy=strategy, group_size=group_size)
    if entries:
        batch = W4Batch(entries, "compress", entries[0][0].dtype)
        batch.launch()
        batch.table.record_stream(torch.cuda.current_stream(batch.device))
        for i, e in zip(where, entries):
            out[i] = e[3]
    return out


def unpack_and_dequantize_many(items, *, num_bits, strategy, group_size=None):
    """the inverse of `quantize_and_pack_many` for a list of (packed, shape, scale, zero_point or None): ONE `ct_unpack_dequant_batch`
    launch for the eligible tensors (output dtype = the scale's), `unpack_and_dequantize` for the rest
    (compressors/pack_quantized/base.py:147-161 per item; `zero_point` is the UNPACKED int8 zero point)."""
    items = [(p, tuple(int(v) for v in shape), s, z) for p, shape, s, z in items]
    out = [None] * len(items)
    entries, where = [], []
    for i, (p, shape, s, z) in enumerate(items):
        if (p.is_cuda and p.dtype is torch.int32 and p.is_contiguous() and _aligned16(p) and len(shape) == 2
                and w4_batch_eligible(shape, s.dtype, s, z, num_bits=num_bits, strategy=strategy, group_size=group_size, device=p.device)
                and tuple(p.shape) == (shape[0], shape[1] // 8)
                and (not entries or (p.device == entries[0][0].device and s.dtype == entries[0][1].dtype))):
            group = shape[1] if _strategy_name(strategy) == "channel" else int(group_size)
            dst = torch.empty(shape, dtype=s.dtype, device=p.device)
            entries.append((p, s, z, dst, shape[0], shape[1], group))
            where.append(i)
        else:
            out[i] = unpack_and_dequantize(p, shape, s, z, num_bits=num_bits, strategy=strategy, group_size=group_size)
    if entries:
        batch = W4Batch(entries, "decompress", entries[0][1].dtype)
        batch.launch()
        batch.table.record_stream(torch.cuda.current_stream(batch.device))
        for i, e in zip(where, entries):
            out[i] = e[3]
    return out


def zp4_batch(pairs, direction: str) -> None:
    """`pack_to_int32(zp, 4, packed_dim=0)` ("pack": int8 (R, G) -> int32 (ceil(R / 8), G)) or its inverse ("unpack") for a list of
    (src, dst) tensor pairs on one GPU in ONE launch (`ct_zp4_pack_dim0_batch`); dst tensors are allocated by the caller."""
    assert direction in ("pack", "unpack")
    pairs = list(pairs)
    if not pairs:
        return
    import array

    flat = []
    for src, dst in pairs:
        unpacked = src if direction == "pack" else dst
        flat += (src.data_ptr(), 0, 0, dst.data_ptr(), int(unpacked.shape[0]), int(unpacked.shape[1]), 0, 0, 0, 0, 0, *_ITEM_TAIL)
    words = array.array("q", flat)
    blocks = int(_lib.load().ct_zp4_batch_plan(words.buffer_info()[0], len(pairs)))
    if blocks < 0:
        raise ValueError(_lib.last_error())
    dev = pairs[0][0].device
    table = _upload_table(words, dev)
    call("ct_zp4_pack_dim0_batch", table.data_ptr(), len(pairs), blocks, 0 if direction == "pack" else 1, _lib.stream_on(dev))
    table.record_stream(torch.cuda.current_stream(dev))


def q8_batch_group(shape, w_dtype, scale, zero_point, *, device, strategy=None, group_size=None, g_idx=None, f8_zero_point=False, block_structure=None):
    """Elements per scale (the table's `group`) if this tensor can join a one-launch 8-bit batch, else None.  2-D, 16-bit
    float dtype equal to the scale's, tensor / channel / group scales (strategy given, or inferred from the scale's shape like
    `dequantize` does, forward.py:99-130), cols % 16 == 0, group % 16 == 0, int8 or absent zero point of the scale's shape,
    everything contiguous, 16-byte aligned and on `device`."""
    if g_idx is not None or len(shape) != 2 or w_dtype not in (torch.bfloat16, torch.float16):
        return None
    if scale is None or scale.dtype != w_dtype or scale.device != device or not scale.is_contiguous() or not _aligned16(scale):
        return None
    rows, cols = int(shape[0]), int(shape[1])
    if rows <= 0 or cols % 16:
        return None
    st = _strategy_name(strategy)
    if scale.numel() == 1 and scale.dim() <= 1 and st in (None, "tensor"):
        group = rows * cols
    elif scale.dim() == 2 and scale.shape[0] == rows and scale.shape[1] == 1 and st in (None, "channel"):
        group = cols
    elif scale.dim() == 2 and scale.shape[0] == rows and scale.shape[1] > 1 and cols % scale.shape[1] == 0 and st in (None, "group"):
        group = cols // scale.shape[1]
        if st == "group" and group_size and int(group_size) != group:
            return None
    elif scale.dim() == 2 and st in (None, "block") and scale.shape[0] >= 1 and scale.shape[1] >= 1:
        # block strategy (forward.py:198-216; inferred like `dequantize` does, forward.py:118-130, when no strategy is given): the table's group is
        # -((rows per block << 24) | columns per block), include/ct_hip.h
        if st == "block":
            if block_structure is None or len(block_structure) != 2:
                return None
            bh, bw = int(block_structure[0]), int(block_structure[1])
        else:
            if rows % scale.shape[0] or cols % scale.shape[1]:
                return None
            bh, bw = rows // scale.shape[0], cols // scale.shape[1]
        if (bh < 1 or bw < 16 or bh & (bh - 1) or bw & (bw - 1) or bh >= 1 << 24 or bw >= 1 << 24 or cols % bw or rows * cols >= 1 << 34
                or tuple(scale.shape) != (-(-rows // bh), cols // bw)):
            return None
        group = -((bh << 24) | bw)
    else:
        return None
    if group > 0 and group % 16:
        return None
    # (f8_zero_point, round 6: the float8 zero points a calibrated FLOAT scheme carries — all zeros for the symmetric schemes upstream allows, but
    # read as the float8 values they are: kind "fp8z" of W4Batch)
    if zero_point is not None and (zero_point.dtype != (_F8 if f8_zero_point else torch.int8) or zero_point.shape != scale.shape or zero_point.device != device
                                   or not zero_point.is_contiguous()):
        return None
    return group


# --------------------------------------------------------------------------- bitmask codecs
def _bits_view(t: torch.Tensor) -> torch.Tensor:
    """fp8 payloads are moved as raw bytes"""
    if t.dtype.is_floating_point and t.element_size() == 1:
        return t.view(torch.int8)
    return t


def _elem_code(t: torch.Tensor) -> int:
    if t.dtype not in DT:
        raise NotImplementedError(f"element dtype {t.dtype} is not supported by the MI355X path")
    if t.element_size() not in (1, 2, 4):
        raise NotImplementedError(f"{t.element_size()}-byte elements are not supported by the sparse codecs")
    return DT[t.dtype]


def pack_bitmasks(bytemasks: torch.Tensor) -> torch.Tensor:
    """utils/helpers.py:306-318 (numpy.packbits(..., bitorder="little") on the GPU)."""
    dev = _compute_device(bytemasks)
    m = _dev(bytemasks if bytemasks.dtype in (torch.bool, torch.uint8) else bytemasks != 0, dev)
    cols = m.shape[-1] if m.ndim else 1
    rows = math.prod(m.shape[:-1]) if m.ndim > 1 else 1
    out = torch.empty((*m.shape[:-1], math.ceil(cols / 8)), dtype=torch.uint8, device=dev)
    call("ct_pack_bitmasks", ptr(m), rows, cols, ptr(out), stream_of(m))
    return _home(out, bytemasks)


def unpack_bitmasks(packed_bitmasks: torch.Tensor, original_shape) -> torch.Tensor:
    """utils/helpers.py:321-343; returns a bool tensor of `original_shape`."""
    shape = tuple(int(s) for s in original_shape)
    dev = _compute_device(packed_bitmasks)
    p = _dev(packed_bitmasks, dev)
    cols = shape[-1]
    rows = math.prod(shape[:-1])
    out = torch.empty(shape, dtype=torch.uint8, device=dev)
    call("ct_unpack_bitmasks", ptr(p), rows, cols, ptr(out), stream_of(p))
    return _home(out.view(torch.bool), packed_bitmasks)


_WS_BYTES = {}


def _bitmask_workspace_bytes(rows: int, cols: int) -> int:
    n = _WS_BYTES.get((rows, cols))
    if n is None:
        n = _WS_BYTES[(rows, cols)] = int(_lib.load().ct_bitmask_compress_workspace_bytes(rows, cols))
    return n


def bitmask_compress(tensor: torch.Tensor, two_pass: bool = False, exact: bool = True):
    """sparse-bitmask compression: returns (values, bitmask uint8 (R, ceil(C/8)), row_offsets int64 (R,)).

    `exact` (round 6, default True): `values` owns exactly nnz elements, as `tensor[mask]` does (restated S1 over utils/helpers.py:306-343) —
    the kernel writes into a worst-case sized buffer (nnz is not known before it has run) and the kept prefix is then copied out, one
    asynchronous device copy of 2 x nnz x itemsize bytes that the caller does not wait for; the worst-case buffer goes back to the allocator.
    `exact=False`: `values` is a VIEW of the worst-case buffer — no copy, but the result pins numel x itemsize bytes however sparse the
    tensor is (a "compressed" 50 %-sparse weight then holds as much memory as the dense one): for a caller that consumes the values at
    once (serialises them, copies them elsewhere) and drops them.

    Default: the fused form (`ct_bitmask_compress`; 16- and 32-bit elements: the register-resident kernel that reads the tensor once,
    otherwise count + scatter whose prefixes are sums of the counts; no scan kernel) into a worst-case sized value buffer, then
    one host wait for nnz — as unavoidable as the reference's `tensor[mask]` — to narrow it: the kernel stores nnz into a pinned
    host word (`_lib.Mailbox`) and the host spins on that word instead of copying it back.  `two_pass=True` keeps the
    count / scan / host read / scatter form that sizes `values` exactly before writing it."""
    if tensor.ndim < 1:
        raise ValueError("bitmask compression expects at least a 1-D tensor")
    if not two_pass and tensor.is_cuda and tensor.device.index == torch.cuda.current_device():
        # the same steps without the interpreter (csrc/host/ct_hostpath.cpp:bitmask_compress): this call cannot overlap its own
        # kernel — it returns nnz — so every microsecond of host work around the launch is a microsecond of the call
        hp = _lib.hostpath()
        if hp is not None:
            x = _bits_view(tensor)
            s = stream_of(x)
            mb = _lib.mailbox(s.device_index)
            r = hp.bitmask_compress(x, _elem_code(x), mb.host, mb.dev, s, bool(exact))
            if r is not None:
                _lib.check(r[0])
                return (r[1] if x is tensor else r[1].view(tensor.dtype)), r[2], r[3]
    dev = _compute_device(tensor)
    x = _dev(_bits_view(tensor), dev)
    dt = _elem_code(x)
    cols = x.shape[-1]
    rows = math.prod(x.shape[:-1]) if x.ndim > 1 else 1
    bitmask = torch.empty((rows, math.ceil(cols / 8)), dtype=torch.uint8, device=dev)
    row_offsets = torch.empty(rows, dtype=torch.int64, device=dev)
    s = stream_of(x)
    if two_pass:
        counts = torch.empty(rows + 1, dtype=torch.int64, device=dev)
        call("ct_bitmask_count", ptr(x), dt, rows, cols, ptr(bitmask), ptr(counts), s)
        total = counts[rows:]
        call("ct_exclusive_scan_i64", ptr(counts), rows, ptr(row_offsets), total.data_ptr(), s)
        nnz = int(total.item())
        values = torch.empty(nnz, dtype=x.dtype, device=dev)
        if nnz:
            call("ct_bitmask_scatter", ptr(x), dt, rows, cols, ptr(row_offsets), ptr(values), s)
    else:
        numel = rows * cols
        ws_bytes = _bitmask_workspace_bytes(rows, cols)
        ws = torch.empty(ws_bytes // 8, dtype=torch.int64, device=dev)
        buf = torch.empty(numel, dtype=x.dtype, device=dev)
        # nnz comes back through the thread's pinned mailbox word: the kernel's last wave stores it there (system scope) as soon
        # as the prefix is known — before the last values have left — and the host spins on the word; no D2H copy, no `.item()`
        mb = _lib.mailbox(s.device_index)
        mb.words[0] = -1
        call("ct_bitmask_compress", ptr(x), dt, rows, cols, ptr(buf), numel, ptr(bitmask), ptr(row_offsets), mb.dev, ptr(ws), ws_bytes, s)
        nnz = mb.wait_word(0, -1, s)
        if nnz < 0 or nnz > numel:  # the stream drained and the word still holds the pending mark (or junk): never slice with it
            raise RuntimeError(f"ct_bitmask_compress finished without reporting the number of non-zeros (mailbox word {nnz}, numel {numel})")
        # exact: the kept prefix leaves the worst-case buffer (an asynchronous copy; nothing waits for it) unless the buffer is full anyway
        values = buf[:nnz] if (not exact or nnz == numel) else buf[:nnz].clone()
    return _home(values.view(tensor.dtype), tensor), _home(bitmask, tensor), _home(row_offsets, tensor)


def bitmask_compress_many(tensors, exact: bool = True, arena_bytes: int = 1 << 30):
    """`bitmask_compress` for a LIST of tensors (a checkpoint's sparse weights): [(values, bitmask, row_offsets), ...] in order.  The tensors on
    the current GPU go through the C++ host loop in windows — their compress launches are queued back to back, each reporting nnz into its own
    mailbox word, and the host sizes the results afterwards: one wait per window instead of ~9 us of waiting per tensor (a single call cannot
    overlap its own kernel; a batch can).  `exact` as in `bitmask_compress`; in exact mode a window's kernels share ONE worst-case arena of at
    most `arena_bytes` (never less than one tensor) instead of a worst-case allocation per tensor, and the exact-size results are filled by
    one batched copy.  A window is ONE kernel launch (`ct_bitmask_compress_batch`): a single launch of a checkpoint-sized tensor is a latency
    chain at 2-27 % of the HBM rate, the tensors of a table run theirs side by side.  Everything the loop does not take (CPU tensors, other
    devices, views, 8-bit payloads, no host extension) is compressed one by one."""
    tensors = list(tensors)
    out = [None] * len(tensors)
    hp = _lib.hostpath()
    if hp is not None and torch.cuda.is_available():
        cur = torch.cuda.current_device()
        idx = [i for i, t in enumerate(tensors) if t.ndim >= 1 and t.is_cuda and t.device.index == cur and t.numel() > 0 and t.is_contiguous()]
        if idx:
            xs = [_bits_view(tensors[i]) for i in idx]
            dts = [DT[x.dtype] if (x.dtype in DT and x.element_size() in (1, 2, 4)) else -1 for x in xs]
            s = _lib.stream_on(xs[0].device)
            mb = _lib.mailbox(s.device_index)
            r = hp.bitmask_compress_many(xs, dts, mb.host, mb.dev, mb.BATCH_WORD0, mb.BATCH_WORDS, s, bool(exact), int(arena_bytes))
            _lib.check(r[0])
            for i, x, res in zip(idx, xs, r[1:]):
                if res is not None:
                    out[i] = (res[0] if x is tensors[i] else res[0].view(tensors[i].dtype), res[1], res[2])
    for i, t in enumerate(tensors):
        if out[i] is None:
            out[i] = bitmask_compress(t, exact=exact)
    return out


def bitmask_decompress(values: torch.Tensor, bitmask: torch.Tensor, shape, row_offsets: Optional[torch.Tensor] = None,
                       fixed_row_nnz: Optional[int] = None) -> torch.Tensor:
    """sparse-bitmask decompression: zeros(shape) with `values` written at the set bits."""
    shape = tuple(int(s) for s in shape)
    dev = _compute_device(values, bitmask)
    v = _dev(_bits_view(values).reshape(-1), dev)
    b = _dev(bitmask, dev)
    dt = _elem_code(v)
    cols = shape[-1]
    rows = math.prod(shape[:-1])
    s = stream_of(b)
    ro = None
    if fixed_row_nnz is None:
        if row_offsets is None:
            counts = torch.empty(rows, dtype=torch.int64, device=dev)
            ro = torch.empty(rows, dtype=torch.int64, device=dev)
            call("ct_bitmask_row_popcount", ptr(b), rows, cols, ptr(counts), s)
            call("ct_exclusive_scan_i64", ptr(counts), rows, ptr(ro), None, s)
        else:
            ro = _dev(row_offsets.to(torch.int64), dev)
    out = torch.empty(shape, dtype=v.dtype, device=dev)
    call("ct_bitmask_decompress", ptr(v), v.numel(), ptr(b), ptr(ro), -1 if fixed_row_nnz is None else int(fixed_row_nnz),
         dt, rows, cols, ptr(out), s)
    return _home(out.view(values.dtype), values)


def bitmask_decompress_many(items):
    """`bitmask_decompress` for a LIST of (values, bitmask, shape, row_offsets): [dense tensor, ...] in order — loading a sparse checkpoint.  The
    16- / 32-bit tensors on the current GPU that have row offsets leave in ONE launch per element size (`ct_bitmask_decompress_batch`, through the
    C++ host loop: allocation of the dense outputs, the table, the launch; nothing waits); everything else one by one."""
    items = [(v, b, tuple(int(s) for s in shape), ro) for v, b, shape, ro in items]
    out = [None] * len(items)
    hp = _lib.hostpath()
    if hp is not None and torch.cuda.is_available() and items:
        cur = torch.cuda.current_device()
        idx = [i for i, (v, b, shape, ro) in enumerate(items) if ro is not None and v.is_cuda and b.is_cuda and ro.is_cuda and v.device.index == cur and len(shape) >= 1
               and v.element_size() in (2, 4) and v.dtype in DT and ro.dtype is torch.int64]
        if idx:
            vs = [_bits_view(items[i][0]).reshape(-1) for i in idx]
            r = hp.bitmask_decompress_many(vs, [items[i][1] for i in idx], [items[i][3] for i in idx], [list(items[i][2]) for i in idx], [DT[v.dtype] for v in vs],
                                           _lib.stream_on(vs[0].device))
            _lib.check(r[0])
            for i, v, res in zip(idx, vs, r[1:]):
                if res is not None:
                    out[i] = res if v.dtype is items[i][0].dtype else res.view(items[i][0].dtype)
    for i, (v, b, shape, ro) in enumerate(items):
        if out[i] is None:
            out[i] = bitmask_decompress(v, b, shape, ro)
    return out


def sparse24_mask(tensor: torch.Tensor) -> torch.Tensor:
    """bool mask keeping the 2 largest-|x| of every 4 consecutive elements
    (mask_creator, utils/semi_structured_conversions.py:301-330; ties: lower index first)."""
    if tensor.numel() % 4 != 0:
        raise ValueError(f"Tensor of size {tensor.shape} can't be evenly divided into 4 groups")
    dev = _compute_device(tensor)
    x = _dev(_bits_view(tensor), dev)
    if x.numel() % 8 != 0:
        # the kernel works on units of 8 elements; mask_creator accepts any multiple of 4 (:301-330): one zero quad is appended
        # and its mask dropped again (a copy of the tensor — only for numel % 8 == 4, which no weight matrix of the path has)
        flat = torch.cat([x.reshape(-1), torch.zeros(4, dtype=x.dtype, device=dev)])
        mask = torch.empty(flat.shape, dtype=torch.uint8, device=dev)
        call("ct_sparse24_mask", ptr(flat), _elem_code(flat), flat.numel(), ptr(mask), stream_of(flat))
        return _home(mask[:-4].reshape(x.shape).view(torch.bool), tensor)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    call("ct_sparse24_mask", ptr(x), _elem_code(x), x.numel(), ptr(mask), stream_of(x))
    return _home(mask.view(torch.bool), tensor)


def sparse24_bitmask_compress(tensor: torch.Tensor):
    """sparse-24-bitmask compression: (values (R, C/2), bitmask uint8 (R, C/8))."""
    if tensor.ndim != 2:
        raise ValueError("2:4 bitmask compression expects a 2-D tensor")
    if tensor.numel() % 4 != 0:
        raise ValueError("Tensor size must be a multiple of 4 for TWO_FOUR sparsity")
    dev = _compute_device(tensor)
    x = _dev(_bits_view(tensor), dev)
    rows, cols = x.shape
    values = torch.empty((rows, cols // 2), dtype=x.dtype, device=dev)
    bitmask = torch.empty((rows, math.ceil(cols / 8)), dtype=torch.uint8, device=dev)
    call("ct_sparse24_compress", ptr(x), _elem_code(x), rows, cols, ptr(values), ptr(bitmask), stream_of(x))
    return _home(values.view(tensor.dtype), tensor), _home(bitmask, tensor)


def sparse24_bitmask_decompress(values: torch.Tensor, bitmask: torch.Tensor, shape) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    return bitmask_decompress(values, bitmask, shape, fixed_row_nnz=shape[-1] // 2)


# --------------------------------------------------------------------------- 2:4 cutlass + marlin-24
def cutlass24_from_dense(dense: torch.Tensor):
    """utils/semi_structured_conversions.py:66-197 -> (sparse (m, k/2), meta_reordered)."""
    if dense.dim() != 2:
        raise RuntimeError(f"Expected 2-dimensional dense tensor, got {dense.dim()}-dimensional tensor")
    if dense.dtype not in (torch.int8, torch.float16, torch.bfloat16):
        if dense.dtype in (torch.float32, torch.int32):
            raise NotImplementedError(f"{dense.dtype} 2:4 conversion is not supported by the MI355X path")
        raise RuntimeError(f"Invalid datatype {dense.dtype} of dense matrix")
    dev = _compute_device(dense)
    d = _dev(dense, dev)
    m, k = d.shape
    meta_dtype = torch.int32 if d.dtype == torch.int8 else torch.int16
    q = meta_dtype.itemsize * 2
    if m % 64 != 0:
        raise RuntimeError(f"Number of rows of dense matrix {m} must be divisible by 64")
    if k % (4 * q) != 0:
        raise RuntimeError(f"Number of columns of dense matrix {k} must be divisible by {4 * q}")
    sparse = torch.empty((m, k // 2), dtype=d.dtype, device=dev)
    meta = torch.empty((m, k // (4 * q)), dtype=meta_dtype, device=dev)
    call("ct_cutlass24_from_dense", ptr(d), DT[d.dtype], m, k, ptr(sparse), ptr(meta), stream_of(d))
    return _home(sparse, dense), _home(meta, dense)


def cutlass24_to_dense(sparse: torch.Tensor, meta_reordered: torch.Tensor) -> torch.Tensor:
    """utils/semi_structured_conversions.py:204-298."""
    if sparse.dim() != 2:
        raise RuntimeError(f"Expected 2-dimensional sparse tensor, got {sparse.dim()}-dimensional tensor")
    if meta_reordered.dim() != 2:
        raise RuntimeError(f"Expected 2-dimensional meta tensor, got {meta_reordered.dim()}-dimensional tensor")
    if meta_reordered.dtype not in (torch.int16, torch.int32):
        raise RuntimeError(f"Invalid datatype {meta_reordered.dtype} of meta matrix")
    dev = _compute_device(sparse)
    s, mt = _dev(sparse, dev), _dev(meta_reordered, dev)
    m, k = s.shape
    if mt.shape[0] != m:
        raise RuntimeError(
            f"Number of rows of meta matrix {mt.shape[0]} must be equal to number of columns of spase matrix {m}"
        )
    dense = torch.empty((m, 2 * k), dtype=s.dtype, device=dev)
    call("ct_cutlass24_to_dense", ptr(s), DT[s.dtype], ptr(mt), mt.dtype.itemsize, m, k, ptr(dense), stream_of(s))
    return _home(dense, sparse)


def marlin24_quant_compress(weight: torch.Tensor, scale: torch.Tensor, zero_point, *, num_bits: int, group_size: Optional[int]):
    """fused marlin-24 front end: fp16 quantize + 2:4 compress of a 16-bit 2:4-sparse weight.
    Returns (codes int8 (m, k/2), meta int16 (m, k/16) reordered, violated: bool) — `violated` costs one
    host read, as the reference pipeline's structure check did."""
    dev = _compute_device(weight)
    w, s = _dev(weight, dev).contiguous(), _dev(scale, dev).contiguous()
    zp = _dev(zero_point, dev).contiguous() if zero_point is not None else None
    m, k = w.shape
    g = k if not group_size or group_size > k else int(group_size)
    comp = torch.empty((m, k // 2), dtype=torch.int8, device=dev)
    meta = torch.empty((m, k // 16), dtype=torch.int16, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    call("ct_marlin24_quant_compress", ptr(w), DT[w.dtype], ptr(s), DT[s.dtype], ptr(zp), DT[zp.dtype] if zp is not None else -1,
         m, k, g, int(num_bits), ptr(comp), ptr(meta), ptr(bad), stream_of(w))
    return _home(comp, weight), _home(meta, weight), bad


def marlin24_compress_w4(weight: torch.Tensor, scale: torch.Tensor, zero_point, *, group_size: Optional[int]):
    """fully fused int4 marlin-24 weight path (rows % 64 == 0, cols % 256 == 0): fp16 quantize + 2:4 compress
    + tile-permuted nibble packing in one launch.  Returns (weight_packed int32 (k/32, m*2), meta int16
    (m, k/16) reordered, violated flag tensor)."""
    dev = _compute_device(weight)
    w, s = _dev(weight, dev).contiguous(), _dev(scale, dev).contiguous()
    zp = _dev(zero_point, dev).contiguous() if zero_point is not None else None
    m, k = w.shape
    g = k if not group_size or group_size > k else int(group_size)
    packed = torch.empty((k // 32, m * 2), dtype=torch.int32, device=dev)
    meta = torch.empty((m, k // 16), dtype=torch.int16, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    call("ct_marlin24_compress_w4", ptr(w), DT[w.dtype], ptr(s), DT[s.dtype], ptr(zp), DT[zp.dtype] if zp is not None else -1,
         m, k, g, ptr(packed), ptr(meta), ptr(bad), stream_of(w))
    return _home(packed, weight), _home(meta, weight), bad


def marlin24_compress_w4_full(weight: torch.Tensor, scale: torch.Tensor, zero_point, *, group_size: Optional[int], group_perm: bool,
                              flag_ptr: Optional[int] = None, stream=None):
    """the whole int4 marlin-24 compress in ONE host call and (16-bit scales) ONE launch (`ct_marlin24_compress_w4_full`): the
    weight path of `marlin24_compress_w4` plus the permuted fp16 scales.  `flag_ptr`: device address of a pre-zeroed int32 the
    violation is OR-ed into (a slot of the caller's ring; nothing is cleared or read here); None allocates and clears one.
    GPU tensors only — no staging, this is the fast path, kept to three allocations and one ctypes call.  Returns
    (weight_packed, meta already in the stored (k/32, 2m) shape, scale_packed fp16 (groups, m), flag tensor or None)."""
    m, k = weight.shape
    dev = weight.device
    g = k if not group_size or group_size > k else int(group_size)
    packed = torch.empty((k // 32, m * 2), dtype=torch.int32, device=dev)
    meta = torch.empty((k // 32, m * 2), dtype=torch.int16, device=dev)  # the (m, k/16) reordered matrix, viewed as upstream stores it
    scale_packed = torch.empty((k // g, m), dtype=torch.float16, device=dev)
    flag = None
    if flag_ptr is None:
        flag = torch.empty(1, dtype=torch.int32, device=dev)
        flag_ptr = flag.data_ptr()
    call("ct_marlin24_compress_w4_full", weight.data_ptr(), DT[weight.dtype], scale.data_ptr(), DT[scale.dtype],
         None if zero_point is None else zero_point.data_ptr(), -1 if zero_point is None else DT[zero_point.dtype], m, k, g, int(group_perm),
         packed.data_ptr(), meta.data_ptr(), scale_packed.data_ptr(), flag_ptr, int(flag is not None), stream if stream is not None else stream_of(weight))
    return packed, meta, scale_packed, flag


def selftest_m24_div(mode: int, s_lo_bits: int = 0, s_hi_bits: int = 65536) -> int:
    """mismatch count of the lean marlin-24 quotients against the IEEE divide (mode 0: fp16 / fp16 with the Newton step,
    mode 1: bf16 / bf16 by one multiply; mode 2: the kernel's reciprocal against `1.0f / s`, bit for bit)"""
    dev = _lib.require_device()
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    call("ct_selftest_m24_div", int(mode), s_lo_bits, s_hi_bits, ptr(out), stream_of(out))
    return int(out.item())


def marlin24_pack_weights(q: torch.Tensor, num_bits: int, *, transposed: bool = False, add_offset: bool = False):
    """marlin-24 weight packing.  q: (size_k, size_n) codes, or with transposed=True the
    un-transposed (size_n, size_k) 2:4-compressed matrix."""
    if num_bits not in (4, 8):
        raise ValueError("num_bits must be 4 or 8, got {}".format(num_bits))
    dev = _compute_device(q)
    qd = _dev(q, dev)
    size_k, size_n = (qd.shape[1], qd.shape[0]) if transposed else qd.shape
    out = torch.empty((size_k // 16, size_n * 16 * num_bits // 32), dtype=torch.int32, device=dev)
    call("ct_marlin24_pack_weights", ptr(qd), DT[qd.dtype], int(transposed), int(add_offset), size_k, size_n, num_bits,
         ptr(out), stream_of(qd))
    return _home(out, q)


def marlin24_pack_scales(scale: torch.Tensor, *, single: bool, to_float16: bool = False):
    """marlin-24 scale packing: (size_n, groups) -> permuted (groups, size_n); to_float16 folds the reference's
    `scale.to(torch.float16)` into the same kernel (a bfloat16 scale comes out as float16)."""
    dev = _compute_device(scale)
    s = _dev(scale, dev)
    size_n, groups = s.shape
    out = torch.empty((groups, size_n), dtype=torch.float16 if to_float16 else s.dtype, device=dev)
    call("ct_marlin24_pack_scales_f16" if to_float16 else "ct_marlin24_pack_scales", ptr(s), DT[s.dtype], size_n, groups, int(single), ptr(out), stream_of(s))
    return _home(out, scale)


# --------------------------------------------------------------------------- FP4 (E2M1) codecs
_FP4_SCALE_KIND = {"plain": 0, "f8e4m3": 1, "e8m0": 2}


def cast_to_fp4(x: torch.Tensor) -> torch.Tensor:
    """quantization/utils/fp4_utils.py:77-98: nearest E2M1 value (0, 0.5, 1, 1.5, 2, 3, 4, 6 and negatives), same dtype."""
    if x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise NotImplementedError(f"cast_to_fp4 takes float tensors, got {x.dtype}")
    dev = _compute_device(x)
    xin = _dev(x, dev).contiguous()
    out = torch.empty_like(xin)
    call("ct_fp4_cast", ptr(xin), DT[xin.dtype], ptr(out), xin.numel(), stream_of(xin))
    return _home(out, x)


def pack_fp4_to_uint8(x: torch.Tensor) -> torch.Tensor:
    """compressors/nvfp4/helpers.py:108-150 (same argument, same error): (m, n) E2M1-valued -> uint8 (m, n / 2)."""
    m, n = x.shape
    if n % 2 != 0:
        raise ValueError("tensor must have an even number of columns for nvfp4 compression")
    if x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        x = x.to(torch.bfloat16)
    dev = _compute_device(x)
    xin = _dev(x, dev).contiguous()
    out = torch.empty((m, n // 2), dtype=torch.uint8, device=dev)
    call("ct_fp4_pack", ptr(xin), DT[xin.dtype], ptr(out), xin.numel(), stream_of(xin))
    return _home(out, x)


def unpack_fp4_from_uint8(a: torch.Tensor, m: int, n: int, dtype: Optional[torch.dtype] = torch.bfloat16) -> torch.Tensor:
    """compressors/nvfp4/helpers.py:153-193 (same arguments): uint8 (m, n / 2) -> (m, n) E2M1 values of `dtype`."""
    assert a.dtype == torch.uint8
    dtype = dtype or torch.float32  # upstream's lookup table is float32
    if dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise NotImplementedError(f"unpack_fp4_from_uint8 produces float tensors, got {dtype}")
    if a.numel() * 2 != m * n:
        raise ValueError(f"{a.numel()} packed bytes do not hold a {m} x {n} tensor")
    dev = _compute_device(a)
    ain = _dev(a, dev).contiguous()
    out = torch.empty((m, n), dtype=dtype, device=dev)
    call("ct_fp4_unpack", ptr(ain), m * n, ptr(out), DT[dtype], stream_of(ain))
    return _home(out, a)


def compress_mx_scale(scale: torch.Tensor, scale_dtype: torch.dtype = torch.uint8) -> torch.Tensor:
    """compressors/mx_utils.py:18-31: E8M0 code 127 + floor(log2(scale)); log2 is evaluated in the scale's dtype, as
    upstream (a small tensor: one element per 32 weights).  16-bit scales on the GPU: ONE launch that reads the codes from the table of this very
    expression over every 16-bit pattern (`_mx_code_table`) instead of four tensor ops."""
    if scale.is_cuda and scale_dtype is torch.uint8 and scale.dtype in (torch.bfloat16, torch.float16) and scale.is_contiguous() and scale.numel():
        out = torch.empty(scale.shape, dtype=torch.uint8, device=scale.device)
        call("ct_mx_scale_compress", ptr(scale), DT[scale.dtype], scale.numel(), ptr(_mx_code_table(scale.dtype, scale.device)), ptr(out), stream_of(scale))
        return out
    return (127 + torch.floor(torch.log2(scale)).to(torch.int32)).to(scale_dtype)


def decompress_mx_scale(scale: torch.Tensor) -> torch.Tensor:
    """compressors/mx_utils.py:34-44: 2 ** (code - 127) as bfloat16 (uint8 codes on the GPU: one launch)"""
    if scale.is_cuda and scale.dtype is torch.uint8 and scale.is_contiguous() and scale.numel():
        out = torch.empty(scale.shape, dtype=torch.bfloat16, device=scale.device)
        call("ct_mx_scale_decompress", ptr(scale), scale.numel(), ptr(out), stream_of(scale))
        return out
    return 2.0 ** (scale.to(torch.int32) - 127).to(torch.bfloat16)


_MX_CODE_TABLES = {}


def _mx_code_table(dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """compress_mx_scale(scale, uint8) (mx_utils.py:18-31) of EVERY 16-bit pattern of `dtype`, evaluated once by that very expression on the host and
    kept on `device` (64 KB): the MXFP4 compress kernel reads its groups' stored codes from it, so they are upstream's for every input"""
    key = (dtype, device)
    t = _MX_CODE_TABLES.get(key)
    if t is None:
        every = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dtype)
        t = _MX_CODE_TABLES[key] = compress_mx_scale(every, torch.uint8).contiguous().to(device)
    return t


def _gs_fast(global_scale, dev):
    """the global scale as a contiguous float32 tensor of one element on `dev` (as _gs_arg; without tensor ops when it already is one)"""
    if global_scale is None:
        return None
    if global_scale.dtype is torch.float32 and global_scale.numel() == 1 and global_scale.device == dev and global_scale.data_ptr() % 4 == 0:
        return global_scale
    return _dev(global_scale, dev).to(torch.float32).reshape(-1)[:1].contiguous()


def fp4_quantize_and_pack_stored(weight: torch.Tensor, scale: torch.Tensor, global_scale: Optional[torch.Tensor], *, group_size: int, scale_dtype: torch.dtype):
    """`fp4_quantize_and_pack` plus the scale in its STORED form from the same launch — NVFP4 (group 16): `scale.to(float8_e4m3fn)`; MXFP4 (group 32):
    `compress_mx_scale(scale, uint8)` — or None when the layout is outside that kernel (the caller then converts the scale itself).  Returns
    (packed, stored_scale)."""
    want = torch.float8_e4m3fn if group_size == 16 else torch.uint8
    if (scale_dtype is not want or weight.dim() != 2 or weight.dtype not in (torch.bfloat16, torch.float16) or not weight.is_cuda or scale.device != weight.device
            or scale.dtype not in _FLOATS or (group_size == 32 and scale.dtype is torch.float32) or (global_scale is not None) != (group_size == 16)
            or not weight.is_contiguous() or not scale.is_contiguous() or weight.data_ptr() % 16 or scale.data_ptr() % 8):
        return None
    rows, cols = weight.shape
    if cols % group_size or (rows * cols) % 32 or tuple(scale.shape) != (rows, cols // group_size) or rows == 0:
        return None
    dev = weight.device
    gs = _gs_fast(global_scale, dev)
    out = torch.empty((rows, cols // 2), dtype=torch.uint8, device=dev)
    stored = torch.empty((rows, cols // group_size), dtype=want, device=dev)
    lut = _mx_code_table(scale.dtype, dev) if group_size == 32 else None
    call("ct_fp4_quant_pack_stored", ptr(weight), DT[weight.dtype], ptr(scale), DT[scale.dtype], ptr(gs), rows, cols, int(group_size), ptr(out), ptr(stored), ptr(lut),
         stream_of(weight))
    return out, stored


def fp4_quantize_and_pack(weight: torch.Tensor, scale: torch.Tensor, global_scale: Optional[torch.Tensor], *, group_size: int) -> torch.Tensor:
    """quantize(x, scale, global_scale, FP4 args) -> cast_to_fp4 -> pack_fp4_to_uint8 in one launch
    (compressors/nvfp4/base.py:88-95): uint8 (rows, cols / 2)."""
    if weight.dim() != 2:
        raise ValueError("FP4 compression expects a 2-D weight")
    if weight.shape[1] % 2 != 0:
        raise ValueError("tensor must have an even number of columns for nvfp4 compression")
    if weight.dtype not in _FLOATS:
        raise NotImplementedError(f"the MI355X FP4 path compresses float weights, got {weight.dtype}")
    dev = _compute_device(weight)
    w, s = _dev(weight, dev).contiguous(), _dev(scale, dev).contiguous()
    rows, cols = w.shape
    if tuple(s.shape) != (rows, cols // group_size):
        raise ValueError(f"scale shape {tuple(s.shape)} does not match ({rows}, {cols // group_size}) for group size {group_size}")
    gs = _gs_fast(global_scale, dev)
    out = torch.empty((rows, cols // 2), dtype=torch.uint8, device=dev)
    call("ct_fp4_quant_pack", ptr(w), DT[w.dtype], ptr(s), DT[s.dtype], ptr(gs), rows, cols, int(group_size), ptr(out), stream_of(w))
    return _home(out, weight)


def fp4_unpack_and_dequantize(packed: torch.Tensor, scale: torch.Tensor, global_scale: Optional[torch.Tensor], *, group_size: int,
                              scale_kind: str = "plain", dtype: torch.dtype = torch.bfloat16, return_scale: bool = False):
    """unpack_fp4_from_uint8 -> dequantize(x_q, scale, global_scale) in one launch (nvfp4/base.py:118-131).  `scale` is
    the STORED scale: float8-e4m3 (scale_kind "f8e4m3"), E8M0 uint8 ("e8m0") or a float tensor ("plain").
    `return_scale`: also the decompressed scale as bfloat16 — `scale.to(bfloat16)` resp. `decompress_mx_scale(scale)` — written by the same launch:
    returns (weight, scale_bf16)."""
    if packed.dtype != torch.uint8 or packed.dim() != 2:
        raise ValueError("packed FP4 weights are 2-D uint8 tensors")
    dev = _compute_device(packed)
    p, s = _dev(packed, dev).contiguous(), _dev(scale, dev).contiguous()
    rows, cols = p.shape[0], p.shape[1] * 2
    kind = _FP4_SCALE_KIND[scale_kind]
    sview = s.view(torch.uint8) if kind == 1 else s
    sdt = DT[sview.dtype] if kind == 0 else -1
    gs = _gs_fast(global_scale, dev)
    out = torch.empty((rows, cols), dtype=dtype, device=dev)
    if return_scale:
        sout = torch.empty((rows, cols // int(group_size)), dtype=torch.bfloat16, device=dev)
        if rows and cols:
            call("ct_fp4_unpack_dequant_scale", ptr(p), rows, cols, ptr(sview), kind, sdt, ptr(gs), int(group_size), ptr(out), DT[dtype], ptr(sout), stream_of(p))
        return _home(out, packed), _home(sout, packed)
    call("ct_fp4_unpack_dequant", ptr(p), rows, cols, ptr(sview), kind, sdt, ptr(gs), int(group_size), ptr(out), DT[dtype], stream_of(p))
    return _home(out, packed)


# --------------------------------------------------------------------------- diagnostics
def selftest_bf16_div(s_lo_bits: int = 0, s_hi_bits: int = 65536) -> int:
    """number of (x, s) bf16 pairs for which the reciprocal fast path disagrees with the IEEE
    divide (must be 0; see ct_quant.hip)."""
    dev = _lib.require_device()
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    call("ct_selftest_bf16_div", s_lo_bits, s_hi_bits, ptr(out), torch.cuda.current_stream(dev).cuda_stream)
    return int(out.item())


def selftest_f16_div(s_lo_bits: int = 0, s_hi_bits: int = 65536) -> int:
    """number of (x, s) fp16 pairs for which the reciprocal + Newton quotient of the marlin-24 front end
    disagrees with the IEEE divide after rounding to fp16 (must be 0; see ct_marlin24.hip)."""
    dev = _lib.require_device()
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    call("ct_selftest_f16_div", s_lo_bits, s_hi_bits, ptr(out), torch.cuda.current_stream(dev).cuda_stream)
    return int(out.item())


def selftest_fp4_div(x_dtype: torch.dtype = torch.bfloat16, m_lo: int = 0, m_hi: int = 1 << 23) -> int:
    """number of (x mantissa, s mantissa) pairs for which the shared-reciprocal quotient of the FP4 compress kernel
    differs from the IEEE fp32 divide (must be 0; see ct_fp4.hip)."""
    dev = _lib.require_device()
    out = torch.zeros(1, dtype=torch.int64, device=dev)
    call("ct_selftest_fp4_div", DT[x_dtype], m_lo, m_hi, ptr(out), torch.cuda.current_stream(dev).cuda_stream)
    return int(out.item())
