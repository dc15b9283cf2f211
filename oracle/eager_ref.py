"""Torch-eager CPU restatement of the reference's op sequence (TEST / BENCHMARK INFRASTRUCTURE ONLY).

The C oracle (ct_oracle.c) restates WHAT the reference computes; this module restates HOW it computes it —
the same eager tensor ops in the same order, so that its cost on the host cores is the reference's cost:
every op materialises a full-size temporary in the tensor dtype, the pack upcasts to int32, OR-accumulates
through `scatter_add_`, the unpack gathers a (rows x groups, 32) int32 matrix.  `bench.py`'s `cpu_baseline`
leg times it (`torch.set_num_threads(os.cpu_count())`) because /root/reference does not exist on the GPU
box; `tests/test_oracle_golden.py` pins it against the C oracle, the committed goldens and — where the
reference is importable — against the reference itself, outputs and timing shape.

Sequence followed (paths relative to /root/reference/src/compressed_tensors):
  quantize      quantization/lifecycle/forward.py:36-73 -> forward_helpers.py:118-177 (group reshape)
                -> forward_helpers.py:523-546 (`x / scale`, `+= zp.to(x.dtype)`, clamp, round, cast)
                with quant_args.py:460-496 (clamp with 0-dim fp32 bounds, torch.round, cast back)
  dequantize    forward.py:76-145 (strategy inferred from the scale shape) -> forward_helpers.py:549-572
  pack          compressors/pack_quantized/helpers.py:20-101
  unpack        compressors/pack_quantized/helpers.py:104-180
  compressors   compressors/pack_quantized/base.py:62-163, compressors/naive_quantized/base.py:48-126

The functions take tensors on any device: on the GPU box they also run the same op sequence through PyTorch-ROCm's own eager kernels —
what the reference itself would execute there (tests/test_gpu_parity.py::test_reference_op_sequence_on_the_gpu).

Never imported by the product package.
"""
import math

import torch

__all__ = ["quantize", "dequantize", "pack_to_int32", "unpack_from_int32", "pack_quantized_compress",
           "pack_quantized_decompress", "int_quantized_compress", "int_quantized_decompress"]


def _bounds(num_bits, device="cpu"):
    half = 2 ** num_bits // 2
    return torch.tensor(-half, dtype=torch.float32, device=device), torch.tensor(half - 1, dtype=torch.float32, device=device)


def _q(x, scale, zp, lo, hi, out_dtype):
    t = x / scale
    if zp is not None:
        t += zp.to(x.dtype)
    keep = t.dtype
    t = torch.round(torch.clamp(t, lo, hi)).to(keep)
    return t if out_dtype is None else t.to(out_dtype)


@torch.no_grad()
def quantize(x, scale, zero_point, *, num_bits, strategy, group_size=None, dtype=torch.int8):
    lo, hi = _bounds(num_bits, x.device)
    if strategy == "group":
        while scale.ndim < 2:
            scale = scale.unsqueeze(1)
            zero_point = None if zero_point is None else zero_point.unsqueeze(1)
        cols = x.shape[-1]
        if cols >= group_size and cols % group_size:
            raise ValueError(f"tensor column shape must be divisble by the given group_size {group_size} but got {cols}")
        xg = x.unflatten(-1, (math.ceil(cols / group_size), group_size))
        out = _q(xg, scale.unsqueeze(-1), None if zero_point is None else zero_point.unsqueeze(-1), lo, hi, dtype)
        return out.flatten(start_dim=-2).to(dtype if dtype is not None else x.dtype)
    return _q(x, scale, zero_point, lo, hi, dtype)  # tensor / channel: plain broadcasting


@torch.no_grad()
def dequantize(x_q, scale, zero_point=None, dtype=None):
    """strategy from the scale's shape, output in scale.dtype unless told otherwise"""
    dtype = scale.dtype if dtype is None else dtype
    grouped = scale.ndim == 2 and scale.shape[1] not in (1,) and scale.shape[0] in (1, x_q.shape[0]) and x_q.ndim == 2
    if grouped:
        gs = x_q.shape[-1] // scale.shape[1]
        v = x_q.unflatten(-1, (scale.shape[1], gs))
        s, z = scale.unsqueeze(-1), None if zero_point is None else zero_point.unsqueeze(-1)
    else:
        v, s, z = x_q, scale, zero_point
    d = v.to(s.dtype)
    if z is not None:
        d = d - z.to(s.dtype)
    d = d * s
    if grouped:
        d = d.flatten(start_dim=-2)
    return d.to(dtype)


def _lanes(num_bits, device="cpu"):
    first_bit = torch.arange(32, dtype=torch.int32, device=device) * num_bits
    return (first_bit // 32).long(), first_bit % 32


@torch.no_grad()
def pack_to_int32(value, num_bits, packed_dim=1):
    if value.dtype is not torch.int8:
        raise ValueError("Tensor must be quantized to torch.int8 before packing")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Packing is only supported for num_bits in [1, 8], got {num_bits}")
    if value.ndim > 2:
        return torch.stack([pack_to_int32(v, num_bits, packed_dim) for v in value])
    u = value.to(torch.int32) + (1 << (num_bits - 1))
    if packed_dim == 0:
        u = u.transpose(0, 1)
    rows, cols = u.shape
    words = math.ceil(cols * num_bits / 32)
    full = math.ceil(cols / 32) * 32
    if full > cols:
        u = torch.nn.functional.pad(u, (0, full - cols))
    groups = full // 32
    ug = u.reshape(rows * groups, 32)
    acc = torch.zeros(rows * groups, num_bits, dtype=torch.int32, device=u.device)
    word, off = _lanes(num_bits, u.device)
    acc.scatter_add_(1, word.unsqueeze(0).expand(rows * groups, -1), ug << off.unsqueeze(0))
    spill = off + num_bits - 32
    straddles = spill > 0
    if straddles.any():
        hi = ug[:, straddles] >> (num_bits - spill[straddles]).unsqueeze(0)
        acc.scatter_add_(1, (word[straddles] + 1).unsqueeze(0).expand(rows * groups, -1), hi)
    out = acc.view(rows, groups * num_bits)[:, :words]
    return out.transpose(0, 1) if packed_dim == 0 else out


@torch.no_grad()
def unpack_from_int32(value, num_bits, shape, packed_dim=1):
    if value.dtype is not torch.int32:
        raise ValueError(f"Expected {torch.int32} but got {value.dtype}, Aborting unpack.")
    if not 1 <= num_bits <= 8:
        raise ValueError(f"Unpacking is only supported for num_bits in [1, 8], got {num_bits}")
    if value.ndim > 2:
        return torch.stack([unpack_from_int32(v, num_bits, shape[1:], packed_dim) for v in value])
    if packed_dim == 0:
        value = value.transpose(0, 1)
    rows, words = value.shape
    cols = int(shape[packed_dim])
    if words % num_bits:
        extra = num_bits - words % num_bits
        value = torch.nn.functional.pad(value, (0, extra))
        words += extra
    groups = words // num_bits
    vg = value.reshape(rows * groups, num_bits)
    word, off = _lanes(num_bits, value.device)
    low = torch.clamp(32 - off, max=num_bits)
    got = (vg[:, word] >> off.unsqueeze(0)) & ((1 << low) - 1).unsqueeze(0)
    straddles = low < num_bits
    rest = num_bits - low[straddles]
    got[:, straddles] |= (vg[:, word[straddles] + 1] & ((1 << rest) - 1).unsqueeze(0)) << low[straddles].unsqueeze(0)
    out = got.view(rows, groups * 32)[:, :cols]
    if packed_dim == 0:
        out = out.transpose(0, 1)
    return (out - (1 << (num_bits - 1))).to(torch.int8)


@torch.no_grad()
def pack_quantized_compress(state_dict, *, num_bits, strategy, group_size=None, symmetric=True):
    sd = dict(state_dict)
    w = sd.pop("weight")
    zp = sd.get("weight_zero_point")
    q = quantize(w, sd["weight_scale"], zp, num_bits=num_bits, strategy=strategy, group_size=group_size, dtype=torch.int8)
    sd["weight_packed"] = pack_to_int32(q, num_bits)
    sd["weight_shape"] = torch.tensor(w.shape)
    if not symmetric and strategy in ("group", "channel"):
        sd["weight_zero_point"] = pack_to_int32(zp.to(torch.int8), num_bits, packed_dim=0).contiguous()
    if symmetric:
        sd.pop("weight_zero_point", None)
    return sd


@torch.no_grad()
def pack_quantized_decompress(state_dict, *, num_bits, strategy, symmetric=True):
    sd = dict(state_dict)
    packed, scale, zp = sd.pop("weight_packed"), sd["weight_scale"], sd.get("weight_zero_point")
    shape = torch.Size(sd["weight_shape"].tolist())
    if not symmetric and strategy in ("group", "channel"):
        zp = unpack_from_int32(zp, num_bits, (*shape[:-1], scale.shape[-1]), packed_dim=0)
        sd["weight_zero_point"] = zp
    sd["weight"] = dequantize(unpack_from_int32(packed, num_bits, shape), scale, zp)
    return sd


@torch.no_grad()
def int_quantized_compress(state_dict, *, num_bits=8, strategy="tensor", group_size=None, symmetric=True):
    sd = dict(state_dict)
    sd["weight"] = quantize(sd["weight"], sd["weight_scale"], sd.get("weight_zero_point"), num_bits=num_bits, strategy=strategy,
                            group_size=group_size, dtype=torch.int8)
    if symmetric:
        sd.pop("weight_zero_point", None)
    return sd


@torch.no_grad()
def int_quantized_decompress(state_dict):
    sd = dict(state_dict)
    sd["weight"] = dequantize(sd["weight"], sd["weight_scale"], sd.get("weight_zero_point"))
    return sd
