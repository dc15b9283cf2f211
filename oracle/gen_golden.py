"""Generate golden vectors by running the UPSTREAM REFERENCE itself (build container only).

Usage (from the repo root, in the container where /root/reference exists):
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Writes tests/golden/*.safetensors (+ manifest.json).  Each file holds the inputs and the
reference's outputs for one family of hot-path functions; tests/test_oracle_golden.py checks
the C oracle against them on CPU, and tests/test_gpu_parity.py checks the HIP kernels
against the same files on the GPU box (where /root/reference does not exist).

TEST INFRASTRUCTURE ONLY.  Nothing in the product imports this.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ref_import.import_reference()

from compressed_tensors.compressors import BaseCompressor  # noqa: E402
from compressed_tensors.compressors.pack_quantized.helpers import (  # noqa: E402
    pack_to_int32,
    unpack_from_int32,
)
from compressed_tensors.quantization import QuantizationArgs, QuantizationScheme  # noqa: E402
from compressed_tensors.quantization.lifecycle.forward import (  # noqa: E402
    dequantize,
    fake_quantize,
    quantize,
)
from compressed_tensors.quantization.utils import calculate_qparams  # noqa: E402
from compressed_tensors.utils.helpers import pack_bitmasks, unpack_bitmasks  # noqa: E402
from compressed_tensors.utils.permutations_24 import get_permutations_24  # noqa: E402
from compressed_tensors.utils.semi_structured_conversions import (  # noqa: E402
    mask_creator,
    sparse_semi_structured_from_dense_cutlass,
    sparse_semi_structured_to_dense_cutlass,
)
from safetensors.torch import save_file  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
manifest = {}


def save(name, tensors, meta):
    tensors = {k: v.contiguous().clone() for k, v in tensors.items()}
    save_file(tensors, os.path.join(OUT, name + ".safetensors"))
    manifest[name] = meta


def special_values(dtype):
    """values that stress the rounding model: NaN, infs, signed zero, ties, huge, tiny"""
    v = [float("nan"), float("inf"), -float("inf"), 0.0, -0.0, 1e30, -1e30, 1e-30, 2.498, 3.496,
         0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 6.5, 7.5, -7.5, -8.5, 127.5, -128.5, 65504.0, 1e-8]
    return torch.tensor(v, dtype=torch.float32).to(dtype)


# ----------------------------------------------------------------------------- pack / unpack
def gen_pack():
    g = torch.Generator().manual_seed(1234)
    tensors, cases = {}, []
    shapes = [(5, 33), (7, 100), (4, 1024), (3, 64), (1, 1), (2, 31), (16, 256)]
    for bits in range(1, 9):
        lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
        for shape in shapes:
            for pd in (1, 0):
                key = f"b{bits}_{shape[0]}x{shape[1]}_d{pd}"
                v = torch.randint(lo, hi + 1, shape, dtype=torch.int8, generator=g)
                p = pack_to_int32(v, bits, packed_dim=pd)
                u = unpack_from_int32(p, bits, torch.Size(shape), packed_dim=pd)
                assert torch.equal(u, v)
                tensors[key + ".value"] = v
                tensors[key + ".packed"] = p.contiguous()
                cases.append({"key": key, "bits": bits, "shape": list(shape), "packed_dim": pd})
        # 3-D (MoE) slice-wise packing
        key = f"b{bits}_3d"
        v = torch.randint(lo, hi + 1, (3, 8, 40), dtype=torch.int8, generator=g)
        tensors[key + ".value"] = v
        tensors[key + ".packed"] = pack_to_int32(v, bits).contiguous()
        cases.append({"key": key, "bits": bits, "shape": [3, 8, 40], "packed_dim": 1})
    # out-of-range int8 input (the reference does not mask before shifting; helpers.py:54-83)
    v = torch.randint(-128, 128, (4, 64), dtype=torch.int8, generator=g)
    for bits in (4, 8):
        key = f"b{bits}_oob"
        tensors[key + ".value"] = v
        tensors[key + ".packed"] = pack_to_int32(v, bits).contiguous()
        cases.append({"key": key, "bits": bits, "shape": [4, 64], "packed_dim": 1, "oob": True})
    save("pack", tensors, {"cases": cases})


# ----------------------------------------------------------------------------- quant / dequant
def make_qparams(x, args):
    """per-strategy min/max -> reference calculate_qparams (what a min-max observer does)"""
    st = str(getattr(args.strategy, "value", args.strategy))
    if st == "tensor":
        mn, mx = torch.aminmax(x)
    elif st == "channel":
        mn, mx = torch.aminmax(x, dim=-1, keepdim=True)
    elif st == "group":
        xg = x.unflatten(-1, (x.shape[-1] // args.group_size, args.group_size))
        mn, mx = torch.aminmax(xg, dim=-1)
    elif st == "block":
        bh, bw = args.block_structure
        xb = x.reshape(x.shape[0] // bh, bh, x.shape[1] // bw, bw)
        mn = xb.amin(dim=(1, 3))
        mx = xb.amax(dim=(1, 3))
    else:
        raise ValueError(st)
    return calculate_qparams(mn, mx, args)


def gen_quant():
    g = torch.Generator().manual_seed(4321)
    tensors, cases = {}, []
    configs = [
        # (name, kwargs for QuantizationArgs, shape)
        ("g128_b4_sym", dict(num_bits=4, strategy="group", group_size=128, symmetric=True), (8, 512)),
        ("g128_b4_asym", dict(num_bits=4, strategy="group", group_size=128, symmetric=False), (8, 512)),
        ("g32_b8_asym", dict(num_bits=8, strategy="group", group_size=32, symmetric=False), (6, 128)),
        ("g16_b3_sym", dict(num_bits=3, strategy="group", group_size=16, symmetric=True), (5, 64)),
        ("ch_b4_sym", dict(num_bits=4, strategy="channel", symmetric=True), (9, 200)),
        ("ch_b8_asym", dict(num_bits=8, strategy="channel", symmetric=False), (9, 200)),
        ("t_b8_sym", dict(num_bits=8, strategy="tensor", symmetric=True), (16, 96)),
        ("t_b4_asym", dict(num_bits=4, strategy="tensor", symmetric=False), (16, 96)),
        ("blk_b8_sym", dict(num_bits=8, strategy="block", block_structure=[4, 32], symmetric=True), (8, 128)),
    ]
    for name, kw, shape in configs:
        args = QuantizationArgs(**kw)
        for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
            for scale_dt_name, sdt in (("same", None), ("f32", torch.float32)):
                if dt is torch.float32 and sdt is not None:
                    continue
                key = f"{name}_{dt_name}_s{scale_dt_name}"
                x = torch.randn(shape, generator=g).mul(3.0).to(dt)
                sp = special_values(dt)
                x.view(-1)[: sp.numel()] = sp
                # qparams from the finite part only (NaN/inf would poison every scale)
                xf = torch.nan_to_num(x.float(), nan=0.0, posinf=4.0, neginf=-4.0).clamp(-9, 9).to(dt)
                scale, zp = make_qparams(xf, args)
                if sdt is not None:
                    scale = scale.to(sdt)
                if str(getattr(args.strategy, "value", args.strategy)) == "tensor" and scale_dt_name == "f32":
                    scale = scale.reshape(())  # 0-dim fp32 scale: result type stays x.dtype
                    zp = zp.reshape(())
                q8 = quantize(x, scale, zp, args, dtype=torch.int8)
                qf = quantize(x, scale, zp, args)
                fq = fake_quantize(x, scale, zp, args)
                dq = dequantize(q8, scale, zp, args=args)
                dq_inferred = dequantize(q8, scale, zp) if str(getattr(args.strategy, "value", args.strategy)) != "block" else dq
                q8_nozp = quantize(x, scale, None, args, dtype=torch.int8)
                tensors.update({
                    key + ".x": x, key + ".scale": scale, key + ".zp": zp, key + ".q8": q8,
                    key + ".qf": qf, key + ".fq": fq, key + ".dq": dq, key + ".dq_inferred": dq_inferred,
                    key + ".q8_nozp": q8_nozp,
                })
                cases.append({"key": key, "args": kw, "shape": list(shape)})
    # activation ordering (g_idx) on a group scheme
    args = QuantizationArgs(num_bits=4, strategy="group", group_size=32, symmetric=False, actorder="group")
    x = torch.randn((6, 128), generator=g).to(torch.bfloat16)
    perm = torch.randperm(128, generator=g)
    g_idx = (torch.arange(128, dtype=torch.int32) // 32)[perm].contiguous()
    xp = x.index_select(-1, torch.argsort(g_idx))
    scale, zp = make_qparams(xp, args)
    key = "gidx_b4_asym_bf16"
    tensors.update({
        key + ".x": x, key + ".scale": scale, key + ".zp": zp, key + ".g_idx": g_idx,
        key + ".q8": quantize(x, scale, zp, args, dtype=torch.int8, g_idx=g_idx),
        key + ".fq": fake_quantize(x, scale, zp, args, g_idx=g_idx),
    })
    tensors[key + ".dq"] = dequantize(tensors[key + ".q8"], scale, zp, args=args, g_idx=g_idx)
    cases.append({"key": key, "args": dict(num_bits=4, strategy="group", group_size=32, symmetric=False),
                  "shape": [6, 128], "g_idx": True})
    save("quant", tensors, {"cases": cases})


# ----------------------------------------------------------------------------- qparams
def gen_qparams():
    g = torch.Generator().manual_seed(99)
    tensors, cases = {}, []
    for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
        for bits in (4, 8):
            for sym in (True, False):
                for gs in (None, 32, 128):
                    key = f"{dt_name}_b{bits}_{'sym' if sym else 'asym'}_g{gs}"
                    x = torch.randn((8, 256), generator=g).mul(0.05).to(dt)
                    x[0, :128] = 0  # an all-zero group -> eps substitution path
                    x[1, :32] = x[1, :32].abs()  # strictly positive group (min clamps to 0)
                    kw = dict(num_bits=bits, symmetric=sym, strategy="group" if gs else "channel")
                    if gs:
                        kw["group_size"] = gs
                    args = QuantizationArgs(**kw)
                    scale, zp = make_qparams(x, args)
                    tensors.update({key + ".x": x, key + ".scale": scale, key + ".zp": zp})
                    cases.append({"key": key, "bits": bits, "symmetric": sym, "group_size": gs})
    save("qparams", tensors, {"cases": cases})


# ----------------------------------------------------------------------------- compressors
def gen_compressors():
    g = torch.Generator().manual_seed(777)
    tensors, cases = {}, []
    configs = [
        ("pq_g128_b4_sym_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True), (64, 256), torch.bfloat16),
        ("pq_g128_b4_asym_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=False), (64, 256), torch.bfloat16),
        ("pq_ch_b4_asym_f16", "pack-quantized", dict(num_bits=4, strategy="channel", symmetric=False), (40, 100), torch.float16),
        ("pq_ch_b8_sym_f32", "pack-quantized", dict(num_bits=8, strategy="channel", symmetric=True), (33, 70), torch.float32),
        ("pq_t_b3_asym_f32", "pack-quantized", dict(num_bits=3, strategy="tensor", symmetric=False), (17, 45), torch.float32),
        ("pq_g32_b5_asym_bf16", "pack-quantized", dict(num_bits=5, strategy="group", group_size=32, symmetric=False), (36, 96), torch.bfloat16),
        ("iq_t_b8_sym_bf16", "int-quantized", dict(num_bits=8, strategy="tensor", symmetric=True), (64, 128), torch.bfloat16),
        ("iq_ch_b8_asym_bf16", "int-quantized", dict(num_bits=8, strategy="channel", symmetric=False), (64, 128), torch.bfloat16),
        ("nq_g64_b8_sym_f16", "naive-quantized", dict(num_bits=8, strategy="group", group_size=64, symmetric=True), (16, 128), torch.float16),
    ]
    for key, fmt, kw, shape, dt in configs:
        args = QuantizationArgs(**kw)
        act = QuantizationArgs(num_bits=8, strategy="tensor", symmetric=True) if fmt == "int-quantized" else None
        scheme = QuantizationScheme(targets=["Linear"], weights=args, input_activations=act)
        w = torch.randn(shape, generator=g).to(dt)
        scale, zp = make_qparams(w, args)
        sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
        comp = BaseCompressor.get_value_from_registry(fmt)
        c = comp.compress(sd, scheme)
        d = comp.decompress(c, scheme)
        fq = fake_quantize(w, scale, zp, args)
        assert torch.equal(fq, d["weight"].to(fq.dtype))
        for k, v in sd.items():
            tensors[f"{key}.in.{k}"] = v
        for k, v in c.items():
            tensors[f"{key}.c.{k}"] = v
        for k, v in d.items():
            tensors[f"{key}.d.{k}"] = v
        cases.append({"key": key, "format": fmt, "args": kw, "shape": list(shape),
                      "compressed_keys": sorted(c.keys()), "decompressed_keys": sorted(d.keys())})
    save("compressors", tensors, {"cases": cases})


def _sha(t):
    import hashlib

    return hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()


def gen_compressors2():
    """round 3: the cases the first family lacked — activation ordering GROUP / WEIGHT through the class
    (tests/test_compressors/test_pack_quant.py:238-277), a 3-D (experts, rows, cols) weight through compress
    (pack_quantized/helpers.py:45-51), channel-symmetric int4, `block` for naive int8 with padding (naive_quantized/base.py:72-77)."""
    g = torch.Generator().manual_seed(4242)
    tensors, cases = {}, []
    configs = [
        ("pq_g128_b4_sym_actgroup_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True, actorder="group"), (64, 512), torch.bfloat16, True),
        ("pq_g128_b4_asym_actgroup_f16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=False, actorder="group"), (32, 512), torch.float16, True),
        ("pq_g32_b4_sym_actgroup_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=32, symmetric=True, actorder="group"), (40, 256), torch.bfloat16, True),
        ("pq_g128_b4_sym_actweight_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True, actorder="weight"), (64, 512), torch.bfloat16, True),
        ("pq_ch_b4_sym_bf16", "pack-quantized", dict(num_bits=4, strategy="channel", symmetric=True), (64, 256), torch.bfloat16, True),
        ("pq_g128_b4_sym_experts3d_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True), (4, 32, 256), torch.bfloat16, False),
        ("pq_g64_b8_asym_experts3d_f16", "pack-quantized", dict(num_bits=8, strategy="group", group_size=64, symmetric=False), (3, 16, 128), torch.float16, False),
        ("nq_block_b8_sym_bf16", "naive-quantized", dict(num_bits=8, strategy="block", block_structure=[128, 128], symmetric=True), (200, 300), torch.bfloat16, True),
        ("nq_block_b8_asym_f16", "naive-quantized", dict(num_bits=8, strategy="block", block_structure=[64, 32], symmetric=False), (128, 96), torch.float16, True),
    ]
    for key, fmt, kw, shape, dt, round_trip in configs:
        args = QuantizationArgs(**kw)
        scheme = QuantizationScheme(targets=["Linear"], weights=args)
        w = torch.randn(shape, generator=g).to(dt)
        if kw["strategy"] == "block":
            from compressed_tensors.quantization.utils import maybe_pad_tensor_for_block_quant

            scale, zp = make_qparams(maybe_pad_tensor_for_block_quant(w, tuple(kw["block_structure"])), args)
        else:
            scale, zp = make_qparams(w, args)
        sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
        if kw.get("actorder") == "group":
            cols = shape[-1]
            sd["weight_g_idx"] = (torch.randperm(cols, generator=g) // kw["group_size"]).to(torch.int32)
        comp = BaseCompressor.get_value_from_registry(fmt)
        c = comp.compress(dict(sd), scheme)
        d = {}
        if round_trip:
            d = comp.decompress(dict(c), scheme)
            if kw["strategy"] != "block":  # (the padded block grid has no unpadded fake_quantize counterpart)
                fq = fake_quantize(w, scale, zp, args, g_idx=sd.get("weight_g_idx"))
                assert torch.equal(fq, d["weight"].to(fq.dtype)), key
            assert d["weight"].shape == w.shape, key
        for k, v in sd.items():
            tensors[f"{key}.in.{k}"] = v
        for k, v in c.items():
            tensors[f"{key}.c.{k}"] = v
        for k, v in d.items():
            tensors[f"{key}.d.{k}"] = v
        cases.append({"key": key, "format": fmt, "args": kw, "shape": list(shape), "round_trip": round_trip,
                      "compressed_keys": sorted(c.keys()), "decompressed_keys": sorted(d.keys())})
    save("compressors2", tensors, {"cases": cases})


def gen_compressors_big():
    """one 1024 x 4096 case per format / scheme so that the reference-generated vectors reach the flat / lean / rows-per-workgroup
    kernels the small shapes do not all select.  To keep the fixtures small the weight is regenerated from its seed by the test
    (and checked against the sha256 recorded here) and the outputs are stored as sha256 digests of their bytes; only the scales /
    zero points / g_idx — what the reference's calculate_qparams produced — are stored as tensors."""
    tensors, cases = {}, []
    configs = [
        ("big_pq_g128_b4_sym_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True), torch.bfloat16),
        ("big_pq_g128_b4_asym_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=False), torch.bfloat16),
        ("big_pq_g128_b4_sym_f16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True), torch.float16),
        ("big_pq_g128_b4_sym_actgroup_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True, actorder="group"), torch.bfloat16),
        ("big_pq_g128_b4_asym_actgroup_bf16", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=False, actorder="group"), torch.bfloat16),
        ("big_pq_ch_b4_sym_bf16", "pack-quantized", dict(num_bits=4, strategy="channel", symmetric=True), torch.bfloat16),
        ("big_pq_g128_b8_sym_bf16", "pack-quantized", dict(num_bits=8, strategy="group", group_size=128, symmetric=True), torch.bfloat16),
        ("big_pq_g128_b3_asym_bf16", "pack-quantized", dict(num_bits=3, strategy="group", group_size=128, symmetric=False), torch.bfloat16),
        ("big_iq_t_b8_sym_bf16", "int-quantized", dict(num_bits=8, strategy="tensor", symmetric=True), torch.bfloat16),
        ("big_iq_ch_b8_asym_bf16", "int-quantized", dict(num_bits=8, strategy="channel", symmetric=False), torch.bfloat16),
        ("big_nq_g128_b8_sym_f16", "naive-quantized", dict(num_bits=8, strategy="group", group_size=128, symmetric=True), torch.float16),
        ("big_pq_g128_b4_sym_f32", "pack-quantized", dict(num_bits=4, strategy="group", group_size=128, symmetric=True), torch.float32),
    ]
    shape = (1024, 4096)
    for i, (key, fmt, kw, dt) in enumerate(configs):
        seed = 9000 + i
        args = QuantizationArgs(**kw)
        act = QuantizationArgs(num_bits=8, strategy="tensor", symmetric=True) if fmt == "int-quantized" else None
        scheme = QuantizationScheme(targets=["Linear"], weights=args, input_activations=act)
        w = torch.randn(shape, generator=torch.Generator().manual_seed(seed)).mul_(0.05).to(dt)
        scale, zp = make_qparams(w, args)
        sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
        if kw.get("actorder") == "group":
            sd["weight_g_idx"] = (torch.randperm(shape[1], generator=torch.Generator().manual_seed(seed + 500)) // kw["group_size"]).to(torch.int32)
        comp = BaseCompressor.get_value_from_registry(fmt)
        c = comp.compress(dict(sd), scheme)
        d = comp.decompress(dict(c), scheme)
        fq = fake_quantize(w, scale, zp, args, g_idx=sd.get("weight_g_idx"))
        assert torch.equal(fq, d["weight"].to(fq.dtype)), key
        for k, v in sd.items():
            if k != "weight":
                tensors[f"{key}.in.{k}"] = v
        cases.append({"key": key, "format": fmt, "args": kw, "shape": list(shape), "dtype": str(dt).split(".")[-1], "seed": seed,
                      "weight_sha256": _sha(w),
                      "compressed": {k: {"sha256": _sha(v), "shape": list(v.shape), "dtype": str(v.dtype).split(".")[-1]} for k, v in c.items()},
                      "decompressed": {k: {"sha256": _sha(v), "shape": list(v.shape), "dtype": str(v.dtype).split(".")[-1]} for k, v in d.items()}})
    save("compressors_big", tensors, {"cases": cases})


# ----------------------------------------------------------------------------- sparse primitives
def gen_sparse():
    g = torch.Generator().manual_seed(2024)
    tensors, cases = {}, []
    # bitmask primitives
    for shape in [(1, 10), (5, 64), (7, 100), (3, 8), (4, 129)]:
        key = f"bm_{shape[0]}x{shape[1]}"
        m = torch.rand(shape, generator=g) < 0.5
        p = pack_bitmasks(m)
        assert torch.equal(unpack_bitmasks(p, list(shape)), m)
        tensors[key + ".mask"] = m.to(torch.uint8)
        tensors[key + ".packed"] = p
        cases.append({"key": key, "kind": "bitmask", "shape": list(shape)})
    # cutlass 2:4
    for dt_name, dt, shape in (("bf16", torch.bfloat16, (64, 128)), ("f16", torch.float16, (64, 64)),
                               ("i8", torch.int8, (64, 128))):
        key = f"c24_{dt_name}"
        if dt is torch.int8:
            d = torch.randint(-8, 8, shape, generator=g, dtype=torch.int8)
            d = torch.where(d == 0, torch.ones_like(d), d)
        else:
            d = torch.randn(shape, generator=g).to(dt)
        mask = mask_creator(d.float()).bool()
        dm = d * mask.to(d.dtype)
        # a few degenerate quads: all-zero, single non-zero, three/four non-zero
        dm[0, 0:4] = 0
        dm[0, 4:8] = 0
        dm[0, 5] = 1
        dm[1, 0:4] = 1
        dm[1, 4:8] = 1
        dm[1, 6] = 0
        sparse, meta = sparse_semi_structured_from_dense_cutlass(dm)
        dense_rt = sparse_semi_structured_to_dense_cutlass(sparse, meta)
        tensors.update({key + ".raw": d, key + ".mask": mask.to(torch.uint8), key + ".dense": dm,
                        key + ".sparse": sparse.contiguous(), key + ".meta": meta.contiguous(),
                        key + ".dense_rt": dense_rt})
        cases.append({"key": key, "kind": "cutlass24", "shape": list(shape)})
    # marlin-24 permutation tables
    for bits in (4, 8):
        perm, scale_perm, scale_perm_single = get_permutations_24(bits)
        tensors[f"perm24_b{bits}.perm"] = perm.to(torch.int64)
        tensors[f"perm24_b{bits}.scale_perm"] = torch.tensor(scale_perm)
        tensors[f"perm24_b{bits}.scale_perm_single"] = torch.tensor(scale_perm_single)
        cases.append({"key": f"perm24_b{bits}", "kind": "perm24", "bits": bits})
    save("sparse", tensors, {"cases": cases})


# ----------------------------------------------------------------------------- FP4 (NVFP4 / MXFP4), SURVEY §8f N4
def gen_fp4():
    """nvfp4-pack-quantized / mxfp4-pack-quantized codecs: the reference's compress / decompress on weights that
    contain every rounding boundary of the E2M1 grid, plus the primitives (cast_to_fp4, pack / unpack, E8M0)."""
    from compressed_tensors.compressors.mx_utils import compress_mx_scale, decompress_mx_scale
    from compressed_tensors.compressors.nvfp4.helpers import pack_fp4_to_uint8, unpack_fp4_from_uint8
    from compressed_tensors.quantization.quant_args import FP4_E2M1_DATA
    from compressed_tensors.quantization.quant_scheme import preset_name_to_scheme
    from compressed_tensors.quantization.utils import generate_gparam

    g = torch.Generator().manual_seed(4242)
    tensors, cases = {}, []
    # primitives
    grid = torch.tensor([0.0, -0.0, 0.1, 0.25, 0.26, 0.5, 0.74, 0.75, 1.0, 1.25, 1.26, 1.5, 1.74, 1.75, 2.0, 2.5, 2.51, 3.0, 3.49,
                         3.5, 4.0, 5.0, 5.01, 6.0, -0.1, -0.25, -0.3, -0.75, -1.25, -1.75, -2.5, -3.5, -5.0, -6.0, 5.5, -5.5], dtype=torch.float32)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        x = torch.cat([grid, (torch.rand(92, generator=g) * 12 - 6)]).to(dt)
        tensors[f"cast_{dt}.in"] = x
        tensors[f"cast_{dt}.out"] = FP4_E2M1_DATA.cast_to_fp4(x.clone())
    vals = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
    q = torch.cat([vals, -vals]).repeat(8).reshape(8, 16).to(torch.bfloat16)
    tensors["pack.in"] = q
    tensors["pack.out"] = pack_fp4_to_uint8(q)
    tensors["unpack.out"] = unpack_fp4_from_uint8(tensors["pack.out"], 8, 16)
    sc = torch.tensor([2.0 ** e for e in range(-20, 12)] + [3.0, 0.7, 1.99, 1e-3], dtype=torch.float32)
    tensors["e8m0.in"] = sc
    tensors["e8m0.out"] = compress_mx_scale(sc, torch.uint8)
    tensors["e8m0.back"] = decompress_mx_scale(tensors["e8m0.out"])

    for preset, fmt, gs in (("NVFP4A16", "nvfp4-pack-quantized", 16), ("MXFP4A16", "mxfp4-pack-quantized", 32)):
        scheme = preset_name_to_scheme(preset, ["Linear"])
        args = scheme.weights
        comp = BaseCompressor.get_value_from_registry(fmt)
        for dt in (torch.bfloat16, torch.float16):
            for shape in ((8, 64), (16, 256)):
                key = f"{fmt}_{str(dt).split('.')[-1]}_{shape[0]}x{shape[1]}"
                w = torch.randn(shape, generator=g, dtype=torch.float32)
                w[0, :8] = torch.tensor([0.0, -0.0, 1e-6, -1e-6, 3.0, -3.0, 100.0, -100.0])
                w = w.to(dt)
                wg = w.reshape(shape[0], shape[1] // gs, gs)
                mn, mx = wg.amin(-1), wg.amax(-1)
                sd = {"weight": w}
                if preset.startswith("NV"):
                    gsc = generate_gparam(w.min().reshape(1), w.max().reshape(1))
                    sd["weight_global_scale"] = gsc
                    scale, zp = calculate_qparams(mn, mx, args, global_scale=gsc)
                else:
                    scale, zp = calculate_qparams(mn, mx, args)
                sd["weight_scale"] = scale
                sd["weight_zero_point"] = zp
                # put exact rounding boundaries of the first group into the weight: k * effective scale
                eff = (scale[0, 0].float() / sd["weight_global_scale"][0].float()) if "weight_global_scale" in sd else scale[0, 0].float()
                sd["weight"][0, 8:16] = (torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, -2.5]) * eff).to(dt)
                c = comp.compress(sd, scheme)
                d = comp.decompress(c, scheme)
                tensors[key + ".in.weight"] = sd["weight"]
                tensors[key + ".in.weight_scale"] = scale
                if "weight_global_scale" in sd:
                    tensors[key + ".in.weight_global_scale"] = sd["weight_global_scale"]
                for k, v in c.items():
                    tensors[key + ".comp." + k] = v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v
                for k, v in d.items():
                    tensors[key + ".dec." + k] = v.data if isinstance(v, torch.nn.Parameter) else v
                cases.append({"key": key, "format": fmt, "group_size": gs, "shape": list(shape), "dtype": str(dt).split(".")[-1],
                              "scale_dtype": str(scale.dtype).split(".")[-1], "compressed_keys": sorted(c), "decompressed_keys": sorted(d),
                              "compressed_scale_dtype": str(c["weight_scale"].dtype).split(".")[-1]})
    save("fp4", tensors, {"cases": cases})


# ----------------------------------------------------------------------------- FP8 (float-quantized, mxfp8-quantized)
def gen_fp8():
    """FLOAT 8-bit (float8_e4m3fn) quantize / dequantize / fake_quantize for every strategy, and the float-quantized,
    naive-quantized(float) and mxfp8-quantized codecs.  Weights carry NaN / inf / signed zeros / the 448 clamp edge."""
    g = torch.Generator().manual_seed(8888)
    tensors, cases = {}, []
    configs = [
        ("t", dict(strategy="tensor"), (16, 96)),
        ("ch", dict(strategy="channel"), (9, 200)),
        ("g128", dict(strategy="group", group_size=128), (8, 512)),
        ("g32", dict(strategy="group", group_size=32), (6, 128)),
        ("blk", dict(strategy="block", block_structure=[4, 32]), (8, 128)),
    ]
    for name, kw0, shape in configs:
        kw = dict(num_bits=8, type="float", symmetric=True, **kw0)
        args = QuantizationArgs(**kw)
        for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
            for scale_dt_name, sdt in (("same", None), ("f32", torch.float32)):
                if dt is torch.float32 and sdt is not None:
                    continue
                key = f"{name}_{dt_name}_s{scale_dt_name}"
                x = torch.randn(shape, generator=g).mul(3.0).to(dt)
                sp = special_values(dt)
                x.view(-1)[: sp.numel()] = sp
                xf = torch.nan_to_num(x.float(), nan=0.0, posinf=4.0, neginf=-4.0).clamp(-9, 9).to(dt)
                scale, zp = make_qparams(xf, args)
                if sdt is not None:
                    scale = scale.to(sdt)
                if name == "t" and scale_dt_name == "f32":
                    scale, zp = scale.reshape(()), zp.reshape(())
                # put values that land exactly on fp8 ties and on the clamp edge into the first row / group
                s0 = scale.reshape(-1)[0].float()
                edge = torch.tensor([448.0, 449.0, 464.0, 480.0, -448.0, -500.0, 17.0, 18.0, 19.0, 0.0009765625, 0.00292969, 2.0 ** -10, 1.0625, 1.1875, -1.0625, 240.0])
                x.view(-1)[sp.numel(): sp.numel() + edge.numel()] = (edge * s0).to(dt)
                q = quantize(x, scale, zp, args, dtype=args.pytorch_dtype())
                qf = quantize(x, scale, zp, args)
                fq = fake_quantize(x, scale, zp, args)
                dq = dequantize(q, scale, zp, args=args)
                dq_inferred = dequantize(q, scale, zp) if name != "blk" else dq
                q_nozp = quantize(x, scale, None, args, dtype=args.pytorch_dtype())
                tensors.update({
                    key + ".x": x, key + ".scale": scale, key + ".zp": zp.view(torch.uint8) if zp.dtype == torch.float8_e4m3fn else zp,
                    key + ".q": q.view(torch.uint8), key + ".qf": qf, key + ".fq": fq, key + ".dq": dq, key + ".dq_inferred": dq_inferred,
                    key + ".q_nozp": q_nozp.view(torch.uint8),
                })
                cases.append({"key": key, "args": kw0, "shape": list(shape), "zp_dtype": str(zp.dtype).split(".")[-1]})

    # codecs
    from compressed_tensors.quantization.quant_scheme import preset_name_to_scheme

    codec_cases = []
    act = QuantizationArgs(num_bits=8, type="float", strategy="tensor", symmetric=True, dynamic=False)
    for key, fmt, wkw, shape, dt in (
        ("fq_ch_bf16", "float-quantized", dict(strategy="channel"), (64, 128), torch.bfloat16),
        ("fq_t_f16", "float-quantized", dict(strategy="tensor"), (32, 96), torch.float16),
        ("fq_blk_bf16", "float-quantized", dict(strategy="block", block_structure=[128, 128]), (300, 400), torch.bfloat16),
        ("nq_g64_f32", "naive-quantized", dict(strategy="group", group_size=64), (16, 128), torch.float32),
    ):
        args = QuantizationArgs(num_bits=8, type="float", symmetric=True, **wkw)
        scheme = QuantizationScheme(targets=["Linear"], weights=args, input_activations=act if fmt == "float-quantized" else None)
        w = torch.randn(shape, generator=g).to(dt)
        if wkw.get("strategy") == "block":
            bh, bw = wkw["block_structure"]
            rows_p, cols_p = -(-shape[0] // bh) * bh, -(-shape[1] // bw) * bw
            wp = torch.zeros((rows_p, cols_p), dtype=dt)
            wp[: shape[0], : shape[1]] = w
            scale, zp = make_qparams(wp, args)
        else:
            scale, zp = make_qparams(w, args)
        sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
        comp = BaseCompressor.get_value_from_registry(fmt)
        cdict = comp.compress(sd, scheme)
        ddict = comp.decompress(cdict, scheme)
        for k, v in sd.items():
            tensors[f"{key}.in.{k}"] = v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v
        for k, v in cdict.items():
            tensors[f"{key}.c.{k}"] = v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v
        for k, v in ddict.items():
            tensors[f"{key}.d.{k}"] = v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v
        codec_cases.append({"key": key, "format": fmt, "args": wkw, "shape": list(shape), "dtype": str(dt).split(".")[-1],
                            "compressed_keys": sorted(cdict), "decompressed_keys": sorted(ddict),
                            "compressed_dtypes": {k: str(v.dtype).split(".")[-1] for k, v in cdict.items()},
                            "decompressed_dtypes": {k: str(v.dtype).split(".")[-1] for k, v in ddict.items()}})
    # mxfp8: group 32, E8M0 scales
    scheme = preset_name_to_scheme("MXFP8A16", ["Linear"])
    args = scheme.weights
    comp = BaseCompressor.get_value_from_registry("mxfp8-quantized")
    for dt in (torch.bfloat16, torch.float16):
        for shape in ((8, 64), (16, 256)):
            key = f"mxfp8_{str(dt).split('.')[-1]}_{shape[0]}x{shape[1]}"
            w = torch.randn(shape, generator=g).to(dt)
            w[0, :4] = torch.tensor([0.0, -0.0, 1e-6, 300.0]).to(dt)
            wg = w.reshape(shape[0], shape[1] // 32, 32)
            scale, zp = calculate_qparams(wg.amin(-1), wg.amax(-1), args)
            sd = {"weight": w, "weight_scale": scale, "weight_zero_point": zp}
            cdict = comp.compress(sd, scheme)
            ddict = comp.decompress(cdict, scheme)
            for k, v in sd.items():
                tensors[f"{key}.in.{k}"] = v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v
            for k, v in cdict.items():
                tensors[f"{key}.c.{k}"] = v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v
            for k, v in ddict.items():
                tensors[f"{key}.d.{k}"] = v.view(torch.uint8) if v.dtype == torch.float8_e4m3fn else v
            codec_cases.append({"key": key, "format": "mxfp8-quantized", "args": dict(strategy="group", group_size=32), "shape": list(shape),
                                "dtype": str(dt).split(".")[-1], "compressed_keys": sorted(cdict), "decompressed_keys": sorted(ddict),
                                "scale_dtype": str(scale.dtype).split(".")[-1],
                                "compressed_dtypes": {k: str(v.dtype).split(".")[-1] for k, v in cdict.items()},
                                "decompressed_dtypes": {k: str(v.dtype).split(".")[-1] for k, v in ddict.items()}})
    save("fp8", tensors, {"cases": cases, "codecs": codec_cases})


# ----------------------------------------------------------------------------- FLOAT 4-bit quantize / dequantize / fake_quantize
def gen_fp4q():
    """quantize / dequantize / fake_quantize with FLOAT 4-bit args (the QDQ form of the NVFP4 / MXFP4 schemes): with and
    without a global scale, group / tensor_group / channel / tensor strategies."""
    from compressed_tensors.quantization.utils import generate_gparam

    g = torch.Generator().manual_seed(4444)
    tensors, cases = {}, []
    configs = [
        ("nv_tg16", dict(strategy="tensor_group", group_size=16, scale_dtype=torch.float8_e4m3fn, zp_dtype=torch.float8_e4m3fn), (8, 64), True),
        ("mx_g32", dict(strategy="group", group_size=32, scale_dtype=torch.uint8, zp_dtype=torch.uint8), (6, 128), False),
        ("g16_plain", dict(strategy="group", group_size=16), (5, 64), False),
        ("ch", dict(strategy="channel"), (9, 40), False),
        ("t", dict(strategy="tensor"), (4, 24), False),
    ]
    for name, kw0, shape, use_gs in configs:
        kw = dict(num_bits=4, type="float", symmetric=True, **kw0)
        args = QuantizationArgs(**kw)
        for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
            key = f"{name}_{dt_name}"
            x = torch.randn(shape, generator=g).mul(2.0).to(dt)
            sp = special_values(dt)
            x.view(-1)[: sp.numel()] = sp[: x.numel()]
            xf = torch.nan_to_num(x.float(), nan=0.0, posinf=4.0, neginf=-4.0).clamp(-9, 9).to(dt)
            gs = generate_gparam(xf.min().reshape(1), xf.max().reshape(1)) if use_gs else None
            st = kw0["strategy"]
            if st in ("group", "tensor_group"):
                xg = xf.unflatten(-1, (shape[1] // kw0["group_size"], kw0["group_size"]))
                mn, mx = torch.aminmax(xg, dim=-1)
            elif st == "channel":
                mn, mx = torch.aminmax(xf, dim=-1, keepdim=True)
            else:
                mn, mx = torch.aminmax(xf)
            scale, zp = calculate_qparams(mn, mx, args, global_scale=gs)
            s0 = (scale.reshape(-1)[0].float() / (gs[0].float() if gs is not None else 1.0))
            edge = torch.tensor([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0, 6.0, 7.0, -0.25, -0.2, -2.5, 0.26, 0.74, -6.5, 1e-4])
            n0 = min(sp.numel(), x.numel() - edge.numel())
            x.view(-1)[n0: n0 + edge.numel()] = (edge * s0).to(dt)
            qf = quantize(x, scale, zp, args, global_scale=gs)
            qf_nozp = quantize(x, scale, None, args, global_scale=gs)
            fq = fake_quantize(x, scale, zp, args, global_scale=gs)
            dq = dequantize(qf, scale, zp, args=args, global_scale=gs)
            tensors.update({key + ".x": x, key + ".scale": scale.view(torch.uint8) if scale.dtype == torch.float8_e4m3fn else scale,
                            key + ".zp": zp.view(torch.uint8) if zp.dtype == torch.float8_e4m3fn else zp,
                            key + ".qf": qf, key + ".qf_nozp": qf_nozp, key + ".fq": fq, key + ".dq": dq})
            if gs is not None:
                tensors[key + ".gs"] = gs
            cases.append({"key": key, "args": {k: v for k, v in kw0.items() if not k.endswith("_dtype")}, "shape": list(shape), "global_scale": use_gs,
                          "scale_dtype": str(scale.dtype).split(".")[-1], "zp_dtype": str(zp.dtype).split(".")[-1]})
    save("fp4q", tensors, {"cases": cases})


# ----------------------------------------------------------------------------- qparams of the FLOAT schemes
def gen_qparams_float():
    """calculate_qparams for FLOAT args from per-group min / max (what a min-max observer feeds it): FP8 (channel,
    group 128), NVFP4 (tensor_group 16 under generate_gparam's global scale), MXFP4 / MXFP8 (group 32, E8M0 scales)."""
    from compressed_tensors.quantization.utils import generate_gparam

    g = torch.Generator().manual_seed(555)
    tensors, cases = {}, []
    configs = [
        ("fp8_ch", dict(num_bits=8, type="float", strategy="channel", symmetric=True), None, False),
        ("fp8_g128", dict(num_bits=8, type="float", strategy="group", group_size=128, symmetric=True), 128, False),
        ("nvfp4", dict(num_bits=4, type="float", strategy="tensor_group", group_size=16, symmetric=True, scale_dtype=torch.float8_e4m3fn,
                       zp_dtype=torch.float8_e4m3fn), 16, True),
        ("mxfp4", dict(num_bits=4, type="float", strategy="group", group_size=32, symmetric=True, scale_dtype=torch.uint8, zp_dtype=torch.uint8), 32, False),
        ("mxfp8", dict(num_bits=8, type="float", strategy="group", group_size=32, symmetric=True, scale_dtype=torch.uint8, zp_dtype=torch.uint8), 32, False),
    ]
    for name, kw, gs, use_g in configs:
        args = QuantizationArgs(**kw)
        for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)):
            key = f"{name}_{dt_name}"
            x = torch.randn((8, 256), generator=g).mul(0.05).to(dt)
            x[0, :128] = 0                       # all-zero groups -> eps substitution
            x[1, :32] = x[1, :32].abs()          # strictly positive group
            x[2, :32] = (torch.rand(32, generator=g) * 1e-6).to(dt)   # tiny values
            x[3, :32] = (torch.randn(32, generator=g) * 3000).to(dt)  # large values
            x[4, 0:16] = torch.tensor([0.74, 0.75, 0.76, 1.24, 1.25, 1.26, 1.49, 1.5, 1.51, 1.74, 1.75, 1.76, 1.99, 2.0, 3.0, 0.0]).to(dt)  # power-of-two rounding
            if gs:
                xg = x.unflatten(-1, (256 // gs, gs))
                mn, mx = torch.aminmax(xg, dim=-1)
            else:
                mn, mx = torch.aminmax(x, dim=-1, keepdim=True)
            gsc = generate_gparam(x.min().reshape(1), x.max().reshape(1)) if use_g else None
            scale, zp = calculate_qparams(mn, mx, args, global_scale=gsc)
            tensors.update({key + ".x": x, key + ".scale": scale, key + ".zp": zp.view(torch.uint8) if zp.dtype == torch.float8_e4m3fn else zp})
            if gsc is not None:
                tensors[key + ".gs"] = gsc
            cases.append({"key": key, "kind": name, "group_size": gs, "scale_dtype": str(scale.dtype).split(".")[-1], "zp_dtype": str(zp.dtype).split(".")[-1]})
    save("qparams_float", tensors, {"cases": cases})


if __name__ == "__main__":
    torch.manual_seed(0)
    families = {"pack": gen_pack, "quant": gen_quant, "qparams": gen_qparams, "compressors": gen_compressors, "compressors2": gen_compressors2, "compressors_big": gen_compressors_big,
                "sparse": gen_sparse,
                "fp4": gen_fp4, "fp8": gen_fp8, "fp4q": gen_fp4q, "qparams_float": gen_qparams_float}
    wanted = sys.argv[1:] or list(families)  # `python oracle/gen_golden.py fp4` regenerates one family only
    mpath = os.path.join(OUT, "manifest.json")
    if os.path.exists(mpath) and sys.argv[1:]:
        with open(mpath) as f:
            manifest.update(json.load(f))
    for name in wanted:
        families[name]()
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(OUT, n)) for n in os.listdir(OUT))
    print(f"wrote {len(wanted)} golden families ({len(manifest)} in the manifest), {total/1e6:.2f} MB -> {OUT}")
