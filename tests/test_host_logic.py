"""CPU-only tests: the C-ABI library loads and exports every symbol the header declares, and
the host-side mirror of the reference interface (registry, format inference, argument
validation, meta-device paths, work partitioning) behaves like the reference.  No kernel is
launched here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cta():
    import __graft_entry__ as g

    g.build_hip()  # no-op when the library is up to date
    import compressed_tensors_amd as m

    return m


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "ct_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ct_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(cta):
    from compressed_tensors_amd import _lib

    declared = _header_symbols()
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/ct_hip.h but not exported"
    # the Python binding covers exactly the declared ABI
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    assert _lib.load().ct_abi_version() == 2


def test_ctypes_prototypes_match_the_header():
    """every entry point's parameter list in include/ct_hip.h against the ctypes prototype in _lib.py: same arity, and pointers /
    64-bit integers / ints in the same positions (an ABI drift between the two would corrupt arguments silently)"""
    import ctypes as C

    from compressed_tensors_amd import _lib

    text = open(os.path.join(ROOT, "include", "ct_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    decls = dict(re.findall(r"\b(?:int64_t|int|const char\*)\s+(ct_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert set(decls) == set(_lib._PROTOTYPES)

    def kind(param):
        p = " ".join(param.split())
        if "*" in p or p.startswith("ct_stream_t"):
            return "ptr"
        if p.startswith(("int64_t", "long long")):
            return "i64"
        if p.startswith(("uint32_t", "unsigned")):
            return "u32"
        if p.startswith("int"):
            return "i32"
        raise AssertionError(f"unrecognised parameter {param!r}")

    ckind = {C.c_void_p: "ptr", C.c_char_p: "ptr", C.c_int64: "i64", C.c_int: "i32", C.c_uint32: "u32"}
    for name, params in decls.items():
        params = [q for q in (p.strip() for p in params.split(",")) if q and q != "void"]
        argtypes, _ = _lib._PROTOTYPES[name]
        assert len(params) == len(argtypes), (name, len(params), len(argtypes))
        assert [kind(q) for q in params] == [ckind[a] for a in argtypes], name


def test_library_targets_gfx950_only():
    out = subprocess.run(["strings", "-a", os.path.join(ROOT, "compressed_tensors_amd", "libct_hip.so")],
                         capture_output=True, text=True).stdout
    archs = set(re.findall(r"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", out))
    assert archs == {"gfx950"}, archs


def test_no_product_import_of_the_oracle():
    """the product must never import/link/execute anything under oracle/"""
    pkg = os.path.join(ROOT, "compressed_tensors_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "libct_oracle" not in text and "ct_oracle.c" not in text, f


def test_missing_extension_fails_loudly(cta, monkeypatch):
    from compressed_tensors_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libct_hip.so")
    with pytest.raises(_lib.HipExtensionMissing):
        _lib.load()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_compute_without_gpu_raises(cta):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cta.codec.pack_to_int32(torch.zeros(2, 32, dtype=torch.int8), 4)


def test_argument_errors_match_reference(cta):
    """reference compressors/pack_quantized/helpers.py:36-42,122-130; forward_helpers.py:141-145"""
    with pytest.raises(ValueError, match="torch.int8"):
        cta.codec.pack_to_int32(torch.zeros(2, 2, dtype=torch.int32), 4)
    with pytest.raises(ValueError, match=r"num_bits in \[1, 8\]"):
        cta.codec.pack_to_int32(torch.zeros(2, 2, dtype=torch.int8), 9)
    with pytest.raises(ValueError, match="Aborting unpack"):
        cta.codec.unpack_from_int32(torch.zeros(2, 2, dtype=torch.int8), 4, (2, 2))
    with pytest.raises(ValueError, match="divisble"):
        cta.codec.QuantLayout((4, 100), torch.ones(4, 1), "group", group_size=64)
    with pytest.raises(NotImplementedError):
        args = cta.QuantizationArgs(num_bits=6, type="float")  # FLOAT types exist for 4 and 8 bits only (quant_args.py:479)
        cta.quantize(torch.zeros(2, 2), torch.ones(1), None, args)
    with pytest.raises(ValueError, match="Could not infer"):
        from compressed_tensors_amd.codec import infer_dequant_layout

        infer_dequant_layout((2, 2, 2), torch.ones(2, 2, 2))


def test_layout_resolution(cta):
    L = cta.codec.QuantLayout
    g = L((16, 256), torch.ones(16, 2), "group", group_size=128)
    assert (g.rows, g.cols, g.rdiv, g.cdiv, g.scale_cols) == (16, 256, 1, 128, 2)
    g1 = L((16, 256), torch.ones(1, 2), "group", group_size=128)
    assert g1.rdiv == 16
    c = L((16, 256), torch.ones(16, 1), "channel")
    assert (c.rdiv, c.cdiv, c.scale_cols) == (1, 256, 1)
    t = L((16, 256), torch.ones(1), "tensor")
    assert (t.rdiv, t.cdiv, t.scale_cols) == (16, 256, 1) and not t.scale_zero_dim
    t0 = L((16, 256), torch.tensor(1.0), "tensor")
    assert t0.scale_zero_dim
    b = L((16, 256), torch.ones(4, 4), "block", block_structure=[4, 64])
    assert (b.rdiv, b.cdiv, b.scale_cols) == (4, 64, 4)
    moe = L((3, 8, 256), torch.ones(3, 8, 2), "group", group_size=128)
    assert (moe.rows, moe.cols, moe.rdiv) == (24, 256, 1)
    # activation ordering: column -> group of its rank in the sorted order
    g_idx = torch.tensor([1, 0, 1, 0], dtype=torch.int32)
    a = L((2, 4), torch.ones(2, 2), "group", group_size=2, g_idx=g_idx)
    assert a.col_group.tolist() == [1, 0, 1, 0]
    from compressed_tensors_amd.codec import _result_dtype

    x = torch.zeros(2, 2, dtype=torch.bfloat16)
    assert _result_dtype(x, torch.tensor(1.0), True) == torch.bfloat16  # 0-dim fp32 scale does not promote
    assert _result_dtype(x, torch.ones(1), False) == torch.float32
    assert _result_dtype(x, torch.ones(1, dtype=torch.float16), False) == torch.float32


def test_registry_and_format_inference(cta):
    B = cta.BaseCompressor
    assert B.get_value_from_registry("pack-quantized") is cta.PackedQuantizationCompressor
    assert B.get_value_from_registry("pack_quantized") is cta.PackedQuantizationCompressor  # standardised
    assert B.get_value_from_registry("int-quantized") is cta.IntQuantizationCompressor
    with pytest.raises(KeyError):
        B.get_value_from_registry("no-such-format")
    with pytest.raises(RuntimeError):  # registry.py:215-223
        B.register(name="pack-quantized")(type("Other", (B,), {}))
    from compressed_tensors_amd.compressors import infer_module_format

    w4 = cta.QuantizationScheme(weights=cta.QuantizationArgs(num_bits=4, group_size=128))
    w8a8 = cta.QuantizationScheme(weights=cta.QuantizationArgs(num_bits=8), input_activations=cta.QuantizationArgs(num_bits=8))
    dense = cta.QuantizationScheme()
    assert infer_module_format(torch.nn.Linear, w4) == cta.CompressionFormat.pack_quantized
    assert infer_module_format(torch.nn.Linear, w8a8) == cta.CompressionFormat.int_quantized
    assert infer_module_format(torch.nn.Embedding, w4) == cta.CompressionFormat.pack_quantized
    assert infer_module_format(torch.nn.Linear, dense) == cta.CompressionFormat.dense
    assert infer_module_format(torch.nn.Conv2d, w4) == cta.CompressionFormat.dense
    names = cta.PackedQuantizationCompressor.compression_param_names(
        cta.QuantizationScheme(weights=cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=False, actorder="group")))
    assert names == ("weight_packed", "weight_scale", "weight_shape", "weight_zero_point", "weight_g_idx")


def test_meta_device_paths(cta):
    """reference compressors/pack_quantized/base.py:86-94,138-144: shapes only, no kernels"""
    scheme = cta.QuantizationScheme(weights=cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True))
    sd = {"weight": torch.empty(64, 100, device="meta", dtype=torch.bfloat16),
          "weight_scale": torch.empty(64, 1, device="meta", dtype=torch.bfloat16),
          "weight_zero_point": torch.empty(64, 1, device="meta", dtype=torch.int8)}
    c = cta.PackedQuantizationCompressor.compress(sd, scheme)
    assert c["weight_packed"].shape == (64, 13) and c["weight_packed"].device.type == "meta"
    assert c["weight_shape"].tolist() == [64, 100] and "weight_zero_point" not in c and "weight" not in c
    d = cta.PackedQuantizationCompressor.decompress(c, scheme)
    assert d["weight"].shape == (64, 100) and d["weight"].dtype == torch.bfloat16 and d["weight"].device.type == "meta"


def test_module_state_dict_glue(cta):
    from compressed_tensors_amd.utils import get_direct_state_dict, replace_direct_state_dict

    lin = torch.nn.Linear(4, 4)
    sd = get_direct_state_dict(lin)
    assert set(sd) == {"weight", "bias"} and not isinstance(sd["weight"], torch.nn.Parameter)
    new = {"bias": sd["bias"], "weight_packed": torch.zeros(4, 1, dtype=torch.int32)}
    lin.register_buffer("running", torch.ones(2))
    new["running"] = lin._buffers["running"]  # plain-tensor buffers are returned by identity -> untouched
    bias_ptr = lin.bias.data_ptr()
    replace_direct_state_dict(lin, new)
    # Parameters are always re-wrapped (`.data` is a fresh object, as upstream utils/module.py:56-65);
    # the storage is shared, and everything becomes non-trainable
    assert not hasattr(lin, "weight") and lin.bias.data_ptr() == bias_ptr and not lin.bias.requires_grad
    assert "running" in lin._buffers and lin._buffers["running"] is new["running"]
    assert isinstance(lin.weight_packed, torch.nn.Parameter) and not lin.weight_packed.requires_grad


def _module_state(m):
    return ({k: (None if v is None else (type(v).__name__, v.requires_grad, v.data_ptr(), tuple(v.shape), v.dtype)) for k, v in m._parameters.items()},
            {k: (None if v is None else (type(v).__name__, v.data_ptr())) for k, v in m._buffers.items()}, sorted(m._non_persistent_buffers_set))


@pytest.mark.parametrize("hooked", [False, True], ids=["plain", "custom_setattr"])
def test_swap_direct_entries_equals_replace_direct_state_dict(cta, hooked):
    """the delta form the batched module paths use leaves a module exactly as upstream's full replacement does (utils/module.py:33-65):
    trainable parameters end non-trainable, torch.nn.Buffer entries become Parameters, plain-tensor buffers stay buffers, None
    entries stay, added names overwrite parameters and buffers alike — for a plain nn.Module (direct dictionary writes) and for a
    class with its own __setattr__ (the generic path)"""
    import copy

    from compressed_tensors_amd.utils.module import get_direct_state_dict, replace_direct_state_dict, swap_direct_entries

    class Odd(torch.nn.Linear):
        def __setattr__(self, name, value):
            super().__setattr__(name, value)

    def build():
        lin = (Odd if hooked else torch.nn.Linear)(8, 4, bias=False)  # bias registered as None
        lin.weight_scale = torch.nn.Parameter(torch.ones(4, 1), requires_grad=True)
        lin.register_buffer("weight_zero_point", torch.zeros(4, 1, dtype=torch.int8))
        lin.register_buffer("plain_buf", torch.ones(3), persistent=False)
        if hasattr(torch.nn, "Buffer"):
            lin.wrapped_buf = torch.nn.Buffer(torch.ones(2))
        lin.weight_g_idx = torch.nn.Parameter(torch.arange(8, dtype=torch.int32), requires_grad=False)

        class Tagged(torch.nn.Parameter):  # a Parameter SUBCLASS that stays and is trainable: upstream re-creates it as a plain Parameter
            pass

        lin.tagged = Tagged(torch.ones(2), requires_grad=True)
        return lin

    for remove, add_fn in (
        (("weight", "weight_zero_point"), lambda: {"weight_packed": torch.zeros(4, 1, dtype=torch.int32), "weight_shape": torch.tensor([4, 8])}),
        (("weight",), lambda: {"weight_packed": torch.zeros(4, 1, dtype=torch.int32), "weight_zero_point": torch.zeros(1, 1, dtype=torch.int32)}),
        ((), lambda: {"plain_buf": torch.zeros(5), "weight_scale": torch.zeros(4, 2)}),
    ):
        a = build()
        b = copy.deepcopy(a)
        add = add_fn()
        new = {k: v for k, v in get_direct_state_dict(a).items() if k not in remove}
        new.update(add)
        replace_direct_state_dict(a, new)
        swap_direct_entries(b, remove, add, status="compressed")
        sa, sb = _module_state(a), _module_state(b)
        # same names, kinds, trainability, shapes and dtypes; storage differs between the two copies, except for the added tensors
        strip = lambda st: ({k: v and (v[0], v[1], v[3], v[4]) for k, v in st[0].items()}, {k: v and v[0] for k, v in st[1].items()}, st[2])
        assert strip(sa) == strip(sb)
        assert type(a._parameters["tagged"]) is torch.nn.Parameter and type(b._parameters["tagged"]) is torch.nn.Parameter and not b._parameters["tagged"].requires_grad
        assert list(a._parameters) == list(b._parameters) or set(a._parameters) == set(b._parameters)
        for k, v in add.items():
            assert b._parameters[k].data_ptr() == v.data_ptr() and a._parameters[k].data_ptr() == v.data_ptr()
        assert b.quantization_status == "compressed"


def test_impl_backend_dispatch(cta, monkeypatch):
    from compressed_tensors_amd.utils.impl_backend import ImplBackend

    calls = []

    @ImplBackend.register("demo_op", req=lambda x: x > 0, priority=1)
    def demo_fast(x):
        calls.append("fast")
        return x * 2

    @ImplBackend.entrypoint("demo_op")
    def demo_op(x):
        calls.append("eager")
        return x

    assert demo_op(3) == 6 and demo_op(-1) == -1 and calls == ["fast", "eager"]
    assert ImplBackend.call("demo_fast", 5) == 10
    with pytest.raises(ValueError):
        ImplBackend.register("demo_op", req=lambda x: True, priority=0)(demo_fast)


def test_greedy_bin_packing_matches_reference_rule(cta):
    from compressed_tensors_amd.distributed import greedy_bin_packing, shard_items, shard_rows

    items = list("abcdefg")
    w = dict(zip(items, [7, 3, 5, 5, 2, 9, 1]))
    sorted_items, bins, where = greedy_bin_packing(items, 3, w.__getitem__)
    assert [w[i] for i in sorted_items] == sorted(w.values(), reverse=True)
    loads = [sum(w[i] for i in b) for b in bins]
    assert sum(loads) == sum(w.values()) and max(loads) - min(loads) <= 2
    assert all(i in bins[where[i]] for i in items)
    # TinyLlama-1.1B linear shapes: 154 modules over 8 ranks balance to <= one k_proj
    shapes = ([(2048, 2048)] * 2 + [(256, 2048)] * 2 + [(5632, 2048)] * 2 + [(2048, 5632)]) * 22
    per_rank = [sum(a * b for a, b in shard_items(shapes, lambda s: s[0] * s[1], rank=r, world_size=8)) for r in range(8)]
    assert sum(per_rank) == 968884224 and max(per_rank) - min(per_rank) <= 256 * 2048
    covered = [shard_rows(8192, rank=r, world_size=8, multiple=64) for r in range(8)]
    assert covered[0][0] == 0 and covered[-1][1] == 8192 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    odd = [shard_rows(1000, rank=r, world_size=3) for r in range(3)]
    assert [b - a for a, b in odd] == [334, 333, 333]


def test_permutation_tables_host(cta):
    perm, sp, sps = cta.utils.get_permutations_24(4)
    assert perm.numel() == 1024 and sorted(perm.tolist()) == list(range(1024))
    assert sp[:8] == [0, 4, 1, 5, 2, 6, 3, 7] and sps == list(range(64))
    with pytest.raises(ValueError):
        cta.utils.get_permutations_24(3)


_DIST_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from compressed_tensors_amd.distributed import init_dist, rank_and_world, shard_modules, shard_rows, module_size
init_dist()
rank, world = rank_and_world()
assert dist.get_backend() == "gloo" and world == 2
mods = [torch.nn.Linear(8 * (i + 1), 16, bias=False) for i in range(7)]
mine = shard_modules(mods)
ids = torch.zeros(7, dtype=torch.int64)
for m in mine:
    ids[mods.index(m)] = 1
# test-only all_reduce to verify the shards partition the work (the data path itself has no collective)
dist.all_reduce(ids)
assert ids.tolist() == [1] * 7, ids
sizes = torch.tensor([sum(module_size(m) for m in mine)], dtype=torch.int64)
gathered = [torch.zeros_like(sizes) for _ in range(world)]
dist.all_gather(gathered, sizes)
total = sum(int(g) for g in gathered)
assert total == sum(module_size(m) for m in mods)
a, b = shard_rows(1000)
assert (a, b) == ((0, 500) if rank == 0 else (500, 1000))
dist.barrier()
# one marker file per rank: two processes printing to the same pipe can interleave their characters
open(os.path.join(os.environ["CT_TEST_OUT"], f"rank{{rank}}.ok"), "w").write("ok")
"""


def test_world_size_2_sharding_gloo(tmp_path):
    script = tmp_path / "dist_check.py"
    script.write_text(_DIST_SCRIPT.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", CT_TEST_OUT=str(tmp_path))
    import socket
    with socket.socket() as sk:  # a free rendezvous port (a fixed one collides with a run in TIME_WAIT)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)],
        capture_output=True, text=True, env=env, timeout=240,
    )
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), r.stdout + r.stderr


_ROWSHARD_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
import oracle as O   # the CPU checker stands in for the kernels on this GPU-less host: what is tested is the sharding logic
from compressed_tensors_amd.distributed import init_dist, rank_and_world, shard_rows, merge_bitmask_row_shards
init_dist()
rank, world = rank_and_world()
torch.manual_seed(0)  # the same tensor on every rank
R, C = 320, 256
w = torch.randn(R, C, dtype=torch.bfloat16)
a, b = shard_rows(R, multiple=64)
assert (a, b) == ((0, 192) if rank == 0 else (192, 320))  # multiples of 64 rows, remainder on the last rank
# config 2: rows are independent -> the shard's packed words / dequantized rows are a slice of the single-rank result
scale, zp = O.calculate_qparams_minmax(w, num_bits=4, group_size=128, symmetric=True)
kw = dict(num_bits=4, strategy="group", group_size=128, symmetric=True)
full = O.pack_quantized_compress({{"weight": w, "weight_scale": scale, "weight_zero_point": zp}}, **kw)
mine = O.pack_quantized_compress({{"weight": w[a:b], "weight_scale": scale[a:b], "weight_zero_point": zp[a:b]}}, **kw)
gathered = [None] * world
dist.all_gather_object(gathered, mine["weight_packed"])   # test-only collective
assert torch.equal(torch.cat(gathered), full["weight_packed"])
# config 3: values / bitmask concatenate, row_offsets are rebased by the earlier shards' nnz
x = w.masked_fill(torch.rand(R, C) < 0.5, 0)
v, bm, ro = O.bitmask_compress(x[a:b])
assert int(ro[0]) == 0
shards = [None] * world
dist.all_gather_object(shards, (v, bm, ro))
mv, mb, mo = merge_bitmask_row_shards(shards)
fv, fb, fo = O.bitmask_compress(x)
assert torch.equal(mv.view(torch.int16), fv.view(torch.int16)) and torch.equal(mb, fb) and torch.equal(mo, fo) and mo.dtype == torch.int64
assert torch.equal(O.bitmask_decompress(mv, mb, x.shape).view(torch.int16), x.view(torch.int16))
dist.barrier()
open(os.path.join(os.environ["CT_TEST_OUT"], f"rank{{rank}}.ok"), "w").write("ok")
"""


def test_world_size_2_row_block_shards_gloo(tmp_path):
    """SURVEY 8e, single-tensor configs: row-block shards of ONE tensor reassemble to the single-rank result bit for bit
    (W4A16 packed words; sparse-bitmask values / bitmask / rebased row_offsets)"""
    import socket

    script = tmp_path / "rowshard_check.py"
    script.write_text(_ROWSHARD_SCRIPT.format(root=ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", CT_TEST_OUT=str(tmp_path))
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)],
        capture_output=True, text=True, env=env, timeout=240,
    )
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), r.stdout + r.stderr


_RECOUPLE_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from compressed_tensors_amd.distributed import init_dist, rank_and_world, replace_module_parallel
from compressed_tensors_amd.quantization.quant_args import QuantizationStatus
from compressed_tensors_amd.utils.module import get_direct_state_dict, replace_direct_state_dict
init_dist()
rank, world = rank_and_world()
torch.manual_seed(0)  # identical models on every rank
mods = [torch.nn.Linear(8 * (i + 1), 16 + i, bias=(i % 2 == 0)) for i in range(7)]
ref = [m.weight.data.clone() for m in mods]

def fake_compress(ms):  # what a codec does to a module: new keys, dropped keys, a small host tensor, a status
    for m in ms:
        sd = get_direct_state_dict(m)
        w = sd.pop("weight")
        sd.pop("bias", None)
        sd["weight_packed"] = (w * 2 + rank * 0).to(torch.float16)   # owner-independent content
        sd["weight_shape"] = torch.tensor(w.shape)
        sd["owner"] = torch.full((3,), float(rank))
        replace_direct_state_dict(m, sd)
        m.quantization_status = QuantizationStatus.COMPRESSED

mine = replace_module_parallel(list(mods), fake_compress, recouple=True)
assert 0 < len(mine) < len(mods)
owners = set()
for m, w in zip(mods, ref):
    sd = get_direct_state_dict(m)
    assert sorted(sd) == ["owner", "weight_packed", "weight_shape"], sorted(sd)
    assert torch.equal(sd["weight_packed"], (w * 2).to(torch.float16)) and sd["weight_packed"].dtype == torch.float16
    assert sd["weight_shape"].tolist() == list(w.shape) and sd["weight_shape"].dtype == torch.int64
    assert m.quantization_status == QuantizationStatus.COMPRESSED
    assert all(isinstance(p, torch.nn.Parameter) and not p.requires_grad for p in m._parameters.values())
    owners.add(int(sd["owner"][0]))
assert owners == {{0, 1}}  # both ranks contributed and both see the other's work
# without recouple only the own share changes
mods2 = [torch.nn.Linear(8, 8) for _ in range(4)]
mine2 = replace_module_parallel(list(mods2), fake_compress, recouple=False)
assert sum(hasattr(m, "weight_packed") for m in mods2) == len(mine2) == 2
dist.barrier()
open(os.path.join(os.environ["CT_TEST_OUT"], f"rank{{rank}}.ok"), "w").write("ok")
"""


def test_world_size_2_recouple_gloo(tmp_path):
    """N3: module-parallel apply + recouple (one flat-buffer broadcast per owner rank) over two gloo ranks"""
    import socket

    script = tmp_path / "recouple_check.py"
    script.write_text(_RECOUPLE_SCRIPT.format(root=ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", CT_TEST_OUT=str(tmp_path))
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)],
        capture_output=True, text=True, env=env, timeout=240,
    )
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), r.stdout + r.stderr


_MODEL_DIST_SCRIPT = r"""
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import compressed_tensors_amd as cta
from compressed_tensors_amd.distributed import init_dist, rank_and_world
from compressed_tensors_amd.quantization.quant_args import QuantizationStatus as St
from compressed_tensors_amd.compressors.model_compressors import model_compressor as mc
from compressed_tensors_amd.utils.module import get_direct_state_dict, replace_direct_state_dict
init_dist()
rank, world = rank_and_world()

# stand-ins for the codec (no GPU on this host): the state-dict transformation a codec performs
def fake_compress(ms, fmt=None):
    for m in ms:
        sd = get_direct_state_dict(m)
        w = sd.pop("weight")
        sd["weight_packed"] = w * 4   # exact
        sd["weight_shape"] = torch.tensor(w.shape)
        replace_direct_state_dict(m, sd)
        m.quantization_status = St.COMPRESSED

def fake_decompress(ms, fmt=None):
    for m in ms:
        sd = get_direct_state_dict(m)   # KeyError on a module that is not compressed, like the real codecs
        w = sd.pop("weight_packed") / 4
        sd.pop("weight_shape")
        sd["weight"] = w
        replace_direct_state_dict(m, sd)
        m.quantization_status = St.DECOMPRESSED

mc.compress_modules, mc.decompress_modules = fake_compress, fake_decompress

def model():
    torch.manual_seed(0)
    net = torch.nn.Sequential(*[torch.nn.Linear(16 * (i + 1), 8, bias=False) for i in range(6)])
    for m in net:
        m.quantization_scheme = cta.QuantizationScheme(targets=["Linear"], weights=cta.QuantizationArgs(num_bits=4, group_size=8))
    return net

cfg = types.SimpleNamespace(quantization_status=St.FROZEN)
comp = cta.ModelCompressor(quantization_config=cfg)

# 1. default compress = the reference's semantics: every rank ends with the whole model compressed
net = model(); ref = [m.weight.data.clone() for m in net]
mine = comp.compress_model(net)
assert 0 < len(mine) < 6
assert all(hasattr(m, "weight_packed") and not hasattr(m, "weight") and m.quantization_status == St.COMPRESSED for m in net)
assert cfg.quantization_status == St.COMPRESSED and hasattr(net, "ct_decompress_hook")
# 2. default decompress: every rank restores every module locally (this is what the forward pre-hook calls)
comp.decompress_model(net)
assert all(torch.equal(m.weight.data, w) for m, w in zip(net, ref)) and not hasattr(net, "ct_decompress_hook")
assert cfg.quantization_status == St.DECOMPRESSED

# 3. shard-per-rank mode: a partial result, so no model-wide status and no hook; decompress still restores a full model
cfg.quantization_status = St.FROZEN
net = model()
mine = comp.compress_model(net, recouple=False)
assert sum(hasattr(m, "weight_packed") for m in net) == len(mine) and 0 < len(mine) < 6
assert cfg.quantization_status == St.FROZEN and not hasattr(net, "ct_decompress_hook")
flags = torch.tensor([int(hasattr(m, "weight_packed")) for m in net]); dist.all_reduce(flags)
assert flags.tolist() == [1] * 6   # the shards partition the model
comp.decompress_model(net)
assert all(torch.equal(m.weight.data, w) for m, w in zip(net, ref))

# 4. distributed decompression (opt-in): each rank decompresses its share, one broadcast per owner replicates it
net = model(); comp.compress_model(net)
comp.decompress_model(net, recouple=True)
assert all(torch.equal(m.weight.data, w) and m.quantization_status == St.DECOMPRESSED for m, w in zip(net, ref))
# ... and ownership is agreed even when the replicas' byte sizes have drifted apart
# (ADVICE r02: each rank skips only what IT compressed, so the ranks' filtered lists differ — identity is agreed by module
# name, a module some rank already holds is owned by that rank, and every module must end with ITS OWN data and shape)
net = model(); first = comp.compress_model(net, recouple=False)
second = comp.compress_model(net, skip_compressed=True)
assert second == []   # every module was already held compressed by one of the two ranks: nothing is recomputed
for m, w in zip(net, ref):
    assert m.quantization_status == St.COMPRESSED and not hasattr(m, "weight")
    assert m.weight_packed.shape == w.shape and torch.equal(m.weight_packed.data, w * 4), "a module received another module's state"
    assert m.weight_shape.tolist() == list(w.shape)
comp.decompress_model(net)
assert all(torch.equal(m.weight.data, w) for m, w in zip(net, ref))
# the collective-free mode with a skip filter: bins from the unfiltered list with a compression-invariant weight -> still a partition
net = model(); a = comp.compress_model(net, recouple=False)
b = comp.compress_model(net, recouple=False, skip_compressed=True)
assert b == [] and sum(hasattr(m, "weight_packed") for m in net) == len(a)
# ranks that disagree about the module list get an error, not a silent mix-up
from compressed_tensors_amd.distributed import replace_module_parallel
mods = [torch.nn.Linear(4, 4) for _ in range(3)]
try:
    replace_module_parallel(mods, lambda ms: None, recouple=True, names=[f"m{{i}}" for i in range(3)] if rank == 0 else ["m0", "mX", "m2"])
    raise SystemExit("differing module lists were accepted")
except RuntimeError as e:
    assert "different modules" in str(e)
dist.barrier()
open(os.path.join(os.environ["CT_TEST_OUT"], f"rank{{rank}}.ok"), "w").write("ok")
"""


def test_world_size_2_model_compressor_semantics_gloo(tmp_path):
    """compress_model replicates by default (reference model_compressor.py:167-172 -> replace_module_parallel), decompress_model
    decompresses everything on every rank (:196), the collective-free shard mode leaves the model-wide status alone"""
    import socket

    script = tmp_path / "model_dist_check.py"
    script.write_text(_MODEL_DIST_SCRIPT.format(root=ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", CT_TEST_OUT=str(tmp_path))
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)],
        capture_output=True, text=True, env=env, timeout=240,
    )
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), r.stdout + r.stderr


def test_float_scheme_format_inference_and_param_names(cta):
    """the FLOAT formats resolve in upstream's priority order (compressors/format.py:18-27) and declare upstream's
    parameter names (nvfp4/base.py:36-47, mxfp4/base.py:34-44, naive_quantized/base.py:27-46)"""
    from compressed_tensors_amd.compressors import infer_module_format
    from compressed_tensors_amd.entrypoints.convert.converters import _args_from_dict
    from compressed_tensors_amd.quantization.utils import _float_kind

    QA, QS, F = cta.QuantizationArgs, cta.QuantizationScheme, cta.CompressionFormat
    f8 = torch.float8_e4m3fn
    act = QA(num_bits=8, type="float", strategy="tensor")
    nv = QA(num_bits=4, type="float", strategy="tensor_group", group_size=16, scale_dtype=f8, zp_dtype=f8)
    mx4 = QA(num_bits=4, type="float", strategy="group", group_size=32, scale_dtype=torch.uint8, zp_dtype=torch.uint8)
    mx8 = QA(num_bits=8, type="float", strategy="group", group_size=32, scale_dtype=torch.uint8, zp_dtype=torch.uint8)
    fp8 = QA(num_bits=8, type="float", strategy="channel")
    lin = torch.nn.Linear
    assert infer_module_format(lin, QS(weights=nv)) == F.nvfp4_pack_quantized
    assert infer_module_format(lin, QS(weights=mx4)) == F.mxfp4_pack_quantized
    assert infer_module_format(lin, QS(weights=mx8)) == F.mxfp8_quantized
    assert infer_module_format(lin, QS(weights=fp8, input_activations=act)) == F.float_quantized
    assert infer_module_format(lin, QS(weights=fp8)) == F.naive_quantized  # weight-only FP8 falls to the generic codec, as upstream
    assert infer_module_format(lin, QS(weights=QA(num_bits=8, type="float", strategy="group", group_size=32))) == F.naive_quantized  # float scales: not MX
    assert [_float_kind(a) for a in (nv, mx4, mx8, fp8)] == ["nvfp4", "mxfp4", "mxfp8", "fp8"]
    assert cta.NVFP4PackedCompressor.compression_param_names(QS(weights=nv)) == ("weight_packed", "weight_scale", "weight_global_scale")
    static_in = QA(num_bits=4, type="float", strategy="tensor_group", group_size=16, dynamic=False)
    assert cta.NVFP4PackedCompressor.compression_param_names(QS(weights=nv, input_activations=static_in))[-1] == "input_global_scale"
    assert cta.MXFP4PackedCompressor.compression_param_names(QS(weights=mx4)) == ("weight_packed", "weight_scale")
    assert cta.MXFP8QuantizationCompressor.compression_param_names(QS(weights=mx8)) == ("weight", "weight_scale")
    assert not cta.MXFP8QuantizationCompressor.can_compress(lin, QS(weights=fp8)) and not cta.MXFP4PackedCompressor.can_compress(lin, QS(weights=nv))
    # config.json round trip of the dtype fields (serialised as str(dtype), quant_args.py:209-224)
    parsed = _args_from_dict({"num_bits": 4, "type": "float", "strategy": "group", "group_size": 32, "scale_dtype": "torch.uint8", "zp_dtype": "torch.uint8",
                              "symmetric": True, "dynamic": False})
    assert parsed.scale_dtype is torch.uint8 and parsed.zp_dtype is torch.uint8 and _float_kind(parsed) == "mxfp4"
    assert cta.quantization.calculate_range(fp8) == (-448.0, 448.0) and cta.quantization.calculate_range(nv) == (-6.0, 6.0)
    assert fp8.pytorch_dtype() is f8
    with pytest.raises(NotImplementedError):
        nv.pytorch_dtype()


def test_launches_run_with_the_tensors_device_current(monkeypatch):
    """_lib.call: the trailing stream argument remembers its device, and the launch happens with THAT device current (the null
    stream resolves against the current device).  No GPU here: torch.cuda's device bookkeeping is stubbed."""
    from compressed_tensors_amd import _lib

    state = {"current": 0, "entered": []}

    class FakeDeviceCtx:
        def __init__(self, idx):
            self.idx = idx

        def __enter__(self):
            state["entered"].append(self.idx)
            self.prev, state["current"] = state["current"], self.idx

        def __exit__(self, *a):
            state["current"] = self.prev

    seen = []

    class FakeLib:
        @staticmethod
        def ct_probe(a, stream):
            seen.append((a, int(stream), state["current"]))
            return 0

    monkeypatch.setattr(_lib, "load", lambda: FakeLib)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["current"])
    monkeypatch.setattr(torch.cuda, "device", FakeDeviceCtx)
    h1 = _lib.StreamHandle(0)
    h1.device_index = 1
    _lib.call("ct_probe", 7, h1)          # tensor on cuda:1 while cuda:0 is current -> switched for the call, restored after
    h0 = _lib.StreamHandle(1234)
    h0.device_index = 0
    _lib.call("ct_probe", 8, h0)          # already current: no switch
    _lib.call("ct_probe", 9, 5678)        # a raw handle from a C-style caller: left alone
    assert seen == [(7, 0, 1), (8, 1234, 0), (9, 5678, 0)] and state["entered"] == [1] and state["current"] == 0


def test_bench_reads_valu_utilisation_from_the_committed_profile():
    """bench.py reports VALU utilisation next to the GB/s of the two headline kernels from profiles/r02_headline_sq.txt:
    the parser must find both kernels and give a fraction in (0, 1); the algorithmic byte count it divides by elsewhere is the survey's"""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    busy = bench.valu_busy_from_profile()
    assert set(busy) == {"w4_quant_pack_lean_kernel<bf16>", "w4_unpack_dequant_kernel<bf16>"}
    for v in busy.values():
        assert 0.0 < v["valu_busy_frac"] < 1.0 and v["valu_insts_per_launch"] > 0
    assert bench.alg_bytes_one_direction() == 168820736


def test_batch_table_words_match_the_struct(cta):
    """the batched entries take a table of `struct ct_w4_item` (include/ct_hip.h); codec.W4Batch fills it as a flat array of 64-bit
    words — the word layout must be the struct's (a miscount shifts every field of every item but the first), and the library's
    host-side planner must accept it and fill the derived fields where the struct has them"""
    import array
    import ctypes

    from compressed_tensors_amd import _lib, codec

    assert ctypes.sizeof(_lib.W4Item) == 8 * codec._ITEM_WORDS
    assert ctypes.sizeof(_lib.BitmaskDItem) == 8 * 9 and _lib.BitmaskDItem.dt.offset == 56 and _lib.BitmaskDItem.first_block.offset == 64
    assert ctypes.sizeof(_lib.BitmaskItem) == 8 * 15 and ctypes.sizeof(_lib.CopyItem) == 8 * 4  # csrc/host/ct_hostpath.cpp fills these as 15 / 4 words
    assert _lib.BitmaskItem.dt.offset == 64 and _lib.BitmaskItem.first_block.offset == 72 and _lib.BitmaskItem.nwg.offset == 104 and _lib.BitmaskItem.gen.offset == 116
    assert codec._ITEM_WORDS == 13  # round 6: + zp_packed, main_blocks, {g_magic, g_shift}
    shapes = [(2048, 2048, 128), (256, 2048, 128), (5632, 2048, 128), (2048, 5632, 5632), (1001, 5632, 128), (64, 96, 32)]
    flat, structs = [], (_lib.W4Item * len(shapes))()
    for i, (r, c, g) in enumerate(shapes):
        ptrs = [0x10000 * (5 * i + k + 1) for k in range(4)]
        zpp = 0x10000 * (5 * i + 5) if g == 128 and c % 512 == 0 and i % 2 == 0 else 0  # some items carry the stored form of their zero points
        flat += (*ptrs, r, c, g, 0, 0, 0, zpp, 0, 0)
        it = structs[i]
        it.src, it.scale, it.zp, it.dst = ptrs
        it.rows, it.cols, it.group = r, c, g
        it.zp_packed = zpp or None
    assert len(flat) == codec._ITEM_WORDS * len(shapes)
    words = array.array("q", flat)
    lib = _lib.load()
    for direction in (0, 1):
        w = array.array("q", words)
        s = (_lib.W4Item * len(shapes)).from_buffer_copy(bytes(structs))
        blocks_w = lib.ct_w4_batch_plan(w.buffer_info()[0], len(shapes), direction)
        blocks_s = lib.ct_w4_batch_plan(ctypes.cast(s, ctypes.c_void_p), len(shapes), direction)
        assert blocks_w == blocks_s > 0
        assert bytes(w) == bytes(s)
        first = 0
        for i, (r, c, g) in enumerate(shapes):
            assert (s[i].rows, s[i].cols, s[i].group) == (r, c, g) and s[i].units == r * c // 8 and s[i].first_block == first
            main = -(-(r * c // 32) // 256) if direction == 0 else -(-(r * c // 8) // 1024)
            G = c // g
            tail = -(-(-(-r // 8) * G) // 256) if s[i].zp_packed else 0  # one lane per stored word (ceil(rows / 8), G)
            assert s[i].main_blocks == main
            first += main + tail
            # n // G == (n * g_magic) >> g_shift for every n the kernels divide (word / group indices below 2^31)
            for nn in (0, 1, G - 1, G, G + 1, 12345 * G + G - 1, 2 ** 31 - 1, 2 ** 31 - G, 7 * G * 1000 + 3):
                assert (nn * s[i].g_magic) >> s[i].g_shift == nn // G, (G, nn)
        assert blocks_s == first
    # the stored form can only be READ by the decompress kernels for groups of 128 and cols % 512 == 0; the plan says so instead of mis-reading
    bad = (_lib.W4Item * 1)()
    bad[0].src, bad[0].scale, bad[0].zp, bad[0].dst, bad[0].zp_packed = 0x10000, 0x20000, 0x30000, 0x40000, 0x50000
    bad[0].rows, bad[0].cols, bad[0].group = 64, 256, 64
    assert lib.ct_w4_batch_plan(ctypes.cast(bad, ctypes.c_void_p), 1, 1) == -1 and "packed form" in _lib.last_error()
    assert lib.ct_w4_batch_plan(ctypes.cast(bad, ctypes.c_void_p), 1, 0) > 0  # compress: any batch item may ask for it
    bad[0].zp = None
    assert lib.ct_w4_batch_plan(ctypes.cast(bad, ctypes.c_void_p), 1, 0) == -1 and "without giving" in _lib.last_error()


def _tree(cta, scheme, shapes, *, trainable_scale=False, buffer_zp=False, odd_class=False, g_idx=False):
    class Odd(torch.nn.Linear):
        def __setattr__(self, name, value):
            super().__setattr__(name, value)

    root = torch.nn.Module()
    root.blocks = torch.nn.ModuleList()
    for k, (r, c) in enumerate(shapes):
        lin = (Odd if odd_class and k == 1 else torch.nn.Linear)(c, r, bias=False, device="meta")
        lin.weight = torch.nn.Parameter(torch.zeros(r, c, dtype=torch.bfloat16), requires_grad=True)  # a fresh Linear's weight is trainable
        lin.weight_scale = torch.nn.Parameter(torch.ones(r, c // 128, dtype=torch.bfloat16), requires_grad=trainable_scale and k == 2)
        zp = torch.zeros(r, c // 128, dtype=torch.int8)
        if buffer_zp and k == 0:
            lin.register_buffer("weight_zero_point", zp)
        else:
            lin.weight_zero_point = torch.nn.Parameter(zp, requires_grad=False)
        if g_idx and k == 3:
            lin.weight_g_idx = torch.nn.Parameter(torch.arange(c, dtype=torch.int32) // 128, requires_grad=False)
        lin.quantization_scheme = scheme
        blk = torch.nn.Module()
        blk.proj = lin
        root.blocks.append(blk)
    root.shared = root.blocks[0]  # the same module twice in the tree: walked once
    return root


@pytest.mark.parametrize("variant", ["plain", "trainable_scale", "buffer_zp", "odd_class", "g_idx", "asymmetric", "asymmetric_buffer_zp"])
def test_cpp_host_loop_matches_the_python_loop(cta, monkeypatch, variant):
    """csrc/host/ct_hostpath.cpp (the per-module loop of PackedQuantizationCompressor.compress_modules / decompress_modules in C++) against
    the Python loop it replaces, on CPU tensors with the two launches stubbed out: the same table (pointers, shapes, groups), the
    same modules handed back for the generic path, and every module left in the same state — names, order, kinds, trainability,
    shapes, dtypes, status; also the module walk against named_modules(remove_duplicate=True) + is_module_quantized"""
    import copy

    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd import codec
    from compressed_tensors_amd.compressors.pack_quantized import base as pq
    from compressed_tensors_amd.quantization.quant_args import QuantizationStatus
    from compressed_tensors_amd.quantization.utils import is_module_quantized

    hp = pq._hostpath()
    assert hp is not None, "the host extension was not built (python -c 'import __graft_entry__ as g; g.build()')"
    asym = variant.startswith("asymmetric")
    args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=not asym, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    shapes = [(64, 256), (32, 512), (96, 128), (64, 384), (20, 256)]  # (20 rows: the packed zero points round up to 3 rows)
    flags = {"plain": {}, "asymmetric": {}, "asymmetric_buffer_zp": {"buffer_zp": True}}.get(variant, {variant: True})
    a = _tree(cta, scheme, shapes, **flags)
    b = copy.deepcopy(a)
    for x, y in zip(a.modules(), b.modules()):
        if hasattr(x, "quantization_scheme"):
            y.quantization_scheme = x.quantization_scheme  # one scheme object per group, as apply_quantization_config attaches it
    assert [id(m) for m in hp.quantized_modules(a)] == [id(m) for _, m in a.named_modules(remove_duplicate=True) if is_module_quantized(m)]

    tables = {"cpp": [], "py": []}
    which = {"now": "cpp"}

    def fake_words(words, n, direction, dtype, device):
        w = words.reshape(n, codec._ITEM_WORDS)
        rows = [r[4:7] + [bool(r[10])] for r in w.tolist()]  # rows, cols, group, carries the stored form of its zero points (ct_w4_item.zp_packed)
        tables[which["now"]].append((direction, dtype, rows))

    class FakeBatch:
        def __init__(self, entries, direction, dtype, kind="w4", bits=8):
            self.rec = (direction, dtype, [[int(e[4]), int(e[5]), int(e[6]), len(e) > 7 and e[7] is not None] for e in entries])

        def launch(self, stream=None):
            if self.rec[2]:
                tables[which["now"]].append(self.rec)

    def fake_zp_words(words, n, direction, device):
        if n:
            tables[which["now"]].append(("zp-" + direction, None, words.reshape(n, codec._ITEM_WORDS)[:, 4:6].tolist()))

    def fake_zp_batch(pairs, direction):
        pairs = list(pairs)
        if pairs:
            tables[which["now"]].append(("zp-" + direction, None, [list((src if direction == "pack" else dst).shape) for src, dst in pairs]))

    monkeypatch.setattr(codec, "launch_w4_words", fake_words)
    monkeypatch.setattr(codec, "launch_zp4_words", fake_zp_words)
    monkeypatch.setattr(codec, "zp4_batch", fake_zp_batch)
    monkeypatch.setattr(codec, "pack_to_int32", lambda zp, bits, packed_dim=1: torch.zeros((zp.shape[0] * 4 + 31) // 32, zp.shape[1], dtype=torch.int32))
    monkeypatch.setattr(codec, "unpack_from_int32", lambda p, bits, shape, packed_dim=1: torch.zeros(tuple(shape), dtype=torch.int8))
    monkeypatch.setattr(codec, "W4Batch", FakeBatch)
    monkeypatch.setattr(codec, "quantize_and_pack_with_zp", lambda *a_, **k: None)  # the single-module one-launch forms decline: the stubs compose
    monkeypatch.setattr(codec, "unpack_and_dequantize_with_zp", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "quantize_and_pack", lambda w, *a_, **k: torch.zeros(w.shape[0], w.shape[1] // 8, dtype=torch.int32))
    monkeypatch.setattr(codec, "unpack_and_dequantize", lambda p, shape, scale, *a_, **k: torch.zeros(shape, dtype=scale.dtype))
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)  # the Python loop's own device test
    hp.set_allow_cpu(True)
    try:
        mods_a = [m for m in a.modules() if isinstance(m, torch.nn.Linear)]
        mods_b = [m for m in b.modules() if isinstance(m, torch.nn.Linear)]
        pq.PackedQuantizationCompressor.compress_modules(mods_a)  # C++ loop first, Python loop for what it hands back
        which["now"] = "py"
        monkeypatch.setattr(ctlib, "_HOSTPATH", [None])  # the Python loop alone
        pq.PackedQuantizationCompressor.compress_modules(mods_b)
        monkeypatch.setattr(ctlib, "_HOSTPATH", [hp])
        n_plain = {"plain": 5, "trainable_scale": 4, "buffer_zp": 4, "odd_class": 4, "g_idx": 4, "asymmetric": 5, "asymmetric_buffer_zp": 4}[variant]
        if asym:  # round 6: every zero point packed by the WEIGHTS' launch (its items carry zp_packed; no zp-pack launch is left), and stored packed
            assert not [t for t in tables["cpp"] + tables["py"] if t[0] == "zp-pack"]
            assert sum(r[3] for t in tables["cpp"] if t[0] == "compress" for r in t[2]) == len(mods_a)
            assert all(x.weight_zero_point.dtype == torch.int32 and x.weight_zero_point.shape == ((x.out_features * 4 + 31) // 32, x.in_features // 128) for x in mods_a)
        assert sum(len(t[2]) for t in tables["cpp"] if t[0] == "compress") >= n_plain - 0
        for x, y in zip(mods_a, mods_b):
            assert _module_state_no_ptr(x) == _module_state_no_ptr(y), variant
            assert x.quantization_status == QuantizationStatus.COMPRESSED == y.quantization_status
            assert x.weight_shape.tolist() == list(x.weight_packed.shape[:1]) + [x.weight_packed.shape[1] * 8]
        which["now"] = "cpp"
        pq.PackedQuantizationCompressor.decompress_modules(mods_a)
        which["now"] = "py"
        monkeypatch.setattr(ctlib, "_HOSTPATH", [None])
        pq.PackedQuantizationCompressor.decompress_modules(mods_b)
        for x, y in zip(mods_a, mods_b):
            assert _module_state_no_ptr(x) == _module_state_no_ptr(y), variant
            assert x.quantization_status == QuantizationStatus.DECOMPRESSED == y.quantization_status
            assert x.weight.shape == (x.out_features, x.in_features) and x.weight.dtype == torch.bfloat16
            if asym:
                assert x.weight_zero_point.dtype == torch.int8 and x.weight_zero_point.shape == (x.out_features, x.in_features // 128)
        # the same work reached the launches, whichever loop built the table
        flat = lambda ts, d: sorted(tuple(r) for t in ts if t[0] == d for r in t[2])
        if asym:  # zero points the weights' launch cannot read in stored form (here: cols % 512 != 0) are unpacked BEFORE that launch; the others ride in it
            order = [t[0] for t in tables["cpp"] if t[0] in ("zp-unpack", "decompress")]
            assert order[:2] == ["zp-unpack", "decompress"], order
            for t in tables["cpp"] + tables["py"]:
                if t[0] == "decompress":
                    assert all(r[3] == (r[1] % 512 == 0 and r[2] == 128) for r in t[2]), t
            assert sum(len(t[2]) for t in tables["cpp"] if t[0] == "zp-unpack") == sum(1 for x in mods_a if x.in_features % 512)
        for d in ("compress", "decompress", "zp-pack", "zp-unpack"):
            assert flat(tables["cpp"], d) == flat(tables["py"], d), (variant, d)
    finally:
        hp.set_allow_cpu(False)


@pytest.mark.parametrize("variant", ["fp8", "fp8_zero_point", "fp8_static_input", "int8_channel", "int8_asymmetric_group", "int8_tensor", "int6", "trainable_scale",
                                     "buffer_zp", "odd_class", "g_idx", "block", "float32"])
def test_cpp_host_loop_of_the_8bit_codecs_matches_the_python_loop(cta, monkeypatch, variant):
    """csrc/host/ct_hostpath.cpp q8_plan_compress / q8_plan_decompress / q8_finish (the per-module loop of NaiveQuantizationCompressor.compress_modules /
    decompress_modules and its Int / Float subclasses in C++) against the Python loop, on CPU tensors with the launches stubbed out: the same table rows
    (shapes, elements per scale, zero point present), kinds and bit widths reach the launches, the same modules are handed back, and every module is left in
    the same state (names, order, kinds, trainability, shapes, dtypes, status) — incl. the zero points a symmetric scheme drops (weight and input)"""
    import copy

    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd import codec
    from compressed_tensors_amd.compressors.naive_quantized import base as nq
    from compressed_tensors_amd.quantization.quant_args import QuantizationStatus

    hp = ctlib.hostpath()
    assert hp is not None and hasattr(hp, "q8_plan_compress"), "the host extension was not built (python -c 'import __graft_entry__ as g; g.build()')"
    F8 = torch.float8_e4m3fn
    wdt = torch.float32 if variant == "float32" else torch.bfloat16
    ia = None
    if variant.startswith("fp8") or variant in ("trainable_scale", "buffer_zp", "odd_class", "g_idx", "float32"):
        wa = cta.QuantizationArgs(num_bits=8, type="float", strategy="channel", symmetric=True)
        if variant == "fp8_static_input":
            ia = cta.QuantizationArgs(num_bits=8, type="float", strategy="tensor", symmetric=True)
    elif variant == "block":
        wa = cta.QuantizationArgs(num_bits=8, type="float", strategy="block", block_structure=[16, 128], symmetric=True)
    elif variant == "int8_channel":
        wa = cta.QuantizationArgs(num_bits=8, type="int", strategy="channel", symmetric=True)
    elif variant == "int8_asymmetric_group":
        wa = cta.QuantizationArgs(num_bits=8, type="int", strategy="group", group_size=128, symmetric=False)
    elif variant == "int8_tensor":
        wa = cta.QuantizationArgs(num_bits=8, type="int", strategy="tensor", symmetric=True)
    else:
        wa = cta.QuantizationArgs(num_bits=6, type="int", strategy="channel", symmetric=True)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa, input_activations=ia)
    st = wa.strategy if isinstance(wa.strategy, str) else wa.strategy.value
    is_float = (wa.type if isinstance(wa.type, str) else wa.type.value) == "float"
    with_zp = variant != "fp8"  # the calibrated flow attaches a zero point (float8 for FLOAT schemes); "fp8": none at all

    class Odd(torch.nn.Linear):
        def __setattr__(self, name, value):
            super().__setattr__(name, value)

    def tree():
        root = torch.nn.Module()
        root.blocks = torch.nn.ModuleList()
        for k, (r, c) in enumerate([(64, 256), (32, 512), (96, 128), (64, 384), (16, 256)]):
            lin = (Odd if variant == "odd_class" and k == 1 else torch.nn.Linear)(c, r, bias=False, device="meta")
            lin.weight = torch.nn.Parameter(torch.zeros(r, c, dtype=wdt), requires_grad=True)
            sshape = {"tensor": (1,), "channel": (r, 1), "group": (r, c // 128), "block": (r // 16, c // 128)}[st]
            lin.weight_scale = torch.nn.Parameter(torch.ones(sshape, dtype=wdt), requires_grad=variant == "trainable_scale" and k == 2)
            if with_zp:
                zp = torch.zeros(sshape, dtype=F8 if is_float else torch.int8)
                if variant == "buffer_zp" and k == 0:
                    lin.register_buffer("weight_zero_point", zp)
                else:
                    lin.weight_zero_point = torch.nn.Parameter(zp, requires_grad=False)
            if ia is not None:
                lin.input_scale = torch.nn.Parameter(torch.ones(1, dtype=wdt), requires_grad=False)
                lin.input_zero_point = torch.nn.Parameter(torch.zeros(1, dtype=F8), requires_grad=False)
            if variant == "g_idx" and k == 3:
                lin.weight_g_idx = torch.nn.Parameter(torch.arange(c, dtype=torch.int32) // 128, requires_grad=False)
            lin.quantization_scheme = scheme
            root.blocks.append(lin)
        return root

    a, b = tree(), tree()
    tables = {"cpp": [], "py": []}
    which = {"now": "cpp"}
    KIND = {"int8": 0, "fp8": 1, "fp8z": 2}

    def fake_words(words, n, direction, dtype, device, kind, bits=8):
        rows = [r[4:7] + [bool(r[2])] for r in words.reshape(n, codec._ITEM_WORDS).tolist()]
        tables[which["now"]].append((direction, dtype, kind, bits if direction == "compress" else 8, rows))
        native[direction] += n

    class FakeBatch:
        def __init__(self, entries, direction, dtype, kind="w4", bits=8):
            self.rec = (direction, dtype, KIND[kind], int(bits) if direction == "compress" else 8, [[int(e[4]), int(e[5]), int(e[6]), e[2] is not None] for e in entries])

        def launch(self, stream=None):
            if self.rec[4]:
                tables[which["now"]].append(self.rec)

    singles = {"cpp": 0, "py": 0}
    native = {"compress": 0, "decompress": 0}  # modules whose table rows the C++ loop wrote

    def fake_quantize(w, scale, zp, **kw):
        singles[which["now"]] += 1
        return torch.zeros(w.shape, dtype=kw["dtype"])

    def fake_dequantize(q, scale, zp, **kw):
        singles[which["now"]] += 1
        return torch.zeros(q.shape, dtype=scale.dtype)

    monkeypatch.setattr(codec, "launch_q8_words", fake_words)
    monkeypatch.setattr(codec, "W4Batch", FakeBatch)
    monkeypatch.setattr(codec, "quantize_tensor", fake_quantize)
    monkeypatch.setattr(codec, "dequantize_tensor", fake_dequantize)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    hp.set_allow_cpu(True)
    try:
        mods_a, mods_b = list(a.blocks), list(b.blocks)
        klass = nq.FloatQuantizationCompressor if is_float else nq.IntQuantizationCompressor
        for direction, status in (("compress", QuantizationStatus.COMPRESSED), ("decompress", QuantizationStatus.DECOMPRESSED)):
            which["now"] = "cpp"
            getattr(klass, direction + "_modules")(mods_a)
            which["now"] = "py"
            monkeypatch.setattr(ctlib, "_HOSTPATH", [None])
            getattr(klass, direction + "_modules")(mods_b)
            monkeypatch.setattr(ctlib, "_HOSTPATH", [hp])
            for x, y in zip(mods_a, mods_b):
                assert _module_state_no_ptr(x) == _module_state_no_ptr(y), (variant, direction)
                assert x.quantization_status == status == y.quantization_status
                assert x.weight.dtype == ((F8 if is_float else torch.int8) if direction == "compress" else wdt)
                assert ("weight_zero_point" in x._parameters) == (with_zp and not wa.symmetric)
                assert "input_zero_point" not in x._parameters
            flat = lambda ts: sorted((t[0], t[1], t[2], t[3], tuple(r)) for t in ts if t[0] == direction for r in t[4])
            assert flat(tables["cpp"]) == flat(tables["py"]), (variant, direction)
            assert singles["cpp"] == singles["py"]
            # the C++ loop took the plain modules itself; a module with a trainable entry that stays, a buffer, a class with its own __setattr__, activation
            # ordering, a block layout or float32 weights went back to the Python loop
            expect_native = {"trainable_scale": 4 if direction == "compress" else 5, "buffer_zp": 4 if direction == "compress" else 5, "odd_class": 4, "g_idx": 4, "block": 5, "float32": 0,
                             "int8_asymmetric_group": 4 if direction == "compress" else 5}.get(variant, 5)  # (96 x 128 in groups of 128: a (96, 1) scale)
            assert native[direction] == expect_native, (variant, direction, native)
    finally:
        hp.set_allow_cpu(False)


@pytest.mark.parametrize("fmt", ["nvfp4", "mxfp4"])
@pytest.mark.parametrize("variant", ["plain", "zero_point", "static_input", "trainable_scale", "buffer_zp", "odd_class", "f32_scale", "odd_global_scale", "other_scale_dtype"])
def test_cpp_host_loop_of_the_fp4_codecs_matches_the_python_loop(cta, monkeypatch, fmt, variant):
    """csrc/host/ct_hostpath.cpp fp4_plan_compress / fp4_plan_decompress / fp4_finish against the Python loops of NVFP4PackedCompressor / MXFP4PackedCompressor on CPU
    tensors (the launch itself skipped / stubbed): the same modules taken, every module left in the same state — names, ORDER, kinds, trainability, shapes,
    dtypes, status — incl. the zero points a symmetric scheme drops"""
    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd import codec
    from compressed_tensors_amd.compressors.fp4 import base as fp4
    from compressed_tensors_amd.quantization.quant_args import QuantizationStatus

    hp = ctlib.hostpath()
    assert hp is not None and hasattr(hp, "fp4_plan_compress"), "the host extension was not built (python -c 'import __graft_entry__ as g; g.build()')"
    F8 = torch.float8_e4m3fn
    group = 16 if fmt == "nvfp4" else 32
    if fmt == "mxfp4" and variant in ("f32_scale", "odd_global_scale"):
        pytest.skip("NVFP4 only")
    want = F8 if group == 16 else torch.uint8
    sdt = (torch.uint8 if group == 16 else F8) if variant == "other_scale_dtype" else want
    wa = (cta.QuantizationArgs(num_bits=4, type="float", strategy="tensor_group", symmetric=True, group_size=16, scale_dtype=sdt) if group == 16
          else cta.QuantizationArgs(num_bits=4, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=sdt))
    ia = cta.QuantizationArgs(num_bits=4, type="float", strategy="tensor", symmetric=True) if variant == "static_input" else None
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa, input_activations=ia)
    klass = fp4.NVFP4PackedCompressor if group == 16 else fp4.MXFP4PackedCompressor

    class Odd(torch.nn.Linear):
        def __setattr__(self, name, value):
            super().__setattr__(name, value)

    def tree():
        mods = []
        for k, (r, c) in enumerate([(64, 256), (32, 512), (96, 128), (8, 64)]):
            lin = (Odd if variant == "odd_class" and k == 1 else torch.nn.Linear)(c, r, bias=True, device="meta")
            lin.bias = torch.nn.Parameter(torch.zeros(r, dtype=torch.bfloat16), requires_grad=False)
            lin.weight = torch.nn.Parameter(torch.zeros(r, c, dtype=torch.bfloat16), requires_grad=True)
            lin.weight_scale = torch.nn.Parameter(torch.ones(r, c // group, dtype=torch.float32 if variant == "f32_scale" else torch.bfloat16),
                                                  requires_grad=variant == "trainable_scale" and k == 2)
            if group == 16:
                lin.weight_global_scale = torch.nn.Parameter(torch.ones(1, dtype=torch.float64 if variant == "odd_global_scale" and k == 0 else torch.float32), requires_grad=False)
            if variant in ("zero_point", "buffer_zp"):
                zp = torch.zeros(r, c // group, dtype=F8)
                if variant == "buffer_zp" and k == 0:
                    lin.register_buffer("weight_zero_point", zp)
                else:
                    lin.weight_zero_point = torch.nn.Parameter(zp, requires_grad=False)
            if ia is not None:
                lin.input_global_scale = torch.nn.Parameter(torch.ones(1), requires_grad=False)
                lin.input_zero_point = torch.nn.Parameter(torch.zeros(1, dtype=F8), requires_grad=False)
            lin.quantization_scheme = scheme
            mods.append(lin)
        return mods

    calls = {"stored": 0, "plain": 0, "unpack": 0}

    def fake_stored(weight, scale, gs, *, group_size, scale_dtype):
        if scale_dtype is not want or (group_size == 32 and scale.dtype is torch.float32):
            return None
        calls["stored"] += 1
        return torch.zeros(weight.shape[0], weight.shape[1] // 2, dtype=torch.uint8), torch.zeros(scale.shape, dtype=want)

    def fake_pack(weight, scale, gs, *, group_size):
        calls["plain"] += 1
        return torch.zeros(weight.shape[0], weight.shape[1] // 2, dtype=torch.uint8)

    def fake_unpack(packed, scale, gs, *, group_size, scale_kind="plain", dtype=torch.bfloat16, return_scale=False):
        calls["unpack"] += 1
        w = torch.zeros(packed.shape[0], packed.shape[1] * 2, dtype=dtype)
        return (w, torch.zeros(scale.shape, dtype=torch.bfloat16)) if return_scale else w

    monkeypatch.setattr(codec, "fp4_quantize_and_pack_stored", fake_stored)
    monkeypatch.setattr(codec, "fp4_quantize_and_pack", fake_pack)
    monkeypatch.setattr(codec, "fp4_unpack_and_dequantize", fake_unpack)
    monkeypatch.setattr(codec, "compress_mx_scale", lambda scale, dtype: torch.zeros(scale.shape, dtype=dtype))
    monkeypatch.setattr(codec, "_mx_code_table", lambda dt, dev: torch.zeros(65536, dtype=torch.uint8))
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    launched = []

    def fake_words(words, n, direction, device, group_, x_dtype=None, scale_dtype=None):
        rows = words.reshape(n, codec._ITEM_WORDS)
        assert group_ == group and all(int(r[6]) == group and bool(r[2]) == (group == 16) and r[10] for r in rows.tolist())
        launched.append((direction, n, x_dtype, scale_dtype, sorted((int(r[4]), int(r[5])) for r in rows.tolist())))

    monkeypatch.setattr(codec, "launch_fp4_words", fake_words)
    hp.set_allow_cpu(True)
    try:
        a, b = tree(), tree()
        for direction, status in (("compress", QuantizationStatus.COMPRESSED), ("decompress", QuantizationStatus.DECOMPRESSED)):
            before = dict(calls)
            getattr(klass, direction + "_modules")(a)
            python_calls_with_cpp = sum(calls.values()) - sum(before.values())
            monkeypatch.setattr(ctlib, "_HOSTPATH", [None])
            getattr(klass, direction + "_modules")(b)
            monkeypatch.setattr(ctlib, "_HOSTPATH", [hp])
            for x, y in zip(a, b):
                assert _module_state_no_ptr(x) == _module_state_no_ptr(y), (fmt, variant, direction)
                assert x.quantization_status == status == y.quantization_status
                assert "input_zero_point" not in x._parameters and "weight_zero_point" not in x._parameters
            # modules the C++ loop left to the Python loop (each makes one codec call there)
            left = {"trainable_scale": 0, "buffer_zp": 1 if direction == "compress" else 0, "odd_class": 1,
                    "odd_global_scale": 1, "other_scale_dtype": 4}.get(variant, 0)
            if variant == "other_scale_dtype" and direction == "decompress":
                left = 4  # a stored scale of another dtype is not this format's byte layout
            assert python_calls_with_cpp == left, (fmt, variant, direction, python_calls_with_cpp)
            mine = [t for t in launched if t[0] == direction]
            assert sum(t[1] for t in mine) == 4 - left, (fmt, variant, direction, mine)
            if mine and direction == "compress":
                assert mine[0][2] is torch.bfloat16 and mine[0][3] is (torch.float32 if variant == "f32_scale" else torch.bfloat16)
    finally:
        hp.set_allow_cpu(False)


@pytest.mark.parametrize("variant", ["channel", "group", "tensor", "asymmetric", "trainable_scale", "odd_cols", "activations"])
def test_cpp_host_loop_of_8bit_pack_quantized_matches_the_python_loop(cta, monkeypatch, variant):
    """csrc/host/ct_hostpath.cpp w8_plan_compress / w8_plan_decompress (pack-quantized with num_bits = 8, symmetric, weights only — the W8A16 preset — on the
    8-bit tables' packed kind, finished by the W4 finish functions) against the Python loop on CPU tensors with the launches stubbed out: same modules taken,
    same table rows, every module left in the same state (names, order, kinds, shapes, dtypes, status)"""
    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd import codec
    from compressed_tensors_amd.compressors.pack_quantized import base as pq
    from compressed_tensors_amd.quantization.quant_args import QuantizationStatus

    hp = ctlib.hostpath()
    assert hp is not None and hasattr(hp, "w8_plan_compress"), "the host extension was not built (python -c 'import __graft_entry__ as g; g.build()')"
    st = variant if variant in ("channel", "group", "tensor") else "channel"
    wa = cta.QuantizationArgs(num_bits=8, type="int", strategy=st, group_size=128 if st == "group" else None, symmetric=variant != "asymmetric")
    ia = cta.QuantizationArgs(num_bits=8, type="int", strategy="tensor", symmetric=True, dynamic=True) if variant == "activations" else None
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa, input_activations=ia)

    def tree():
        mods = []
        for k, (r, c) in enumerate([(64, 256), (32, 512), (96, 128), (8, 264 if variant == "odd_cols" else 384)]):
            lin = torch.nn.Linear(c, r, bias=False, device="meta")
            lin.weight = torch.nn.Parameter(torch.zeros(r, c, dtype=torch.bfloat16), requires_grad=True)
            sshape = {"tensor": (1,), "channel": (r, 1), "group": (r, max(c // 128, 1))}[st]
            lin.weight_scale = torch.nn.Parameter(torch.ones(sshape, dtype=torch.bfloat16), requires_grad=variant == "trainable_scale" and k == 2)
            lin.weight_zero_point = torch.nn.Parameter(torch.zeros(sshape, dtype=torch.int8), requires_grad=False)
            lin.quantization_scheme = scheme
            mods.append(lin)
        return mods

    native = {"compress": [], "decompress": []}

    def fake_words(words, n, direction, dtype, device, kind, bits=8):
        assert kind == 3 and bits == 8 and dtype is torch.bfloat16
        native[direction] += [tuple(r[4:7]) for r in words.reshape(n, codec._ITEM_WORDS).tolist()]

    monkeypatch.setattr(codec, "launch_q8_words", fake_words)
    monkeypatch.setattr(codec, "launch_w4_words", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "launch_zp4_words", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "quantize_and_pack_with_zp", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "unpack_and_dequantize_with_zp", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "pack_to_int32", lambda zp, bits, packed_dim=1: torch.zeros((zp.shape[0] * bits + 31) // 32, zp.shape[1], dtype=torch.int32))
    monkeypatch.setattr(codec, "unpack_from_int32", lambda p_, bits, shape, packed_dim=1: torch.zeros(tuple(shape), dtype=torch.int8))
    monkeypatch.setattr(codec, "quantize_and_pack", lambda w, *a_, **k: torch.zeros(w.shape[0], -(-w.shape[1] * 8 // 32), dtype=torch.int32))
    monkeypatch.setattr(codec, "unpack_and_dequantize", lambda p_, shape, scale, *a_, **k: torch.zeros(tuple(shape), dtype=scale.dtype))
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    hp.set_allow_cpu(True)
    try:
        a, b = tree(), tree()
        for direction, status in (("compress", QuantizationStatus.COMPRESSED), ("decompress", QuantizationStatus.DECOMPRESSED)):
            getattr(pq.PackedQuantizationCompressor, direction + "_modules")(a)
            monkeypatch.setattr(ctlib, "_HOSTPATH", [None])
            getattr(pq.PackedQuantizationCompressor, direction + "_modules")(b)
            monkeypatch.setattr(ctlib, "_HOSTPATH", [hp])
            for x, y in zip(a, b):
                assert _module_state_no_ptr(x) == _module_state_no_ptr(y), (variant, direction)
                assert x.quantization_status == status == y.quantization_status
            expect = {"asymmetric": 0, "activations": 0, "trainable_scale": 3 if direction == "compress" else 4, "odd_cols": 3,
                      "group": 3 if direction == "compress" else 4}.get(variant, 4)  # (96 x 128 in groups of 128: a (96, 1) scale — channel-wise by inference on the way back)
            assert len(native[direction]) == expect, (variant, direction, native[direction])
    finally:
        hp.set_allow_cpu(False)


@pytest.mark.parametrize("fmt", ["w8a16", "fp8", "fp8_block", "nvfp4", "mxfp4", "mxfp8"])
def test_cpp_host_loops_take_a_second_compress_after_a_decompress(cta, monkeypatch, fmt):
    """compress -> decompress -> compress -> decompress of one tree (what a benchmark loop, or a model that is decompressed for fine-tuning and compressed again, does):
    every module goes through the C++ loop in BOTH rounds — entries an earlier direction leaves behind (`weight_shape`, a bfloat16 scale) do not push the
    module back to the interpreter — and the second round leaves the same entries as the first"""
    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd import codec

    hp = ctlib.hostpath()
    assert hp is not None, "the host extension was not built"
    F8 = torch.float8_e4m3fn
    if fmt == "w8a16":
        wa, sshape, zdt = cta.QuantizationArgs(num_bits=8, type="int", strategy="channel", symmetric=True), lambda r, c: (r, 1), torch.int8
    elif fmt == "fp8":
        wa, sshape, zdt = cta.QuantizationArgs(num_bits=8, type="float", strategy="channel", symmetric=True), lambda r, c: (r, 1), F8
    elif fmt == "fp8_block":
        wa, sshape, zdt = cta.QuantizationArgs(num_bits=8, type="float", strategy="block", block_structure=[16, 128], symmetric=True), lambda r, c: (r // 16, c // 128), F8
    elif fmt == "nvfp4":
        wa, sshape, zdt = cta.QuantizationArgs(num_bits=4, type="float", strategy="tensor_group", symmetric=True, group_size=16, scale_dtype=F8), lambda r, c: (r, c // 16), None
    elif fmt == "mxfp8":
        wa, sshape, zdt = cta.QuantizationArgs(num_bits=8, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=torch.uint8), lambda r, c: (r, c // 32), F8
    else:
        wa, sshape, zdt = cta.QuantizationArgs(num_bits=4, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=torch.uint8), lambda r, c: (r, c // 32), None
    ia = cta.QuantizationArgs(num_bits=8, type="float", strategy="tensor", symmetric=True, dynamic=True) if fmt.startswith("fp8") else None
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa, input_activations=ia)
    if fmt in ("nvfp4", "mxfp4"):
        scheme.format = fmt + "-pack-quantized"
    if fmt == "mxfp8":
        scheme.format = "mxfp8-quantized"
    root = torch.nn.Module()
    root.blocks = torch.nn.ModuleList()
    for r, c in [(64, 256), (32, 512), (96, 128)]:
        lin = torch.nn.Linear(c, r, bias=False, device="meta")
        lin.weight = torch.nn.Parameter(torch.zeros(r, c, dtype=torch.bfloat16), requires_grad=True)
        lin.weight_scale = torch.nn.Parameter(torch.ones(sshape(r, c), dtype=torch.bfloat16), requires_grad=False)
        if zdt is not None:
            lin.weight_zero_point = torch.nn.Parameter(torch.zeros(sshape(r, c), dtype=zdt), requires_grad=False)
        if fmt == "nvfp4":
            lin.weight_global_scale = torch.nn.Parameter(torch.ones(1), requires_grad=False)
        lin.quantization_scheme = scheme
        root.blocks.append(lin)
    taken = []
    monkeypatch.setattr(codec, "launch_q8_words", lambda words, n, direction, *a_, **k: taken.append((direction, n)))
    monkeypatch.setattr(codec, "launch_fp4_words", lambda words, n, direction, *a_, **k: taken.append((direction, n)))
    scale_tables = []
    monkeypatch.setattr(codec, "launch_mx_scale_words", lambda words, n, direction, *a_, **k: scale_tables.append((direction, n, len(taken))))
    monkeypatch.setattr(codec, "_mx_code_table", lambda dt, dev: torch.zeros(65536, dtype=torch.uint8))
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    hp.set_allow_cpu(True)
    try:
        mc = cta.ModelCompressor()
        states = []
        for _ in range(2):
            mc.compress_model(root)
            states.append([_module_state_no_ptr(m) for m in root.blocks])
            mc.decompress_model(root)
            states.append([_module_state_no_ptr(m) for m in root.blocks])
        assert [t for t in taken] == [("compress", 3), ("decompress", 3)] * 2, taken
        if fmt == "mxfp8":  # one scale table per weights' table; on the way back it goes out BEFORE the weights' (which read the bfloat16 scales it writes)
            assert [(d_, n_) for d_, n_, _ in scale_tables] == [("compress", 3), ("decompress", 3)] * 2
            assert [k for d_, _, k in scale_tables if d_ == "decompress"] == [1, 3]
            assert all(m.weight_scale.dtype is torch.bfloat16 and m.weight.dtype is torch.bfloat16 for m in root.blocks)
        # (same entries; `weight_shape`, which a decompress keeps, stays where it is in the second round — as upstream's compress, which assigns to the existing key)
        norm = lambda st: [(sorted(p_, key=lambda kv: kv[0]), b_) for p_, b_ in st]
        assert norm(states[0]) == norm(states[2]) and norm(states[1]) == norm(states[3])
    finally:
        hp.set_allow_cpu(False)


@pytest.mark.parametrize("variant", ["w3_group", "w6_channel", "w2_asymmetric", "w3_odd_class", "w5_trainable_scale"])
def test_cpp_host_loop_of_the_other_word_widths_matches_the_python_loop(cta, monkeypatch, variant):
    """csrc/host/ct_hostpath.cpp wb_compress_modules / wb_decompress_modules (pack-quantized with 2 / 3 / 5 / 6 / 7 bits: one launch per module by address, the
    dictionary rewritten behind it) against the Python loop on CPU tensors (launches skipped / stubbed): same modules taken, every module left in the same
    state over two rounds of compress -> decompress"""
    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd import codec
    from compressed_tensors_amd.compressors.pack_quantized import base as pq
    from compressed_tensors_amd.quantization.quant_args import QuantizationStatus

    hp = ctlib.hostpath()
    assert hp is not None and hasattr(hp, "wb_compress_modules"), "the host extension was not built (python -c 'import __graft_entry__ as g; g.build()')"
    bits = int(variant[1])
    st = "channel" if "channel" in variant else "group"
    wa = cta.QuantizationArgs(num_bits=bits, type="int", strategy=st, group_size=128 if st == "group" else None, symmetric="asymmetric" not in variant)
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=wa)

    class Odd(torch.nn.Linear):
        def __setattr__(self, name, value):
            super().__setattr__(name, value)

    def tree():
        mods = []
        for k, (r, c) in enumerate([(64, 256), (32, 512), (96, 128), (8, 384)]):
            lin = (Odd if variant == "w3_odd_class" and k == 1 else torch.nn.Linear)(c, r, bias=False, device="meta")
            lin.weight = torch.nn.Parameter(torch.zeros(r, c, dtype=torch.bfloat16), requires_grad=True)
            sshape = (r, 1) if st == "channel" else (r, c // 128)
            lin.weight_scale = torch.nn.Parameter(torch.ones(sshape, dtype=torch.bfloat16), requires_grad=variant == "w5_trainable_scale" and k == 2)
            lin.weight_zero_point = torch.nn.Parameter(torch.zeros(sshape, dtype=torch.int8), requires_grad=False)
            lin.quantization_scheme = scheme
            mods.append(lin)
        return mods

    calls = {"n": 0}

    def fake_pack(w, *a_, **k):
        calls["n"] += 1
        return torch.zeros(w.shape[0], -(-w.shape[1] * bits // 32), dtype=torch.int32)

    def fake_unpack(p_, shape, scale, *a_, **k):
        calls["n"] += 1
        return torch.zeros(tuple(shape), dtype=scale.dtype)

    monkeypatch.setattr(codec, "quantize_and_pack", fake_pack)
    monkeypatch.setattr(codec, "unpack_and_dequantize", fake_unpack)
    monkeypatch.setattr(codec, "quantize_and_pack_with_zp", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "unpack_and_dequantize_with_zp", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "pack_to_int32", lambda zp, b_, packed_dim=1: torch.zeros((zp.shape[0] * b_ + 31) // 32, zp.shape[1], dtype=torch.int32))
    monkeypatch.setattr(codec, "unpack_from_int32", lambda p_, b_, shape, packed_dim=1: torch.zeros(tuple(shape), dtype=torch.int8))
    monkeypatch.setattr(codec, "launch_w4_words", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "launch_zp4_words", lambda *a_, **k: None)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(ctlib, "stream_on", lambda device, handle=None: 0)
    hp.set_allow_cpu(True)
    try:
        a, b = tree(), tree()
        for rnd in range(2):
            for direction, status in (("compress", QuantizationStatus.COMPRESSED), ("decompress", QuantizationStatus.DECOMPRESSED)):
                before = calls["n"]
                getattr(pq.PackedQuantizationCompressor, direction + "_modules")(a)
                left = calls["n"] - before  # modules the C++ loop handed back (each makes one codec call in the Python loop)
                monkeypatch.setattr(ctlib, "_HOSTPATH", [None])
                getattr(pq.PackedQuantizationCompressor, direction + "_modules")(b)
                monkeypatch.setattr(ctlib, "_HOSTPATH", [hp])
                for x, y in zip(a, b):
                    assert _module_state_no_ptr(x) == _module_state_no_ptr(y), (variant, rnd, direction)
                    assert x.quantization_status == status == y.quantization_status
                expect = {"w3_odd_class": 1, "w5_trainable_scale": 1 if (rnd == 0 and direction == "compress") else 0}.get(variant, 0)  # (asymmetric too: C++)
                assert left == expect, (variant, rnd, direction, left)
    finally:
        hp.set_allow_cpu(False)


def _module_state_no_ptr(m):
    return ([(k, None if v is None else (type(v).__name__, v.requires_grad, tuple(v.shape), v.dtype)) for k, v in m._parameters.items()],
            [(k, None if v is None else (type(v).__name__, tuple(v.shape))) for k, v in m._buffers.items()])


def test_cpp_waiting_calls_on_a_stub_abi(cta):
    """csrc/host/ct_hostpath.cpp:bitmask_compress / marlin24_w4_full — the two plug-in calls that wait for the device, without the
    interpreter — against stand-ins for the five C-ABI entries they call (ctypes callbacks; CPU tensors): what is allocated, what is
    handed to the launch, the pending word, the keep-the-view rule for `values`, the verdict, and a non-zero status coming back for
    _lib.check.  The kernels behind the real entries are the GPU suite's business."""
    import numpy as np

    from compressed_tensors_amd import _lib as ctlib

    hp = ctlib.hostpath()
    assert hp is not None, "the host extension was not built (python -c 'import __graft_entry__ as g; g.build()')"
    seen = {}

    def bm(x, dt, rows, cols, values, cap, bitmask, ro, total, ws, ws_bytes, stream):
        seen["bm"] = (dt, rows, cols, cap, ws_bytes, stream, ctypes.c_int64.from_address(total).value)
        if seen.get("fail"):
            return ctlib.CT_ERR_INVALID_ARG
        a = np.ctypeslib.as_array(ctypes.cast(x, ctypes.POINTER(ctypes.c_int16)), (rows, cols))
        keep = a != 0
        nnz = int(keep.sum())
        np.ctypeslib.as_array(ctypes.cast(values, ctypes.POINTER(ctypes.c_int16)), (cap,))[:nnz] = a[keep]
        np.ctypeslib.as_array(ctypes.cast(bitmask, ctypes.POINTER(ctypes.c_uint8)), (rows, (cols + 7) // 8))[:] = np.packbits(keep, axis=1, bitorder="little")
        np.ctypeslib.as_array(ctypes.cast(ro, ctypes.POINTER(ctypes.c_int64)), (rows,))[:] = np.concatenate([[0], np.cumsum(keep.sum(1))[:-1]])
        ctypes.c_int64.from_address(total).value = nnz
        return 0

    def wait(word, pending, stream, out):
        ctypes.c_int64.from_address(out).value = ctypes.c_int64.from_address(word).value
        return 0

    def marlin(w, wdt, s, sdt, zp, zdt, m, k, g, perm, packed, meta, sp, bad, clear, stream):
        seen["marlin"] = (wdt, sdt, zp, zdt, m, k, g, perm, clear, stream, ctypes.c_int64.from_address(bad).value)
        np.ctypeslib.as_array(ctypes.cast(packed, ctypes.POINTER(ctypes.c_int32)), (k // 32, 2 * m))[:] = 7
        if seen.get("violate"):
            ctypes.c_int32.from_address(bad).value = 1
        return 0

    V, L, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    cbs = {
        "ct_bitmask_compress": ctypes.CFUNCTYPE(I, V, I, L, L, V, L, V, V, V, V, L, V)(bm),
        "ct_bitmask_compress_workspace_bytes": ctypes.CFUNCTYPE(L, L, L)(lambda r, c: 4096 + 8 * r),
        "ct_mailbox_wait_i64": ctypes.CFUNCTYPE(I, V, L, V, V)(wait),
        "ct_stream_wait": ctypes.CFUNCTYPE(I, V)(lambda s: 0),
        "ct_marlin24_compress_w4_full": ctypes.CFUNCTYPE(I, V, I, V, I, V, I, L, L, L, I, V, V, V, V, I, V)(marlin),
    }
    box = (ctypes.c_int64 * 8)()
    host = ctypes.addressof(box)
    WS = 0x7000  # stands for the caller's verdict workspace (only handed through)
    hp.bind_abi({k: ctypes.cast(v, ctypes.c_void_p).value for k, v in cbs.items()})
    hp.set_allow_cpu(True)
    try:
        g = torch.Generator().manual_seed(3)
        for density, exact in ((0.5, True), (0.5, False), (0.2, True), (0.2, False)):
            x = torch.randint(1, 1000, (48, 96), generator=g, dtype=torch.int16) * (torch.rand(48, 96, generator=g) < density)
            status, values, bitmask, ro = hp.bitmask_compress(x, 7, host, host, 1234, exact)
            assert status == 0 and seen["bm"] == (7, 48, 96, 48 * 96, 4096 + 8 * 48, 1234, -1)  # the pending word was set before the launch
            keep = x != 0
            assert values.dtype == x.dtype and torch.equal(values, x[keep])
            assert torch.equal(bitmask, torch.from_numpy(np.packbits(keep.numpy(), axis=1, bitorder="little")))
            assert torch.equal(ro, torch.cumsum(keep.sum(1), 0) - keep.sum(1))
            # exact (round 6, the default of the Python callers): `values` owns nnz elements, like tensor[mask]; otherwise a view of the worst-case buffer
            assert values.untyped_storage().nbytes() == (2 * int(keep.sum()) if exact else 2 * x.numel())
        assert hp.bitmask_compress(x.t(), 7, host, host, 0, True) is None and hp.bitmask_compress(x[:0], 7, host, host, 0, True) is None  # the Python path's cases
        seen["fail"] = True
        status, *rest = hp.bitmask_compress(x, 7, host, host, 0, True)
        assert status == ctlib.CT_ERR_INVALID_ARG and rest == [None, None, None]

        # the batched form (round 6): a window of tensors = ONE table launch (ct_bitmask_compress_batch), one mailbox word per tensor, results sized
        # after the window and — in exact mode — filled by ONE batched copy; declined tensors come back as None
        seen.pop("fail")
        xs = [torch.randint(1, 1000, (16 + 8 * i, 64), generator=g, dtype=torch.int16) * (torch.rand(16 + 8 * i, 64, generator=g) < 0.2 + 0.15 * i) for i in range(5)]
        xs.insert(2, xs[0].t())  # a view: not taken
        xs.append(torch.randint(1, 9, (8, 64), generator=g, dtype=torch.int32))  # another element size: a window of its own
        launches, copies = [], []

        def batch_plan(items, n, ws_bytes):
            rows = np.ctypeslib.as_array(ctypes.cast(items, ctypes.POINTER(ctypes.c_int64)), (n, 15))
            if seen.get("fail"):
                return -1
            rows[:, 9] = np.arange(n)  # first_block: one "workgroup" per item in this stand-in
            ctypes.c_int64.from_address(ws_bytes).value = 64 * n
            return n

        def batch_launch(items, n, blocks, es, ws, ws_bytes, stream):
            rows = np.ctypeslib.as_array(ctypes.cast(items, ctypes.POINTER(ctypes.c_int64)), (n, 15)).copy()
            launches.append((n, es, [int(t - host) for t in rows[:, 4]], stream, ws_bytes))
            for r in rows:
                x, values, bitmask, ro, total, nrows, ncols, cap, dt = (int(v) for v in r[:9])
                ctype, npdt = (ctypes.c_int16, np.int16) if es == 2 else (ctypes.c_int32, np.int32)
                a = np.ctypeslib.as_array(ctypes.cast(x, ctypes.POINTER(ctype)), (nrows, ncols))
                keep = a != 0
                nnz = int(keep.sum())
                np.ctypeslib.as_array(ctypes.cast(values, ctypes.POINTER(ctype)), (cap,))[:nnz] = a[keep]
                np.ctypeslib.as_array(ctypes.cast(bitmask, ctypes.POINTER(ctypes.c_uint8)), (nrows, (ncols + 7) // 8))[:] = np.packbits(keep, axis=1, bitorder="little")
                np.ctypeslib.as_array(ctypes.cast(ro, ctypes.POINTER(ctypes.c_int64)), (nrows,))[:] = np.concatenate([[0], np.cumsum(keep.sum(1))[:-1]])
                ctypes.c_int64.from_address(total).value = nnz
            return 0

        def copy_plan(items, n):
            rows = np.ctypeslib.as_array(ctypes.cast(items, ctypes.POINTER(ctypes.c_int64)), (n, 4))
            rows[:, 3] = np.arange(n)
            return n

        def copy_launch(items, n, blocks, stream):
            rows = np.ctypeslib.as_array(ctypes.cast(items, ctypes.POINTER(ctypes.c_int64)), (n, 4))
            copies.append(n)
            for src, dst, nbytes, _ in rows.tolist():
                ctypes.memmove(dst, src, nbytes)
            return 0

        cbs["ct_bitmask_batch_plan"] = ctypes.CFUNCTYPE(L, V, I, V)(batch_plan)
        cbs["ct_bitmask_compress_batch"] = ctypes.CFUNCTYPE(I, V, I, L, I, V, L, V)(batch_launch)
        cbs["ct_copy_batch_plan"] = ctypes.CFUNCTYPE(L, V, I)(copy_plan)
        cbs["ct_copy_batch"] = ctypes.CFUNCTYPE(I, V, I, L, V)(copy_launch)
        hp.bind_abi({k: ctypes.cast(v, ctypes.c_void_p).value for k, v in cbs.items()})
        dts = [7] * 6 + [4]  # (int16 rides the code of int64 in this stand-in's `dt` slot: it is only handed through; int32: 4)
        for exact in (True, False):
            for budget in (1 << 30, 1):  # 1 byte: every window holds exactly one tensor
                launches.clear()
                copies.clear()
                r = hp.bitmask_compress_many(xs, dts, host, host, 2, 6, 1234, exact, budget)
                assert r[0] == 0 and len(r) == 1 + len(xs) and r[3] is None
                got = [(n, es, words) for n, es, words, _, _ in launches]
                if budget > 1:  # windows of <= 3 tensors, one element size each; two in flight: they take the two halves of the six words in turn
                    assert got == [(3, 2, [16, 24, 32]), (2, 2, [40, 48]), (1, 4, [16])], got
                else:
                    assert got == [(1, 2, [16]), (1, 2, [40]), (1, 2, [16]), (1, 2, [40]), (1, 2, [16]), (1, 4, [40])], got
                assert all(st == 1234 for _, _, _, st, _ in launches)
                assert copies == ([n for n, _, _ in got] if exact else [])  # one batched copy per window, none for views
                for x, res in zip(xs, r[1:]):
                    if res is None:
                        continue
                    values, bitmask, ro = res
                    keep = x != 0
                    assert values.dtype == x.dtype and torch.equal(values, x[keep]) and torch.equal(ro, torch.cumsum(keep.sum(1), 0) - keep.sum(1))
                    assert torch.equal(bitmask, torch.from_numpy(np.packbits(keep.numpy(), axis=1, bitorder="little")))
                    assert values.untyped_storage().nbytes() == (x.element_size() * int(keep.sum()) if exact else x.element_size() * x.numel())
        # ... and the way back (round 6): a list of compressed tensors -> ONE table launch per element size, outputs allocated by the loop, nothing waited for
        dlaunches = []

        def dplan(items, n):
            rows = np.ctypeslib.as_array(ctypes.cast(items, ctypes.POINTER(ctypes.c_int64)), (n, 9))
            rows[:, 8] = np.arange(n)
            return n

        def dlaunch(items, n, blocks, es, stream):
            rows = np.ctypeslib.as_array(ctypes.cast(items, ctypes.POINTER(ctypes.c_int64)), (n, 9)).copy()
            dlaunches.append((n, es, stream))
            ctype = ctypes.c_int16 if es == 2 else ctypes.c_int32
            for values, bitmask, ro, outp, nrows, ncols, vlen, dt, _ in (tuple(int(v) for v in r) for r in rows):
                m = np.unpackbits(np.ctypeslib.as_array(ctypes.cast(bitmask, ctypes.POINTER(ctypes.c_uint8)), (nrows, (ncols + 7) // 8)), axis=1, bitorder="little")[:, :ncols].astype(bool)
                o = np.ctypeslib.as_array(ctypes.cast(outp, ctypes.POINTER(ctype)), (nrows, ncols))
                o[:] = 0
                if vlen:
                    o[m] = np.ctypeslib.as_array(ctypes.cast(values, ctypes.POINTER(ctype)), (vlen,))[: int(m.sum())]
            return 0

        cbs["ct_bitmask_decompress_batch_plan"] = ctypes.CFUNCTYPE(L, V, I)(dplan)
        cbs["ct_bitmask_decompress_batch"] = ctypes.CFUNCTYPE(I, V, I, L, I, V)(dlaunch)
        hp.bind_abi({k: ctypes.cast(v, ctypes.c_void_p).value for k, v in cbs.items()})
        good = [x for x in xs if x.is_contiguous()]
        comp = [(x[x != 0], torch.from_numpy(np.packbits((x != 0).numpy(), axis=1, bitorder="little")), torch.cumsum((x != 0).sum(1), 0) - (x != 0).sum(1), list(x.shape)) for x in good]
        ro_list = [c[2] for c in comp]
        ro_list[1] = None  # no row offsets: not taken
        r = hp.bitmask_decompress_many([c[0] for c in comp], [c[1] for c in comp], ro_list, [c[3] for c in comp], [7] * 5 + [4], 99)
        assert r[0] == 0 and r[2] is None and dlaunches == [(4, 2, 99), (1, 4, 99)]  # one launch per element size
        for x, res in zip(good, r[1:]):
            assert res is None or (res.dtype == x.dtype and torch.equal(res, x))
        assert hp.bitmask_decompress_many([comp[0][0]], [comp[0][1]], [comp[0][2]], [comp[0][3]], [-1], 0) == [0, None]
        assert hp.bitmask_compress_many(xs, [7, 7, 7, -1, 7, 7, 4], host, host, 2, 6, 0, True, 1 << 20)[4] is None  # no element code: left to the caller
        assert hp.bitmask_compress_many([xs[0][:, :36].contiguous()], [7], host, host, 2, 6, 0, True, 1 << 20) == [0, None]  # rows that are not whole 16-byte units: the single-tensor path
        seen["fail"] = True
        assert hp.bitmask_compress_many(xs, dts, host, host, 2, 6, 0, True, 1 << 20) == [ctlib.CT_ERR_INVALID_ARG]

        w = torch.zeros(64, 256, dtype=torch.bfloat16)
        s = torch.ones(64, 2, dtype=torch.bfloat16)
        for violate in (False, True):
            seen["violate"] = violate
            box[1] = 99
            status, violated, packed, meta, sp = hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, WS, 77)
            assert status == 0 and violated is violate
            assert seen["marlin"] == (2, 2, None, -1, 64, 256, 128, 1, 0, 77, 0)  # the verdict word was cleared before the launch; the kernel only ORs into it
            assert packed.shape == (8, 128) and packed.dtype == torch.int32 and int(packed[0, 0]) == 7
            assert meta.shape == (8, 128) and meta.dtype == torch.int16 and sp.shape == (2, 64) and sp.dtype == torch.float16
        # with the HIP runtime's own hipStreamSynchronize bound, the verdict waits on it; if it reports an error, ct_stream_wait is asked (and reports)
        waits = {"sync": 0, "query": 0, "sync_rc": 0}

        def sync(stream):
            waits["sync"] += 1
            return waits["sync_rc"]

        def query_wait(stream):
            waits["query"] += 1
            return 0

        cbs["hipStreamSynchronize"] = ctypes.CFUNCTYPE(I, V)(sync)
        cbs["ct_stream_wait"] = ctypes.CFUNCTYPE(I, V)(query_wait)
        hp.bind_abi({k: ctypes.cast(v, ctypes.c_void_p).value for k, v in cbs.items()})
        assert hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, WS, 77)[0] == 0 and waits == {"sync": 1, "query": 0, "sync_rc": 0}
        waits["sync_rc"] = 700
        assert hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, WS, 77)[0] == 0 and (waits["sync"], waits["query"]) == (2, 1)
        hp.set_wait_mode(0)
        assert hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, WS, 77)[0] == 0 and (waits["sync"], waits["query"]) == (2, 2)
        hp.set_wait_mode(1)
        # round 5: with `ct_marlin24_compress_w4_verdict` bound, the verdict comes from the word the launch stores (1 = 2:4 holds, 3 = violated) — no
        # stream wait at all; CT_ERR_UNSUPPORTED (a layout outside the one-launch kernel) falls back to the full entry + the stream wait; a launch
        # that "finishes" without a verdict is an error, never a silent pass
        vseen = {"calls": 0, "mode": "ok"}

        def verdict(w_, wdt, s_, sdt, zp_, zdt, m, k, g, perm, packed, meta, sp, word, ws, clear_ws, stream):
            vseen["calls"] += 1
            vseen["args"] = (wdt, sdt, zp_, zdt, m, k, g, perm, stream, ctypes.c_int64.from_address(word).value)
            assert (ws, clear_ws) == (WS, 0)  # round 6: the ticket tree is the caller's, kept zero by the launches themselves
            if vseen["mode"] == "unsupported":
                return ctlib.CT_ERR_UNSUPPORTED
            if vseen["mode"] != "silent":
                ctypes.c_int64.from_address(word).value = 3 if vseen["mode"] == "violated" else 1
            return 0

        cbs["ct_marlin24_compress_w4_verdict"] = ctypes.CFUNCTYPE(I, V, I, V, I, V, I, L, L, L, I, V, V, V, V, V, I, V)(verdict)
        hp.bind_abi({k: ctypes.cast(v, ctypes.c_void_p).value for k, v in cbs.items()})
        waits.update(sync=0, query=0, sync_rc=0)
        seen.pop("marlin", None)
        for mode, want in (("ok", False), ("violated", True)):
            vseen["mode"] = mode
            box[1] = 99
            status, violated, packed, meta, sp = hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, WS, 77)
            assert status == 0 and violated is want and vseen["args"] == (2, 2, None, -1, 64, 256, 128, 1, 77, 0)  # the word was zeroed before the launch
        assert "marlin" not in seen and waits == {"sync": 0, "query": 0, "sync_rc": 0}  # neither the flag entry nor any stream wait
        calls = vseen["calls"]
        assert hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, 0, 77)[0] == 0  # no workspace: the flag entry + the stream wait
        assert vseen["calls"] == calls and "marlin" in seen and waits["sync"] == 1
        waits.update(sync=0)
        seen.pop("marlin", None)
        vseen["mode"] = "unsupported"
        seen["violate"] = True
        status, violated, *_ = hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, WS, 77)
        assert status == 0 and violated is True and "marlin" in seen and waits["sync"] == 1
        seen["violate"] = False
        vseen["mode"] = "silent"  # the stub's ct_mailbox_wait_i64 hands back the word as it is: still 0 -> not a verdict
        with pytest.raises(RuntimeError, match="without a 2:4 verdict"):
            hp.marlin24_w4_full(w, 2, s, 2, None, -1, 128, True, host + 8, host + 8, WS, 77)
        vseen["mode"] = "ok"
        del cbs["ct_marlin24_compress_w4_verdict"]
        hp.bind_abi({k: ctypes.cast(v, ctypes.c_void_p).value for k, v in cbs.items()})  # an older library without the entry: the full entry + wait
        # the same call from the state-dict entries on (Marlin24Compressor.compress): layout tests, element codes, group / permutation choice
        seen["violate"] = False
        D = hp.marlin24_compress_default
        assert D(w, s, None, 128, host + 8, host + 8, WS, 5)[0] == 0 and seen["marlin"][:9] == (2, 2, None, -1, 64, 256, 128, 0, 0)  # 128 == k / 2: single-column permutation
        s64 = torch.ones(64, 4, dtype=torch.float16)
        zp = torch.zeros(64, 4, dtype=torch.int8)
        assert D(w, s64, zp, 64, host + 8, host + 8, WS, 5)[0] == 0
        assert seen["marlin"][:2] == (2, 1) and seen["marlin"][2] == zp.data_ptr() and seen["marlin"][3:9] == (3, 64, 256, 64, 1, 0)
        assert D(w, torch.ones(64, dtype=torch.bfloat16), None, 0, host + 8, host + 8, WS, 5)[0] == 0 and seen["marlin"][4:8] == (64, 256, 256, 0)  # channel-wise, 1-D scale
        for args in ((w[:32], s[:32], None, 128), (w[:, :128], s[:, :1], None, 128), (w.float(), s, None, 128), (w, s.float(), None, 128), (w, s, None, -1),
                     (w, s, None, 96), (w.t(), s, None, 128), (w, s.t(), None, 128), (w, s, torch.zeros(64, 2, dtype=torch.float64), 128),
                     (w[:, 8:264], s, None, 128)):
            assert D(*args, host + 8, host + 8, WS, 5) is None, [tuple(a.shape) if hasattr(a, "shape") else a for a in args]
    finally:
        hp.set_allow_cpu(False)
        ctlib._HOSTPATH.clear()  # the next hostpath() binds the real entries again
        assert ctlib.hostpath() is hp


def test_cpp_host_loop_fuzz_against_the_python_loop(cta, monkeypatch):
    """random module trees — symmetric / asymmetric, group / channel / odd group sizes, zero points present / absent / as buffers / of the
    wrong dtype, trainable scales, classes with their own __setattr__, activation ordering, fp32 weights, ragged shapes, activation
    schemes, already-compressed modules — through compress_modules + decompress_modules with the C++ loop in front and with the Python
    loop alone: every module must end in the same state (names, order, kinds, trainability, shapes, dtypes, status) and the same work
    must reach the launches.  CPU tensors, launches stubbed: this is about the host logic that decides and rewrites."""
    import copy
    import random

    from compressed_tensors_amd import _lib as ctlib
    from compressed_tensors_amd import codec
    from compressed_tensors_amd.compressors.pack_quantized import base as pq

    hp = ctlib.hostpath()
    assert hp is not None
    tables = {"cpp": [], "py": []}
    which = {"now": "cpp"}
    rec = lambda *t: tables[which["now"]].append(t)
    IW = codec._ITEM_WORDS  # (rows, cols, group, carries the stored form of its zero points)
    monkeypatch.setattr(codec, "launch_w4_words", lambda words, n, direction, dtype, device: n and rec(direction, sorted((*r[4:7], bool(r[10])) for r in words.reshape(n, IW).tolist())))
    monkeypatch.setattr(codec, "launch_zp4_words", lambda words, n, direction, device: n and rec("zp-" + direction, sorted(map(tuple, words.reshape(n, IW)[:, 4:6].tolist()))))
    # the single-module one-launch forms decline (None): the per-module path composes the stubs below, as it does for a layout they do not take
    monkeypatch.setattr(codec, "quantize_and_pack_with_zp", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "unpack_and_dequantize_with_zp", lambda *a_, **k: None)
    monkeypatch.setattr(codec, "zp4_batch", lambda pairs, direction: (lambda ps: ps and rec("zp-" + direction, sorted(tuple((s if direction == "pack" else d).shape) for s, d in ps)))(list(pairs)))

    class FakeBatch:
        def __init__(self, entries, direction, dtype, kind="w4", bits=8):
            self.rec = (direction, sorted((int(e[4]), int(e[5]), int(e[6]), len(e) > 7 and e[7] is not None) for e in entries))

        def launch(self, stream=None):
            if self.rec[1]:
                rec(*self.rec)

    monkeypatch.setattr(codec, "W4Batch", FakeBatch)
    monkeypatch.setattr(codec, "pack_to_int32", lambda v, bits, packed_dim=1: torch.zeros(((v.shape[0] * bits + 31) // 32, v.shape[1]) if packed_dim == 0 else (v.shape[0], (v.shape[1] * bits + 31) // 32), dtype=torch.int32))
    monkeypatch.setattr(codec, "unpack_from_int32", lambda p, bits, shape, packed_dim=1: torch.zeros(tuple(shape), dtype=torch.int8))
    monkeypatch.setattr(codec, "quantize_and_pack", lambda w, *a_, **k: torch.zeros(w.shape[0], (w.shape[1] * 4 + 31) // 32, dtype=torch.int32))
    monkeypatch.setattr(codec, "unpack_and_dequantize", lambda p, shape, scale, *a_, **k: torch.zeros(tuple(shape), dtype=scale.dtype))
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)

    class Odd(torch.nn.Linear):
        def __setattr__(self, name, value):
            super().__setattr__(name, value)

    rng = random.Random(20260927)

    def build():
        schemes = []
        for _ in range(rng.randint(1, 3)):
            sym = rng.random() < 0.5
            if rng.random() < 0.25:
                wa = cta.QuantizationArgs(num_bits=4, symmetric=sym, strategy="channel")
            else:
                wa = cta.QuantizationArgs(num_bits=4, group_size=rng.choice([32, 64, 128, 128, 40]), symmetric=sym, strategy="group")
            ia = cta.QuantizationArgs(num_bits=8, symmetric=True, strategy="tensor") if rng.random() < 0.1 else None
            schemes.append(cta.QuantizationScheme(targets=["Linear"], weights=wa, input_activations=ia))
        root = torch.nn.Module()
        root.blocks = torch.nn.ModuleList()
        for _ in range(rng.randint(1, 9)):
            scheme = rng.choice(schemes)
            wa = scheme.weights
            g = wa.group_size if wa.strategy == "group" else None
            r = rng.choice([16, 20, 32, 64])
            c = rng.choice([128, 256, 384, 640]) if rng.random() < 0.9 else 96
            if g and c % g:
                c = g * rng.randint(1, 4)
            wdt = torch.float32 if rng.random() < 0.08 else rng.choice([torch.bfloat16, torch.float16])
            lin = (Odd if rng.random() < 0.08 else torch.nn.Linear)(c, r, bias=rng.random() < 0.2, device="meta")
            if lin.bias is not None:
                lin.bias = torch.nn.Parameter(torch.zeros(r, dtype=wdt))
            lin.weight = torch.nn.Parameter(torch.zeros(r, c, dtype=wdt), requires_grad=rng.random() < 0.5)
            cols = c // g if g else 1
            lin.weight_scale = torch.nn.Parameter(torch.ones(r, cols, dtype=wdt), requires_grad=rng.random() < 0.1)
            zp = torch.zeros(r, cols, dtype=torch.int8 if rng.random() < 0.92 else torch.int32)
            roll = rng.random()
            if roll < 0.1:
                lin.register_buffer("weight_zero_point", zp)
            elif roll < 0.9 or not wa.symmetric:
                lin.weight_zero_point = torch.nn.Parameter(zp, requires_grad=False)
            if rng.random() < 0.06:
                lin.weight_g_idx = torch.nn.Parameter(torch.arange(c, dtype=torch.int32) // (g or c), requires_grad=False)
            lin.quantization_scheme = scheme
            blk = torch.nn.Module()
            blk.proj = lin
            root.blocks.append(blk)
        return root

    def state(m):
        return ([(k, None if v is None else (type(v).__name__, v.requires_grad, tuple(v.shape), v.dtype)) for k, v in m._parameters.items()],
                [(k, None if v is None else (type(v).__name__, tuple(v.shape), v.dtype)) for k, v in m._buffers.items()], getattr(m, "quantization_status", None))

    hp.set_allow_cpu(True)
    taken = 0
    try:
        for trial in range(150):
            a = build()
            b = copy.deepcopy(a)
            for x, y in zip(a.modules(), b.modules()):
                if hasattr(x, "quantization_scheme"):
                    y.quantization_scheme = x.quantization_scheme
            ma = [m for m in a.modules() if isinstance(m, torch.nn.Linear)]
            mb = [m for m in b.modules() if isinstance(m, torch.nn.Linear)]
            for step in ("compress_modules", "decompress_modules"):
                tables["cpp"].clear(); tables["py"].clear()
                which["now"] = "cpp"
                monkeypatch.setattr(ctlib, "_HOSTPATH", [hp])
                getattr(pq.PackedQuantizationCompressor, step)(ma)
                taken += sum(len(t[1]) for t in tables["cpp"] if t[0] in ("compress", "decompress"))
                which["now"] = "py"
                monkeypatch.setattr(ctlib, "_HOSTPATH", [None])
                getattr(pq.PackedQuantizationCompressor, step)(mb)
                for x, y in zip(ma, mb):
                    assert state(x) == state(y), (trial, step, state(x), state(y))
                merged = lambda ts: {d: sorted(r for t in ts if t[0] == d for r in t[1]) for d in {t[0] for t in ts}}
                assert merged(tables["cpp"]) == merged(tables["py"]), (trial, step)
    finally:
        hp.set_allow_cpu(False)
        monkeypatch.setattr(ctlib, "_HOSTPATH", [hp])
    assert taken > 300  # the batched launches really were exercised


def test_the_product_library_carries_no_diagnostic_knobs():
    """the CT_BITMASK_RESIDENT* environment knobs and the time-stamp argument of the resident sparse compress are compiled
    into libct_hip_diag.so only (-DCT_DIAG, loaded by the tests that force the alternative forms); the shipped libct_hip.so reads no
    environment variable of its own and both libraries export the same C ABI"""
    import re
    import subprocess

    from compressed_tensors_amd import _lib as ctlib

    assert os.path.exists(ctlib.LIB_PATH) and os.path.exists(ctlib.DIAG_LIB_PATH)
    with open(ctlib.LIB_PATH, "rb") as f:
        product = f.read()
    with open(ctlib.DIAG_LIB_PATH, "rb") as f:
        diag = f.read()
    assert b"CT_BITMASK_RESIDENT" not in product and b"CT_BITMASK_RESIDENT_WAIT_US" in diag
    assert not re.search(rb"CT_[A-Z0-9_]{4,}\x00", product), "an environment-knob-like string is left in the product library"

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return sorted(line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("ct_"))

    assert exported(ctlib.LIB_PATH) == exported(ctlib.DIAG_LIB_PATH) == sorted(ctlib.EXPORTED_SYMBOLS)
