"""Drop the MI355X codecs into a live upstream `compressed_tensors` install.

    import compressed_tensors_amd.install as ct_amd
    ct_amd.install()

After this, every upstream caller that looks a codec up by its format string
(`BaseCompressor.get_value_from_registry(...)`: compress_module / decompress_module,
ModelCompressor, transformers' DecompressExperts, CompressedTensorsDequantizer) receives a
subclass of the upstream codec whose `compress` / `decompress` run the HIP kernels whenever
the weight lives on the GPU; anything else (CPU tensors, meta tensors) is handed
to the upstream implementation it inherits from.  `_quantize`, `pack_fp4_to_uint8` and `cast_to_fp4`
are additionally registered as `ImplBackend` backends (upstream utils/impl_backend.py:50-79), which
is the reference's own plug-in point for those functions.

The registry is overwritten directly because re-registering a name with a different class
raises upstream (registry/registry.py:215-223); see SURVEY.md §8b.
"""
import torch

from . import codec
from .quantization.quant_args import enum_value

__all__ = ["install", "uninstall", "install_into", "uninstall_from", "make_hip_subclass", "quantize_backend", "quantize_backend_req"]

_SAVED = {}


def _on_gpu(*tensors) -> bool:
    return any(t is not None and t.is_cuda for t in tensors)


def _int_weights(scheme) -> bool:
    """INT 1..8 bits, or FLOAT 8 bits (float8_e4m3fn): the types the quantize / dequantize kernels implement"""
    w = getattr(scheme, "weights", None)
    if w is None:
        return False
    if enum_value(getattr(w, "type", "int")) == "float":
        return int(w.num_bits) == 8
    return 1 <= int(w.num_bits) <= 8


def _fp4_weights(scheme) -> bool:
    w = getattr(scheme, "weights", None)
    return w is not None and enum_value(getattr(w, "type", "int")) == "float" and int(w.num_bits) == 4


def _fp4_compressible(state_dict, group) -> bool:
    w = state_dict.get("weight")
    return (w is not None and w.is_cuda and w.dim() == 2 and w.dtype in (torch.bfloat16, torch.float16)
            and w.shape[1] % group == 0)


def _fp4_decompressible(state_dict, group) -> bool:
    """the stored layout the FP4 decompress kernel reads: (rows, cols / 2) bytes with one scale per `group` columns.  Anything else —
    upstream infers the group from the scale's shape and accepts e.g. a (1, 4) weight with a (1, 2) scale under group_size 32,
    tests/test_compressors/test_mxfp4_quant.py:60-84 — is upstream's"""
    p, sc = state_dict.get("weight_packed"), state_dict.get("weight_scale")
    if p is None or sc is None or not p.is_cuda or p.dim() != 2 or sc.dim() != 2:
        return False
    cols = p.shape[1] * 2
    return cols % group == 0 and tuple(sc.shape) == (p.shape[0], cols // group)


def make_hip_subclass(up_cls, amd_cls):
    """A subclass of the upstream codec `up_cls` (so `can_compress`, `compression_param_names`, `compress_module`,
    `decompress_module` and any upstream helper are inherited) whose `compress` / `decompress` run `amd_cls`'s HIP
    path for GPU tensors of a type the kernels implement, and upstream's own code for everything else."""
    fp4_group = getattr(amd_cls, "GROUP", None)  # set on the FP4 codecs only

    class _Hip(up_cls):
        @classmethod
        def compress(cls, state_dict, scheme):
            if fp4_group is not None:
                ours = _fp4_weights(scheme) and _fp4_compressible(state_dict, fp4_group)
            else:
                ours = _int_weights(scheme) and _on_gpu(state_dict.get("weight"))
            if ours and fp4_group is not None:
                return amd_cls.compress(state_dict, scheme)  # uses the FP4 class's own scale hooks
            if ours:
                return amd_cls.compress.__func__(cls, state_dict, scheme)
            return up_cls.compress.__func__(cls, state_dict, scheme)

        @classmethod
        def decompress(cls, state_dict, scheme):
            probe = state_dict.get("weight_packed", state_dict.get("weight"))
            ours = _fp4_weights(scheme) if fp4_group is not None else _int_weights(scheme)
            if ours and fp4_group is not None and _fp4_decompressible(state_dict, fp4_group):
                return amd_cls.decompress(state_dict, scheme)
            if fp4_group is not None:
                return up_cls.decompress.__func__(cls, state_dict, scheme)
            if ours and _on_gpu(probe):
                return amd_cls.decompress.__func__(cls, state_dict, scheme)
            return up_cls.decompress.__func__(cls, state_dict, scheme)

    _Hip.__name__ = up_cls.__name__ + "MI355X"
    _Hip.__qualname__ = _Hip.__name__
    return _Hip


def quantize_backend_req(x, scale, zero_point, q_min, q_max, args, dtype=None, global_scale=None) -> bool:
    """`req` of the `_quantize` backend: receives exactly `_quantize`'s arguments (forward_helpers.py:525-534)"""
    return (
        x.is_cuda
        and (enum_value(getattr(args, "type", "int")) == "int" or int(args.num_bits) in (4, 8))
        and (global_scale is None or (enum_value(getattr(args, "type", "int")) == "float" and int(args.num_bits) == 4))
        and x.dtype in (torch.float32, torch.float16, torch.bfloat16)
        and dtype in (None, torch.int8, torch.int32, torch.float8_e4m3fn, torch.float32, torch.float16, torch.bfloat16)
        and x.is_contiguous() and _broadcast_layout(x, scale) is not None
    )


def quantize_backend(x, scale, zero_point, q_min, q_max, args, dtype=None, global_scale=None):
    """HIP `_quantize` (replaces the disabled Triton backend registered at forward_helpers.py:404): `x` arrives already
    reshaped by `_process_group` — (R, G, gs) with scale (R, G, 1) — or as (R, C) with a (R, 1) / one-element scale.
    q_min / q_max are recomputed from `args` on the host (they are 0-dim device tensors upstream: reading them would sync)."""
    layout = _broadcast_layout(x, scale)
    x2 = x.reshape(-1, x.shape[-1]) if layout["strategy"] != "group" else x.reshape(-1, x.shape[-2] * x.shape[-1])
    out = codec.quantize_tensor(
        x2, scale.reshape(layout["scale_shape"]), None if zero_point is None else zero_point.reshape(layout["scale_shape"]),
        num_bits=int(args.num_bits), strategy=layout["strategy"], group_size=layout.get("group_size"),
        qtype=enum_value(getattr(args, "type", "int")), global_scale=global_scale,
        dtype=dtype if dtype is not None else (torch.float32 if global_scale is not None else torch.result_type(x, scale)),
    )
    return out.reshape(x.shape)


_FORMATS = ("pack-quantized", "naive-quantized", "int-quantized", "float-quantized", "mxfp8-quantized",
            "nvfp4-pack-quantized", "mxfp4-pack-quantized")


def _amd_codecs():
    from .compressors.fp4 import MXFP4PackedCompressor, NVFP4PackedCompressor
    from .compressors.mxfp8 import MXFP8QuantizationCompressor
    from .compressors.naive_quantized import NaiveQuantizationCompressor
    from .compressors.pack_quantized import PackedQuantizationCompressor

    return dict(zip(_FORMATS, (PackedQuantizationCompressor, NaiveQuantizationCompressor, NaiveQuantizationCompressor,
                               NaiveQuantizationCompressor, MXFP8QuantizationCompressor, NVFP4PackedCompressor, MXFP4PackedCompressor)))


def install_into(table: dict, impl_backend, saved: dict = None) -> dict:
    """The wiring itself, independent of where the host library lives: `table` is the format-string -> codec-class
    dict that `BaseCompressor.get_value_from_registry` reads (upstream: `registry._REGISTRY[BaseCompressor]`),
    `impl_backend` the `ImplBackend` class whose entrypoints dispatch `_quantize`, `pack_fp4_to_uint8`, `cast_to_fp4`
    (upstream utils/impl_backend.py:50-123).  Returns {format: original class} for `uninstall_from`.  `install()` calls
    this with upstream's objects; the GPU tests call it with stand-ins, because the GPU box has no upstream install."""
    saved = {} if saved is None else saved
    for fmt, amd_cls in _amd_codecs().items():
        if fmt not in table and fmt not in saved:
            continue  # an older upstream without the FP4 codecs
        if fmt not in saved:
            saved[fmt] = table[fmt]
        table[fmt] = make_hip_subclass(saved[fmt], amd_cls)

    def register(op, name, req, fn):
        if name in impl_backend._fn_registry:
            return
        fn = _renamed(fn, name)  # backend __name__s must be globally unique upstream (:126-134)
        impl_backend.register(op, req=req, priority=0)(fn)

    floats = (torch.float32, torch.float16, torch.bfloat16)
    register("_quantize", "_quantize_mi355x", quantize_backend_req, quantize_backend)
    register("pack_fp4_to_uint8", "pack_fp4_to_uint8_mi355x",
             lambda x: x.is_cuda and x.dim() == 2 and x.dtype in floats and x.shape[1] % 2 == 0, codec.pack_fp4_to_uint8)
    register("cast_to_fp4", "cast_to_fp4_mi355x", lambda x: x.is_cuda and x.dtype in floats, codec.cast_to_fp4)
    return saved


def _renamed(fn, name):
    import functools

    @functools.wraps(fn)
    def backend(*args, **kwargs):
        return fn(*args, **kwargs)

    backend.__name__ = backend.__qualname__ = name
    return backend


def uninstall_from(table: dict, saved: dict) -> None:
    for fmt, cls in saved.items():
        table[fmt] = cls
    saved.clear()


_REBOUND = []  # (module, attribute name, original class)


def _rebind_names(table: dict, saved: dict) -> None:
    """callers that reach a codec by NAME instead of through the registry — `from compressed_tensors.compressors import
    PackedQuantizationCompressor`, as upstream's own tests do (tests/test_compressors/test_pack_quant.py:17-20) — get the HIP
    subclass too: every attribute of an already-imported `compressed_tensors.*` module that IS an upstream codec class is pointed at
    its subclass.  (A name imported into some other module before install() keeps the upstream class: that binding is not ours.)"""
    import sys

    swap = {id(orig): (orig, table[fmt]) for fmt, orig in saved.items()}
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == "compressed_tensors" or mod_name.startswith("compressed_tensors.")):
            continue
        for attr, val in list(vars(mod).items()):
            hit = swap.get(id(val))
            if hit is not None and isinstance(val, type):
                setattr(mod, attr, hit[1])
                _REBOUND.append((mod, attr, hit[0]))


def install(rebind_names: bool = True):
    """registry swap + ImplBackend registration (+ the by-name bindings inside upstream's own modules unless rebind_names=False)"""
    import compressed_tensors  # the upstream package; ImportError if it is not installed
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.registry import registry as up_registry
    from compressed_tensors.utils.impl_backend import ImplBackend

    table = up_registry._REGISTRY[BaseCompressor]
    install_into(table, ImplBackend, _SAVED)
    if rebind_names and not _REBOUND:
        _rebind_names(table, _SAVED)
    return compressed_tensors


def _broadcast_layout(x, scale):
    """Recognise the broadcast shapes upstream passes to `_quantize` (forward_helpers.py:154-167):
    group: x (R, G, gs) with scale (R, G, 1); channel: x (R, C) with scale (R, 1); tensor: scale
    with one element.  Returns None for anything else (upstream's eager body then runs)."""
    if scale.numel() == 1 and scale.ndim > 0:
        return {"strategy": "tensor", "scale_shape": (1,)}
    if x.ndim == 3 and scale.ndim == 3 and scale.shape == (x.shape[0], x.shape[1], 1):
        return {"strategy": "group", "group_size": x.shape[2], "scale_shape": (x.shape[0], x.shape[1])}
    if x.ndim == 2 and scale.ndim == 2 and scale.shape == (x.shape[0], 1):
        return {"strategy": "channel", "scale_shape": (x.shape[0], 1)}
    return None


def uninstall():
    if not _SAVED:
        return
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.registry import registry as up_registry

    for mod, attr, orig in _REBOUND:
        setattr(mod, attr, orig)
    _REBOUND.clear()
    uninstall_from(up_registry._REGISTRY[BaseCompressor], _SAVED)
