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

        # ---- a LIST of modules (what the wrapped ModelCompressor loops hand over, see `_wrap_model_compressor`): the GPU modules of a
        # type the kernels implement go through amd_cls's batched launches — one table and one launch per device instead of one
        # launch per module —, the rest one by one through this class (i.e. upstream's code for CPU / meta tensors)
        @classmethod
        def _ct_split(cls, modules, probe_names):
            ours, rest = [], []
            for m in modules:
                # an upstream-OFFLOADED module keeps its tensors in an OffloadCache (a MutableMapping that onloads on access): probing
                # it would pull the weight onto the GPU just to test `.is_cuda`, and the batched path would then hold EVERY module's
                # onloaded tensors until the single launch — the whole model resident where upstream holds one module at a time.  Such
                # modules take upstream's per-module path, untouched.
                if not isinstance(m._parameters, dict) or not isinstance(m._buffers, dict):
                    rest.append(m)
                    continue
                scheme = getattr(m, "quantization_scheme", None)
                probe = None
                for n in probe_names:
                    probe = m._parameters.get(n)
                    if probe is None:
                        probe = m._buffers.get(n)
                    if probe is not None:
                        break
                (ours if (fp4_group is None and scheme is not None and _int_weights(scheme) and _on_gpu(probe)) else rest).append(m)
            return ours, rest

        @classmethod
        def compress_modules(cls, modules, status=None):
            ours, rest = cls._ct_split(modules, ("weight",))
            if ours:
                amd_cls.compress_modules(ours)
                if status is not None:
                    for m in ours:
                        m.__dict__["quantization_status"] = status  # the host library's own enum member, not ours
            for m in rest:
                cls.compress_module(m)

        @classmethod
        def decompress_modules(cls, modules, status=None):
            ours, rest = cls._ct_split(modules, ("weight_packed", "weight"))
            if ours:
                amd_cls.decompress_modules(ours)
                if status is not None:
                    for m in ours:
                        m.__dict__["quantization_status"] = status
            for m in rest:
                cls.decompress_module(m)

    _Hip._ct_batched = True
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


_MC_SAVED = {}  # upstream ModelCompressor's own compress_model / decompress_model while the wrappers are in place


def _wrap_model_compressor() -> None:
    """Upstream's `ModelCompressor.compress_model` / `decompress_model` loop `compress_module` / `decompress_module` over the
    quantized modules (model_compressors/model_compressor.py:167-169,196-198): one codec call, one launch and a full state-dict
    replacement per module.  The wrappers keep everything else of those methods — the module filter (:152-164,191-195), the
    distributed branch (:171-173: untouched, handed to the original), the config status (:175-177,200-204) and the decompress
    hook (:179-182,206-207) — and hand each format's module list to the registry class's `compress_modules` /
    `decompress_modules` when it has them (the HIP subclasses: one launch per device), module by module otherwise."""
    import compressed_tensors.compressors.model_compressors.model_compressor as up_mc
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.compressors.format import infer_module_format
    from compressed_tensors.config import CompressionFormat
    from compressed_tensors.quantization import QuantizationScheme, QuantizationStatus
    from compressed_tensors.quantization.utils import is_module_quantized

    MC = up_mc.ModelCompressor
    if _MC_SAVED:
        return
    _MC_SAVED["compress_model"], _MC_SAVED["decompress_model"] = MC.compress_model, MC.decompress_model
    is_distributed = getattr(up_mc, "is_distributed", lambda: False)

    def by_codec(modules, format):
        """[(registry class, [modules])] in first-seen order; the format resolution and `scheme.format` write-back of upstream's
        compress_module / decompress_module (compressors/base.py:185-192,211-218); modules without a scheme are skipped as there"""
        groups, seen = {}, {}
        for m in modules:
            scheme = getattr(m, "quantization_scheme", None)
            if not isinstance(scheme, QuantizationScheme):
                continue
            key = (id(scheme), type(m))
            cls = seen.get(key)
            if cls is None:
                fmt = format or scheme.format or infer_module_format(type(m), scheme)
                scheme.format = CompressionFormat(fmt)
                cls = seen[key] = BaseCompressor.get_value_from_registry(scheme.format.value)
            groups.setdefault(cls, []).append(m)
        return groups.items()

    def compress_model(self, model, skip_compressed: bool = False) -> None:
        if is_distributed():
            return _MC_SAVED["compress_model"](self, model, skip_compressed)
        modules = [m for _, m in model.named_modules(remove_duplicate=True)
                   if is_module_quantized(m) and (not skip_compressed or getattr(m, "quantization_status", None) != QuantizationStatus.COMPRESSED)]
        for cls, ms in by_codec(modules, self.force_compression_format):
            if getattr(cls, "_ct_batched", False):
                cls.compress_modules(ms, status=QuantizationStatus.COMPRESSED)
            else:
                for m in ms:
                    cls.compress_module(m)
        if self.quantization_config is not None:
            self.quantization_config.quantization_status = QuantizationStatus.COMPRESSED
        self.add_decompress_hook(model)

    def decompress_model(self, model) -> None:
        modules = [m for _, m in model.named_modules(remove_duplicate=True) if is_module_quantized(m)]
        for cls, ms in by_codec(modules, self.force_compression_format):
            if getattr(cls, "_ct_batched", False):
                cls.decompress_modules(ms, status=QuantizationStatus.DECOMPRESSED)
            else:
                for m in ms:
                    cls.decompress_module(m)
        if self.quantization_config is not None:
            self.quantization_config.quantization_status = QuantizationStatus.DECOMPRESSED
        self.remove_decompression_hook(model)

    compress_model.__doc__, decompress_model.__doc__ = MC.compress_model.__doc__, MC.decompress_model.__doc__
    MC.compress_model, MC.decompress_model = compress_model, decompress_model


def _unwrap_model_compressor() -> None:
    if not _MC_SAVED:
        return
    import compressed_tensors.compressors.model_compressors.model_compressor as up_mc

    up_mc.ModelCompressor.compress_model = _MC_SAVED.pop("compress_model")
    up_mc.ModelCompressor.decompress_model = _MC_SAVED.pop("decompress_model")


_FN_REBOUND = []  # (module, attribute name, original function)
_FN_SWAP = {}  # id(original function) -> (original, dispatching wrapper); lives from the first install(patch_functions=True) to uninstall()


def _patch_functions() -> None:
    """`install(patch_functions=True)`: the four plain functions of the hot path that have no plug-in point upstream —
    `pack_to_int32` / `unpack_from_int32` (compressors/pack_quantized/helpers.py:20-24,104-109, bound by name at
    pack_quantized/base.py:11-14) and `dequantize` / `fake_quantize` (quantization/lifecycle/forward.py:76-181) — are rebound, in
    every already-imported `compressed_tensors.*` module that holds them, to wrappers that send GPU tensors of a layout the
    kernels implement through `codec` and everything else (CPU / meta tensors, exotic strategies; also a NotImplementedError from
    the HIP path) to the original — the dispatch rule of upstream's ImplBackend (utils/impl_backend.py:113-119)."""
    import functools
    import sys

    import compressed_tensors.compressors.pack_quantized.helpers as up_helpers
    import compressed_tensors.quantization.lifecycle.forward as up_forward

    from .quantization import forward as amd_forward

    floats = (torch.float32, torch.float16, torch.bfloat16)
    plain = ("tensor", "channel", "group", "block")

    def simple_args(args) -> bool:
        if args is None:
            return True
        qt, bits = enum_value(getattr(args, "type", "int")), int(args.num_bits)
        return (enum_value(args.strategy) in plain and ((qt == "int" and 1 <= bits <= 8) or (qt == "float" and bits == 8))
                and not getattr(args, "dynamic", False))

    def dispatching(orig, ours, take):
        @functools.wraps(orig)
        def fn(*args, **kwargs):
            try:
                if take(*args, **kwargs):
                    return ours(*args, **kwargs)
            except NotImplementedError:
                pass
            return orig(*args, **kwargs)

        fn._ct_original = orig
        return fn

    def take_pack(value, num_bits, packed_dim=1):
        return value.is_cuda and value.dtype is torch.int8 and value.dim() >= 2

    def take_unpack(value, num_bits, shape, packed_dim=1):
        return value.is_cuda and value.dtype is torch.int32 and value.dim() >= 2

    def take_dequantize(x_q, scale, zero_point=None, args=None, dtype=None, g_idx=None, global_scale=None):
        return (x_q.is_cuda and x_q.dim() == 2 and global_scale is None and simple_args(args) and scale.dtype in floats
                and x_q.dtype in (torch.int8, torch.float8_e4m3fn))

    def take_fake_quantize(x, scale, zero_point, args, g_idx=None, global_scale=None):
        return x.is_cuda and x.dim() == 2 and global_scale is None and simple_args(args) and x.dtype in floats and scale.dtype in floats

    # every install(patch_functions=True) re-scans sys.modules (as _rebind_names does for the classes): an upstream module imported
    # since the last call holds the ORIGINAL function under its own name and is covered now.  The wrappers are made once (ids of the
    # originals -> wrapper); an attribute that already holds a wrapper is not in the table, so nothing is recorded twice.
    if not _FN_SWAP:
        def orig_of(fn):
            return getattr(fn, "_ct_original", fn)

        for orig, ours, take in ((orig_of(up_helpers.pack_to_int32), codec.pack_to_int32, take_pack),
                                 (orig_of(up_helpers.unpack_from_int32), codec.unpack_from_int32, take_unpack),
                                 (orig_of(up_forward.dequantize), amd_forward.dequantize, take_dequantize),
                                 (orig_of(up_forward.fake_quantize), amd_forward.fake_quantize, take_fake_quantize)):
            _FN_SWAP[id(orig)] = (orig, dispatching(orig, ours, take))
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == "compressed_tensors" or mod_name.startswith("compressed_tensors.")):
            continue
        for attr, val in list(vars(mod).items()):
            hit = _FN_SWAP.get(id(val))
            if hit is not None and callable(val):
                setattr(mod, attr, hit[1])
                _FN_REBOUND.append((mod, attr, hit[0]))


def install(rebind_names: bool = True, wrap_model_compressor: bool = True, patch_functions: bool = False):
    """registry swap + ImplBackend registration
    (+ the by-name bindings of the codec classes inside upstream's own modules unless rebind_names=False;
     + batched launches behind upstream's ModelCompressor.compress_model / decompress_model unless wrap_model_compressor=False;
     + with patch_functions=True the plain functions pack_to_int32 / unpack_from_int32 / dequantize / fake_quantize)."""
    import compressed_tensors  # the upstream package; ImportError if it is not installed
    from compressed_tensors.compressors import BaseCompressor
    from compressed_tensors.registry import registry as up_registry
    from compressed_tensors.utils.impl_backend import ImplBackend

    table = up_registry._REGISTRY[BaseCompressor]
    install_into(table, ImplBackend, _SAVED)
    if rebind_names:
        # every call re-scans sys.modules, so upstream modules imported since the last install() are covered too; an attribute that
        # already holds the subclass is not in the swap table (ids of the ORIGINAL classes), so nothing is recorded twice
        _rebind_names(table, _SAVED)
    if wrap_model_compressor:
        _wrap_model_compressor()
    if patch_functions:
        _patch_functions()
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

    for mod, attr, orig in _REBOUND + _FN_REBOUND:
        setattr(mod, attr, orig)
    _REBOUND.clear()
    _FN_REBOUND.clear()
    _FN_SWAP.clear()
    _unwrap_model_compressor()
    uninstall_from(up_registry._REGISTRY[BaseCompressor], _SAVED)
