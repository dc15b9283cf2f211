"""Converter protocol, dependency-aware job planning and the compressed-tensors dequantizer
(reference entrypoints/convert/converters/base.py:19-133, ct_dequantizer.py:21-171).

MI355X design of `CompressedTensorsDequantizer.process`: the compressed tensors of ALL matched modules
of a shard are moved to the GPU, decompressed by ONE batched launch per scheme where the codec allows it
(`BaseCompressor.decompress_many`, W4A16 -> `ct_unpack_dequant_batch`), cast, and copied back into pinned
host buffers on the caller's stream; nothing is decompressed on the CPU."""
import contextlib
import re
import threading
from collections import defaultdict
from typing import Dict, Iterable, List, Optional, Set

import torch

from ...compressors.base import BaseCompressor
from ...compressors.format import infer_module_format
from ...config import CompressionFormat
from ...quantization.quant_args import QuantizationArgs, QuantizationScheme
from .safetensors_io import CONFIG_NAME, find_config_path, get_checkpoint_files, get_quantization_config

__all__ = ["Converter", "build_inverse_weight_maps", "CompressedTensorsDequantizer", "match_name", "match_quantizable_tensors"]

KV_CACHE_PARAM_NAMES = ("k_scale", "v_scale", "q_scale")  # quantization/quant_args KVCacheScaleType values


class Converter:
    """converters/base.py:19-74"""

    def process(self, tensors: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        raise NotImplementedError()

    def validate(self, tensors: Dict[str, torch.Tensor]):
        raise NotImplementedError()

    def create_config(self):
        raise NotImplementedError()

    def get_dependencies(self, weight_name: str) -> Set[str]:
        raise NotImplementedError()


def match_name(name: str, target: str) -> bool:
    """utils/match.py:422-445 (without vLLM fused-module mappings)"""
    if target.startswith("re:"):
        return re.match(target[3:], name) is not None
    return target == name


def match_quantizable_tensors(tensors, ignore: Iterable[str], targets: Iterable[str] = (), param_targets: Iterable[str] = ("weight",),
                              allow_nonquantizable: bool = False):
    """utils/match.py:469-523: (module_name, tensor_name) of every targeted, not ignored tensor"""
    targets, ignore, param_targets = list(targets), list(ignore), list(param_targets)
    for name in list(tensors.keys()):
        module_name, _, param_name = name.rpartition(".")
        if not allow_nonquantizable and module_name.endswith("norm"):
            continue
        if not any(match_name(param_name, t) for t in param_targets):
            continue
        if not (len(targets) == 0 or "Linear" in targets or any(match_name(module_name, t) for t in targets)):
            continue
        if any(match_name(module_name, i) for i in ignore):
            continue
        yield module_name, name


def build_inverse_weight_maps(weight_map: Dict[str, str], model_files: Dict[str, str], converters: List[Converter]):
    """converters/base.py:77-133: for every output shard, exactly which tensors to load from which source
    file, partner tensors from other shards included"""
    def deps_of(name, acc):
        for c in converters:
            for d in c.get_dependencies(name):
                if d not in acc:
                    acc.add(d)
                    deps_of(d, acc)
        return acc

    deps = {name: deps_of(name, set()) for name in weight_map}
    for name, d in deps.items():
        assert name not in d, f"{name} found in dependencies {d}"
    all_deps = set().union(*deps.values()) if deps else set()
    out = defaultdict(lambda: defaultdict(list))
    for name, shard in weight_map.items():
        if name in all_deps:
            continue  # partner of some primary tensor: loaded with it
        for n in (name, *deps[name]):
            if n not in weight_map:
                raise ValueError(f"Dependency weight {n} not found in weight map")
            out[shard][model_files[weight_map[n]]].append(n)
    return {k: dict(v) for k, v in out.items()}


def _args_from_dict(d: Optional[dict]) -> Optional[QuantizationArgs]:
    if d is None:
        return None
    known = {k: d[k] for k in ("num_bits", "type", "symmetric", "group_size", "strategy", "block_structure", "dynamic", "actorder") if k in d}
    if known.get("dynamic") not in (True, False):
        known["dynamic"] = bool(known.get("dynamic")) if known.get("dynamic") != "local" else False
    for key in ("scale_dtype", "zp_dtype"):  # serialised as str(dtype), e.g. "torch.uint8" (quant_args.py:209-224)
        v = d.get(key)
        if isinstance(v, str):
            v = getattr(torch, v.split(".")[-1], None)
        if isinstance(v, torch.dtype):
            known[key] = v
    return QuantizationArgs(**known)


_H2D_ALIGN = 256
_READY_BYTES = 32 << 20  # one event per ~32 MB of D2H copies


_STREAMING = threading.local()


@contextlib.contextmanager
def streaming_results():
    """inside this context (per thread) a converter that supports it hands its tensors over while their D2H copies are still in flight"""
    prev = getattr(_STREAMING, "on", False)
    _STREAMING.on = True
    try:
        yield
    finally:
        _STREAMING.on = prev


class ReadyDict(dict):
    """a shard's converted tensors: host tensors whose device-to-host copies may still be in flight.  `ready[name]` is the event
    recorded behind the copy of `name` (absent: the tensor is complete); `keep` holds what must stay alive until then.  Consumers
    that do not know about it call `wait()` first."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.ready = {}
        self.keep = []

    def wait(self, name=None) -> None:
        if name is not None:
            ev = self.ready.pop(name, None)
            if ev is not None:
                ev.synchronize()
            return
        for ev in set(self.ready.values()):
            ev.synchronize()
        self.ready.clear()
        self.keep.clear()


def _stage_to_device(state_dicts, dev, host_only=()):
    """Move a shard's compressed tensors to the device through ONE pinned buffer and ONE copy.  The tensors safetensors
    hands out are lazily mapped file pages: `t.to(device)` per tensor is a pageable copy that faults the file in 4 KB at a
    time on the calling thread (23.0 ms for the 125 MB of a TinyLlama-shaped shard, half of `process`).  Here the I/O
    threads copy the mapped pages into the pinned buffer in parallel, one asynchronous H2D moves it (6.7 ms together), and
    the device tensors are views into the device buffer (256-byte aligned).  Replaces the entries of `state_dicts` in place; returns what must stay alive until
    the stream is synchronised."""
    from .safetensors_io import host_bytes, parallel_copy

    plan, off = [], 0
    for sd in state_dicts:
        for key, t in sd.items():
            if key in host_only or t is None or t.device.type != "cpu":
                continue
            t = t.contiguous()
            n = t.numel() * t.element_size()
            plan.append((sd, key, t, off, n))
            off += (n + _H2D_ALIGN - 1) // _H2D_ALIGN * _H2D_ALIGN
    if not plan:
        return None
    stage = torch.empty(off, dtype=torch.uint8, pin_memory=True)
    flat = stage.numpy()
    parallel_copy([(flat[o:o + n], host_bytes(t)) for _, _, t, o, n in plan if n])
    dbuf = stage.to(dev, non_blocking=True)
    for sd, key, t, o, n in plan:
        sd[key] = dbuf[o:o + n].view(t.dtype).view(t.shape)
    return stage, dbuf


class CompressedTensorsDequantizer(Converter):
    """ct_dequantizer.py:21-171: dequantize a checkpoint in the compressed-tensors format to `dtype`"""

    def __init__(self, model_dir, ignore: Iterable[str] = (), dtype=torch.bfloat16, device=None):
        self.dtype = dtype
        # `process` may return while the D2H copies are still in flight (a ReadyDict whose events the writer waits on tensor by tensor):
        # per thread inside `with streaming_results():` (what convert_files' pipelined shard threads use), or for every call of this
        # converter with `stream_results = True`; off, `process` synchronises before it returns, as every other caller expects
        self.stream_results = False
        self.device = torch.device(device) if device is not None else None
        files = get_checkpoint_files(model_dir)
        cfg_path = files.get(CONFIG_NAME) or files.get("params.json")
        if cfg_path is None:
            raise ValueError("Could not find config.json file")
        data = get_quantization_config(cfg_path)
        if data is None:
            raise ValueError("Could not find quantization_config in config.json")
        if not isinstance(data.get("config_groups"), dict):
            raise ValueError("Model quantization config was found, but it does not match expected compressed-tensors quantization format")
        self.ignore = list(data.get("ignore") or []) + list(ignore)
        self.schemes: List[QuantizationScheme] = []
        for group in data["config_groups"].values():
            scheme = QuantizationScheme(targets=list(group.get("targets", [])), weights=_args_from_dict(group.get("weights")),
                                        input_activations=_args_from_dict(group.get("input_activations")),
                                        output_activations=_args_from_dict(group.get("output_activations")))
            # the format is re-inferred from the scheme, as upstream does (:59-63)
            scheme.format = CompressionFormat(infer_module_format(torch.nn.Linear, scheme)).value
            self.schemes.append(scheme)

    def _compressor(self, scheme):
        return BaseCompressor.get_value_from_registry(scheme.format)

    def process(self, tensors: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        from ... import _lib

        dev = self.device or _lib.require_device()
        out = ReadyDict()
        ready = out.ready  # tensor name -> the event behind its D2H copy (empty once everything has landed)
        keep = out.keep    # the pinned H2D staging buffers, alive until the stream has been synchronised / the shard has been written
        for scheme in self.schemes:
            comp = self._compressor(scheme)
            names = comp.compression_param_names(scheme)
            modules, state_dicts = [], []
            for module_name, _ in match_quantizable_tensors(tensors, self.ignore, scheme.targets, param_targets=[names[0]]):
                # weight_shape stays on the host (upstream keeps it a CPU int64 tensor)
                modules.append(module_name)
                state_dicts.append({p: tensors.pop(f"{module_name}.{p}") for p in names})
            if not modules:
                continue
            keep.append(_stage_to_device(state_dicts, dev, host_only=("weight_shape",)))
            results = comp.decompress_many(state_dicts, scheme)  # one launch for the eligible modules
            weights = [res["weight"].to(self.dtype) for res in results]
            # ONE pinned staging buffer per scheme and shard (a pinned allocation per tensor costs more than its copy)
            sizes = [(w.numel() * w.element_size() + 63) // 64 * 64 for w in weights]
            stage = torch.empty(sum(sizes), dtype=torch.uint8, pin_memory=True)
            # the D2H copies leave in the order the writer stores the tensors (sorted names) and an event follows every ~32 MB of them:
            # `write_safetensors` waits for a tensor's event, not for the whole shard, so the copies run under the file write (round 4)
            order = sorted(range(len(modules)), key=lambda i: modules[i])
            offs, off = [0] * len(modules), 0
            for i, n in enumerate(sizes):
                offs[i] = off
                off += n
            pending, since = [], 0
            for i in order:
                w = weights[i]
                host = stage[offs[i]:offs[i] + w.numel() * w.element_size()].view(w.dtype).view(w.shape)
                host.copy_(w, non_blocking=True)
                name = f"{modules[i]}.weight"
                out[name] = host
                pending.append(name)
                since += sizes[i]
                if since >= _READY_BYTES:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    ready.update(dict.fromkeys(pending, ev))
                    pending, since = [], 0
            if pending:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                ready.update(dict.fromkeys(pending, ev))
        if not (self.stream_results or getattr(_STREAMING, "on", False)):
            torch.cuda.current_stream(dev).synchronize()
            ready.clear()
            keep.clear()  # the list lives on in the returned ReadyDict: empty it, or the pinned H2D staging stays alive for the shard's lifetime
        # remaining (ignored / untargeted) tensors pass through, KV-cache qparams are dropped
        for name, t in tensors.items():
            if name.endswith(KV_CACHE_PARAM_NAMES):
                continue
            out[name] = t
        return out

    def validate(self, tensors) -> None:
        """only the NAMES are inspected: `tensors` may map names to None (:101-141)"""
        consumed, matched = set(), set()
        for scheme in self.schemes:
            names = self._compressor(scheme).compression_param_names(scheme)
            for module_name, _ in match_quantizable_tensors(tensors, self.ignore, scheme.targets, param_targets=[names[0]]):
                matched.add(module_name)
                for p in names:
                    key = f"{module_name}.{p}"
                    if key not in tensors:
                        raise ValueError(f"Expected key {key} not found")
                    consumed.add(key)
        left = [n for n in tensors if n not in consumed and n.rpartition(".")[0] in matched]
        if left:
            raise ValueError(f"Found {len(left)} unconsumed keys -- {left}")

    def create_config(self):
        return None

    def get_dependencies(self, weight_name: str) -> Set[str]:
        """:146-171: the first compression param is the root, the others are its partners"""
        module_name, _, param_name = weight_name.rpartition(".")
        if any(match_name(module_name, i) for i in self.ignore):
            return set()
        for scheme in self.schemes:
            names = self._compressor(scheme).compression_param_names(scheme)
            if "Linear" in scheme.targets or any(match_name(module_name, t) for t in scheme.targets):
                if param_name == names[0]:
                    return {f"{module_name}.{p}" for p in names[1:]}
                return set()
        return set()
