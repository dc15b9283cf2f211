"""ModelCompressor (reference compressors/model_compressors/model_compressor.py:36-273):
iterates the quantized modules of a model and compresses / decompresses each with the codec
named by its scheme's format.

Multi-GPU: when torch.distributed is initialised, modules are partitioned over ranks with the
reference's own rule (largest first onto the lightest bin, distributed/assign.py:12-42) and each
rank compresses its bin on its own MI355X.  Two modes:

* `recouple=True` (default, the reference's semantics): afterwards every rank holds the full compressed model
  (distributed/module_parallel.py:74-90), here as one flat-buffer RCCL broadcast per owner rank.
* `recouple=False` (BASELINE north_star: one weight shard per rank, no collective): every rank keeps ONLY its own
  share compressed — the mode for writing one checkpoint shard per rank.  The result is partial by design, so under
  torch.distributed with more than one rank the model-wide status is left alone and no decompress hook is attached;
  the owned modules are returned.

`decompress_model` decompresses every locally compressed module on every rank (as the reference, which has no
distributed decompression: model_compressor.py:196); `recouple=True` is the opt-in distributed form (each rank
decompresses its share, then the results are replicated) and requires the compressed model to be replicated.
"""
import json
import os
from typing import Optional

import torch

from ...config import CompressionFormat
from ...distributed import dense_numel, is_distributed, module_size, rank_and_world, replace_module_parallel
from ...quantization.quant_args import QuantizationStatus
from ...quantization.utils import is_module_quantized
from ..base import compress_modules, decompress_modules
from ..format import infer_model_format

__all__ = ["ModelCompressor"]

# reference base.py:5-12
QUANTIZATION_CONFIG_NAME = "quantization_config"
COMPRESSION_VERSION_NAME = "version"
QUANTIZATION_METHOD_NAME = "quant_method"
QUANTIZATION_METHOD = "compressed-tensors"
SPARSITY_CONFIG_NAME = "sparsity_config"
TRANSFORM_CONFIG_NAME = "transform_config"


class ModelCompressor:
    def __init__(self, quantization_config=None, transform_config=None, force_compression_format: Optional[str] = None):
        self.quantization_config = quantization_config
        self.transform_config = transform_config
        self.force_compression_format = (
            CompressionFormat(getattr(force_compression_format, "value", force_compression_format))
            if force_compression_format is not None
            else None
        )

    @classmethod
    def from_compression_config(cls, compression_config):
        """model_compressor.py:64-86 (HF quantizer entry point)"""
        q_config = getattr(compression_config, "quantization_config", None)
        if q_config is None and not hasattr(compression_config, "quantization_config"):
            raise ValueError(
                f"Support for compression config of type {type(compression_config)} is no longer supported."
            )
        return cls(quantization_config=q_config, transform_config=getattr(compression_config, "transform_config", None))

    @classmethod
    def from_pretrained_model(cls, model: torch.nn.Module, sparsity_config_or_format=None, quantization_format: Optional[str] = None):
        """model_compressor.py:88-122; the quantization config object itself is the caller's
        (pydantic, reference side) — here only the inferred format is recorded"""
        fmt = infer_model_format(model, quantization_format)
        compressor = cls(quantization_config=getattr(model, "quantization_config", None),
                         transform_config=getattr(model, TRANSFORM_CONFIG_NAME, None),
                         force_compression_format=quantization_format)
        compressor.inferred_format = fmt
        return compressor

    # ------------------------------------------------------------------ compress / decompress
    def _quantized_modules(self, model, skip_compressed=False):
        """every quantized module, in `named_modules(remove_duplicate=True)` order (model_compressor.py:152-164,191-195); the walk runs in
        the C++ host extension when it is built (300 modules: 0.3 ms of interpreter time per call, comparable to the kernels of a
        1B-parameter checkpoint)"""
        from ..pack_quantized.base import _hostpath

        hp = _hostpath()
        mods = hp.quantized_modules(model) if hp is not None else [m for _, m in self._named_quantized_modules(model)]
        return [m for m in mods if not self._is_compressed(m)] if skip_compressed else mods

    @staticmethod
    def _is_compressed(m) -> bool:
        return getattr(m, "quantization_status", None) == QuantizationStatus.COMPRESSED

    @staticmethod
    def _named_quantized_modules(model):
        """(name, module) of every quantized module, in `named_modules` order: the SAME list on every rank of a replicated model —
        what replace_module_parallel needs to agree on module identity (the ranks' skip_compressed filters may differ)"""
        return [(n, m) for n, m in model.named_modules(remove_duplicate=True) if is_module_quantized(m)]

    def _parallel(self, model, apply_many, recouple: bool, skip_compressed: bool):
        named = self._named_quantized_modules(model)
        done_fn = self._is_compressed if skip_compressed else None
        # collective-free mode with a skip filter: the LPT weights must not depend on what a rank has already compressed
        weight_fn = dense_numel if (skip_compressed and not recouple) else module_size
        return replace_module_parallel([m for _, m in named], apply_many, weight_fn, recouple=recouple, names=[n for n, _ in named], done_fn=done_fn)

    def _finish_compress(self, model, recouple: bool) -> None:
        if not recouple and is_distributed() and rank_and_world()[1] > 1:
            return  # shard-per-rank mode: this rank holds a partial result, the model as a whole is not "compressed"
        if self.quantization_config is not None and hasattr(self.quantization_config, "quantization_status"):
            self.quantization_config.quantization_status = QuantizationStatus.COMPRESSED
        self.add_decompress_hook(model)

    def compress_model(self, model: torch.nn.Module, skip_compressed: bool = False, recouple: bool = True):
        """model_compressor.py:138-181.  Under torch.distributed every rank compresses its own LPT share; with
        `recouple=True` (default, as upstream's replace_module_parallel) the results are then replicated on every
        rank at the price of one RCCL broadcast per owner rank; `recouple=False` is the collective-free
        shard-per-rank mode (see the module docstring).  Returns the modules this rank compressed."""
        fmt = self.force_compression_format
        # grouped by format: the pack-quantized codec turns its group into ONE kernel launch
        if not is_distributed():  # one rank: every module is this rank's (what replace_module_parallel does without a process group)
            # (handing the first 32 modules to the codec before the rest of the tree has been walked — an earlier first launch — measured
            # slower: 1.14 vs 1.10 ms for compress + decompress of a 154-module model; the host, not the first launch, is what the wall
            # clock follows, and the split costs it a second grouping and a fourth launch.  DESIGN.md 5.5)
            mine = self._quantized_modules(model, skip_compressed)
            compress_modules(mine, fmt)
        else:
            mine = self._parallel(model, lambda ms: compress_modules(ms, fmt), recouple, skip_compressed)
        self._finish_compress(model, recouple)
        return mine

    def compress_model_rtn(self, model: torch.nn.Module, recouple: bool = True):
        """Data-free (round-to-nearest) compression of a model whose modules carry a `quantization_scheme` but no scales yet:
        for every quantized module the min-max observer, calculate_qparams and the codec run fused — one pass over each
        weight where the scheme allows it (int4 group / channel, MXFP4, NVFP4, channel-wise int8 / float8; see the codecs'
        `compress_rtn`).  Bias and other parameters are kept.  Under torch.distributed the modules are sharded over the
        ranks exactly like `compress_model`.  No upstream counterpart: upstream separates calibration (observers,
        llm-compressor) from `compress_model`; the result equals that two-step flow with min-max observers."""
        from ...utils.module import direct_entry, swap_direct_entries
        from ..base import BaseCompressor
        from ..format import infer_module_format

        def apply(modules):
            # the format resolution and its write-back to the scheme happen once per (scheme object, module type), and the parameter dictionary is
            # rewritten as a delta (every `weight*` entry goes, the codec's entries come) — what get_direct_state_dict / replace_direct_state_dict did
            # per module at 28-45 us of host time beside one 2-30 us kernel (round 6)
            resolved = {}
            for module in modules:
                scheme = module.quantization_scheme
                key = (id(scheme), type(module))
                comp = resolved.get(key)
                if comp is None:
                    fmt = self.force_compression_format or getattr(scheme, "format", None) or infer_module_format(type(module), scheme)
                    fmt = CompressionFormat(getattr(fmt, "value", fmt))
                    comp = BaseCompressor.get_value_from_registry(fmt.value)
                    if not hasattr(comp, "compress_rtn"):
                        raise NotImplementedError(f"round-to-nearest compression is not implemented for format {fmt.value}")
                    try:
                        scheme.format = fmt
                    except Exception:
                        scheme.format = fmt.value
                    resolved[key] = comp
                weight = direct_entry(module, "weight")
                new = comp.compress_rtn(weight.data, scheme)
                remove = [k for k in (*module._parameters, *module._buffers) if k.startswith("weight")]
                swap_direct_entries(module, remove, new, status=QuantizationStatus.COMPRESSED)

        mine = self._parallel(model, apply, recouple, skip_compressed=True)
        self._finish_compress(model, recouple)
        return mine

    def decompress_model(self, model: torch.nn.Module, recouple: bool = False) -> None:
        """model_compressor.py:183-207.  Every rank decompresses every module it holds compressed (upstream has no
        distributed decompression, :196) — also the right thing after a shard-per-rank compress, where each rank
        holds a different subset.  `recouple=True` (opt-in, needs a replicated compressed model): each rank
        decompresses its LPT share only and the dense weights are then replicated by one broadcast per owner."""
        modules = self._quantized_modules(model)
        fmt = self.force_compression_format
        if recouple and is_distributed():
            def apply(ms):
                decompress_modules([m for m in ms if getattr(m, "quantization_status", None) == QuantizationStatus.COMPRESSED], fmt)

            named = self._named_quantized_modules(model)
            replace_module_parallel([m for _, m in named], apply, module_size, recouple=True, names=[n for n, _ in named])
        elif is_distributed():
            decompress_modules([m for m in modules if getattr(m, "quantization_status", None) == QuantizationStatus.COMPRESSED], fmt)
        else:
            decompress_modules(modules, fmt)
        if self.quantization_config is not None and hasattr(self.quantization_config, "quantization_status"):
            self.quantization_config.quantization_status = QuantizationStatus.DECOMPRESSED
        self.remove_decompression_hook(model)

    # ------------------------------------------------------------------ config.json
    def update_config(self, save_directory: str) -> None:
        """model_compressor.py:209-244"""
        if not any((self.quantization_config, self.transform_config)):
            return
        path = os.path.join(save_directory, "config.json")
        data = {}
        if os.path.exists(path):
            with open(path, "r") as f:
                data = json.load(f)

        def dump(cfg, **kw):
            if cfg is None:
                return {}
            if hasattr(cfg, "model_dump"):
                return cfg.model_dump(**kw)
            return dict(cfg) if isinstance(cfg, dict) else dict(vars(cfg))

        from ... import __version__

        data[QUANTIZATION_CONFIG_NAME] = {
            COMPRESSION_VERSION_NAME: __version__,
            QUANTIZATION_METHOD_NAME: QUANTIZATION_METHOD,
            SPARSITY_CONFIG_NAME: {},
            TRANSFORM_CONFIG_NAME: dump(self.transform_config),
            **{k: v for k, v in dump(self.quantization_config).items() if k != "quant_method"},
        }
        with open(path, "w") as f:
            json.dump(data, f, indent=2, sort_keys=True, default=lambda o: getattr(o, "value", str(o)))

    # ------------------------------------------------------------------ first-forward hook
    def add_decompress_hook(self, model: torch.nn.Module):
        """model_compressor.py:246-260"""

        def ct_decompress_hook(model, args):
            self.decompress_model(model)

        model.ct_decompress_hook = model.register_forward_pre_hook(ct_decompress_hook)

    def remove_decompression_hook(self, model: torch.nn.Module):
        """model_compressor.py:262-273"""
        if hasattr(model, "ct_decompress_hook"):
            model.ct_decompress_hook.remove()
            delattr(model, "ct_decompress_hook")
