"""Checkpoint-directory helpers for the model-free path (reference utils/safetensors_load.py:61-258,
470-521).  Local directories only: there is no hub access on the target machines."""
import json
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from safetensors import safe_open

__all__ = ["write_safetensors", "parallel_copy", "host_bytes", "CONFIG_NAME", "SAFE_WEIGHTS_NAME", "SAFE_WEIGHTS_INDEX_NAME", "QUANTIZATION_CONFIG_NAME", "is_weights_file",
           "get_checkpoint_files", "find_config_path", "get_quantization_config", "get_weight_map", "update_safetensors_index",
           "load_tensors_from_inverse_weight_map", "tensor_names_from_inverse_weight_map"]

CONFIG_NAME = "config.json"
SAFE_WEIGHTS_NAME = "model.safetensors"
SAFE_WEIGHTS_INDEX_NAME = "model.safetensors.index.json"
QUANTIZATION_CONFIG_NAME = "quantization_config"

InverseWeightMap = Dict[str, Optional[List[str]]]


def is_weights_file(file_name: str) -> bool:
    """safetensors_load.py:61-78"""
    return file_name.endswith((".bin", ".safetensors", ".pth", ".msgpack", ".pt"))


def get_checkpoint_files(model_dir) -> Dict[str, str]:
    """relative path -> absolute path of every file of a local checkpoint directory (:81-118)"""
    model_dir = os.fspath(model_dir)
    if not os.path.isdir(model_dir):
        raise ValueError(f"{model_dir} is not a local checkpoint directory (hub stubs are not resolvable here)")
    out = {}
    for dirpath, _, filenames in os.walk(model_dir):
        for fn in filenames:
            rel = os.path.relpath(os.path.join(dirpath, fn), model_dir)
            if rel.startswith((".cache", ".gitattributes")):
                continue
            out[rel] = os.path.join(model_dir, rel)
    return out


def find_config_path(directory) -> Optional[str]:
    """:134-150"""
    names = os.listdir(directory)
    for cand in (CONFIG_NAME, "params.json"):
        if cand in names:
            return os.path.join(directory, cand)
    return None


def get_quantization_config(config_path: str) -> Optional[dict]:
    """:153-180 (the cascade vLLM follows)"""
    with open(config_path) as f:
        config = json.load(f)
    if QUANTIZATION_CONFIG_NAME in config:
        return config[QUANTIZATION_CONFIG_NAME]
    if QUANTIZATION_CONFIG_NAME in config.get("text_config", {}):
        return config["text_config"][QUANTIZATION_CONFIG_NAME]
    return config.get("compression_config")


def get_weight_map(model_files: Dict[str, str]) -> Dict[str, str]:
    """tensor name -> shard file name, from the index or the single model.safetensors (:183-225)"""
    index = next((p for f, p in model_files.items() if f.endswith(SAFE_WEIGHTS_INDEX_NAME)), None)
    if index is None:
        index = next((p for f, p in model_files.items() if f.endswith(".safetensors.index.json")), None)
    if index is not None:
        with open(index) as f:
            return json.load(f)["weight_map"]
    if SAFE_WEIGHTS_NAME not in model_files:
        raise ValueError(f"File {SAFE_WEIGHTS_NAME} expected but not found in {list(model_files)}")
    with safe_open(model_files[SAFE_WEIGHTS_NAME], framework="pt") as f:
        return {name: SAFE_WEIGHTS_NAME for name in f.keys()}


def update_safetensors_index(save_directory, total_size: int, weight_map: Dict[str, str]) -> None:
    """:228-258"""
    path = next((os.path.join(save_directory, f) for f in os.listdir(save_directory) if f.endswith("safetensors.index.json")), None)
    if path is None:
        path = os.path.join(save_directory, SAFE_WEIGHTS_INDEX_NAME)
    with open(path, "w") as f:
        json.dump({"metadata": {"total_size": total_size}, "weight_map": weight_map}, f, indent=2, sort_keys=True)


def tensor_names_from_inverse_weight_map(inverse_weight_map: InverseWeightMap) -> Dict[str, None]:
    """names only (safetensors header reads): what a name-based validate() needs"""
    names = {}
    for source, wanted in inverse_weight_map.items():
        with safe_open(source, framework="pt") as f:
            keys = set(f.keys())
        for n in (wanted or keys):
            if n not in keys:
                raise ValueError(f"Expected to find tensor {n} in {source}, but tensor was not found.")
            names[n] = None
    return names


def load_tensors_from_inverse_weight_map(inverse_weight_map: InverseWeightMap, device="cpu") -> Dict[str, torch.Tensor]:
    """:478-521; `device` may be a GPU: safetensors then reads straight into device memory"""
    tensors = {}
    for source, wanted in inverse_weight_map.items():
        with safe_open(source, framework="pt", device=str(device)) as f:
            keys = set(f.keys())
            for n in (wanted or keys):
                if n not in keys:
                    raise ValueError(f"Expected to find tensor {n} in {source}, but tensor was not found.")
                tensors[n] = f.get_tensor(n)
    return tensors


_ST_DTYPE = {torch.float64: "F64", torch.float32: "F32", torch.float16: "F16", torch.bfloat16: "BF16", torch.int64: "I64",
             torch.int32: "I32", torch.int16: "I16", torch.int8: "I8", torch.uint8: "U8", torch.bool: "BOOL"}
for _n, _c in (("float8_e4m3fn", "F8_E4M3"), ("float8_e5m2", "F8_E5M2")):
    if hasattr(torch, _n):
        _ST_DTYPE[getattr(torch, _n)] = _c


_CHUNK = 4 << 20
_pool_lock = threading.Lock()
_pool: Optional[ThreadPoolExecutor] = None


def _io_pool() -> ThreadPoolExecutor:
    """one process-wide pool of copy threads (numpy's copy releases the GIL), shared by the converter's shard workers"""
    global _pool
    with _pool_lock:
        if _pool is None:
            n = int(os.environ.get("CT_CONVERT_IO_THREADS", 0)) or max(1, min(32, (os.cpu_count() or 4) // 4))
            _pool = ThreadPoolExecutor(n, thread_name_prefix="ct-io")
        return _pool


def host_bytes(t: torch.Tensor) -> np.ndarray:
    """the bytes of a contiguous host tensor as a flat uint8 array (no copy; bf16 / fp8 have no numpy dtype of their own)"""
    return t.reshape(-1).view(torch.uint8).numpy()


def parallel_copy(jobs: Sequence[Tuple[np.ndarray, np.ndarray]]) -> None:
    """copy every (destination, source) pair of flat uint8 arrays, in 4 MB pieces spread over the I/O threads.  A single
    thread moves ~5 GB/s when the source is a lazily mapped safetensors file (a page fault every 4 KB); the faults of
    different threads do not serialise."""
    pieces = []
    for dst, src in jobs:
        n = src.size
        if dst.size != n:
            raise ValueError("parallel_copy: size mismatch")
        for o in range(0, n, _CHUNK):
            pieces.append((dst[o:o + _CHUNK], src[o:o + _CHUNK]))
    if not pieces:
        return
    if len(pieces) == 1:
        np.copyto(*pieces[0])
        return
    for f in [_io_pool().submit(np.copyto, d, s_) for d, s_ in pieces]:
        f.result()


def write_safetensors(tensors: Dict[str, torch.Tensor], path) -> None:
    """Write a safetensors file straight from the tensors' host memory (no staging copy: the outputs of the
    converter sit in one pinned buffer and `safetensors.torch.save_file` would copy every tensor twice more;
    this path is bound by host copies, not by the GPU).  Layout: u64 header length, JSON header
    {name: {dtype, shape, data_offsets}} padded to 8 bytes, raw little-endian data.

    One `write()` per tensor on one thread.  Measured on the MI355X host (528 MB into tmpfs, `tools/convert_bench.py` round
    3): 68 ms this way; 169-184 ms with 4 / 16 threads of `pwrite` (buffered writes to ONE file serialise on the inode
    lock); 53-92 ms through a shared mapping filled by 4-64 copy threads after `posix_fallocate` (29 ms of that is the
    allocation itself, serial), 120-450 ms without the allocation.  Nothing beats the plain call by enough to carry it."""
    header, views, off = {}, [], 0
    for name in sorted(tensors):
        t = tensors[name]
        if t.device.type != "cpu":
            raise ValueError(f"{name}: write_safetensors expects host tensors")
        t = t.contiguous()
        if t.dtype not in _ST_DTYPE:
            raise ValueError(f"{name}: dtype {t.dtype} has no safetensors code")
        n = t.numel() * t.element_size()
        header[name] = {"dtype": _ST_DTYPE[t.dtype], "shape": list(t.shape), "data_offsets": [off, off + n]}
        views.append(t)
        off += n
    blob = json.dumps(header, separators=(",", ":")).encode()
    blob += b" " * ((8 - len(blob) % 8) % 8)
    wait = getattr(tensors, "wait", None)  # converters.ReadyDict: a tensor's D2H copy may still be in flight — wait for ITS event only
    names = sorted(tensors)
    with open(path, "wb") as f:
        f.write(len(blob).to_bytes(8, "little"))
        f.write(blob)
        for name, t in zip(names, views):
            if wait is not None:
                wait(name)
            if t.numel():
                f.write(memoryview(host_bytes(t)))
    if wait is not None:
        wait()
