"""convert_checkpoint (reference entrypoints/convert/convert_checkpoint.py:32-134, convert_file.py:29-121).

Shards are independent units: with torch.distributed initialised the safetensors files are split over
the ranks with the same LPT rule as the modules (largest file first onto the lightest rank), every rank
converts its files on its own GPU, and rank 0 merges the per-rank index fragments from the file system —
no tensor ever crosses ranks.  Inside a rank, `max_workers` threads each drive their own HIP stream and hand
their finished shards to `max_workers` writer threads (`convert_files`), so the H2D copy / decompress / D2H of one file
overlaps the output write of another (the end-to-end bound of this path is PCIe and storage, not HBM)."""
import json
import os
import shutil
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import torch

from ...distributed.shard import is_distributed, rank_and_world, shard_items
from .converters import Converter, build_inverse_weight_maps
from .safetensors_io import (QUANTIZATION_CONFIG_NAME, find_config_path, get_checkpoint_files, get_weight_map, is_weights_file,
                             load_tensors_from_inverse_weight_map, tensor_names_from_inverse_weight_map, update_safetensors_index,
                             write_safetensors)

__all__ = ["convert_checkpoint", "convert_file", "convert_files", "validate_file", "exec_jobs", "write_checkpoint_quantization_config"]


def write_checkpoint_quantization_config(save_directory, converter: Converter) -> None:
    """convert_file.py:29-77"""
    from ... import __version__

    new_cfg = converter.create_config()
    data = None
    if new_cfg is not None:
        data = new_cfg.model_dump() if hasattr(new_cfg, "model_dump") else dict(new_cfg)
        data["version"] = __version__
    path = find_config_path(save_directory)
    if path is None:
        return
    with open(path) as f:
        cfg = json.load(f)
    if data is None:
        if QUANTIZATION_CONFIG_NAME in cfg:
            del cfg[QUANTIZATION_CONFIG_NAME]
        elif QUANTIZATION_CONFIG_NAME in cfg.get("text_config", {}):
            del cfg["text_config"][QUANTIZATION_CONFIG_NAME]
    else:
        cfg[QUANTIZATION_CONFIG_NAME] = data
    with open(path, "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)


def validate_file(inverse_weight_map, converter: Converter) -> None:
    """convert_file.py:80-95, from the safetensors headers only (no tensor data is read)"""
    converter.validate(tensor_names_from_inverse_weight_map(inverse_weight_map))


def convert_file(inverse_weight_map, save_path, converter: Converter):
    """convert_file.py:98-121 -> (bytes written, {tensor name: file name})"""
    return _write_file(_process_file(inverse_weight_map, converter), save_path)


def _process_file(inverse_weight_map, converter: Converter, stream_results: bool = False):
    """`stream_results` (the pipelined `convert_files` only): the converter may hand its tensors over while their D2H copies are still in
    flight — CompressedTensorsDequantizer then returns a ReadyDict whose events `write_safetensors` waits on tensor by tensor"""
    from .converters import streaming_results

    def run():
        tensors = load_tensors_from_inverse_weight_map(inverse_weight_map)
        if stream_results:
            with streaming_results():
                return converter.process(tensors)
        return converter.process(tensors)

    if torch.cuda.is_available():
        with torch.cuda.stream(torch.cuda.Stream()):  # this thread's own stream
            return run()
    return run()


def _write_file(tensors, save_path):
    Path(save_path).parent.mkdir(parents=True, exist_ok=True)
    write_safetensors(tensors, str(save_path))
    total = sum(t.numel() * t.element_size() for t in tensors.values())
    return total, {k: os.path.basename(save_path) for k in tensors}


def convert_files(items, converter: Converter, max_workers: int = 1):
    """`convert_file` over [(inverse_weight_map, save_path)], as a two-stage pipeline: `max_workers` threads read + convert (each on its
    own HIP stream) and hand the finished shard — tensors in pinned host memory — to `max_workers` writer threads, so that the output
    write of one shard (four fifths of a shard's time: the kernel and the copies are ~25 ms of ~130 ms) runs under the
    next shard's read, H2D copy, kernel and D2H copy instead of in front of them.  At most 2 x max_workers converted shards exist at
    any time (the bound on pinned memory).  Results in input order."""
    items = list(items)
    if max_workers <= 1 or len(items) <= 1:
        return [convert_file(inv, path, converter) for inv, path in items]
    import threading

    slots = threading.BoundedSemaphore(2 * max_workers)
    with ThreadPoolExecutor(max_workers, thread_name_prefix="ct-convert") as convert, ThreadPoolExecutor(max_workers, thread_name_prefix="ct-write") as write:
        def finish(tensors, path):
            try:
                return _write_file(tensors, path)
            finally:
                slots.release()

        def start(inv, path):
            slots.acquire()
            try:
                tensors = _process_file(inv, converter, stream_results=True)
            except BaseException:
                slots.release()
                raise
            return write.submit(finish, tensors, path)

        started = [convert.submit(start, inv, path) for inv, path in items]
        return [f.result().result() for f in started]


def exec_jobs(jobs, max_workers: int = 1, desc: str = ""):
    """convert_checkpoint.py:110-134"""
    if max_workers == 1:
        return [job[0](*job[1:]) for job in jobs]
    with ThreadPoolExecutor(max_workers) as ex:
        return [f.result() for f in [ex.submit(*job) for job in jobs]]


def convert_checkpoint(model_dir, save_directory, converter: Converter, max_workers: int = 1) -> None:
    """Convert a checkpoint file by file without instantiating the model (convert_checkpoint.py:32-107)."""
    model_files = get_checkpoint_files(model_dir)
    weight_map = get_weight_map(model_files)
    inverse = build_inverse_weight_maps(weight_map, model_files, [converter])
    rank, world = rank_and_world()
    save_directory = Path(save_directory)
    save_directory.mkdir(parents=True, exist_ok=True)

    shards = [s for s in model_files if s.endswith("safetensors")]
    for s in shards:
        if s not in inverse:
            raise ValueError(f"Could not find inverse_weight_map for shard {s}")
    mine = shard_items(shards, weight_fn=lambda s: os.path.getsize(model_files[s]), rank=rank, world_size=world)
    if rank == 0:
        for name, path in model_files.items():
            if name.endswith("safetensors"):
                continue
            dst = save_directory / name
            if str(path) != str(dst) and not name.endswith("safetensors.index.json"):
                dst.parent.mkdir(parents=True, exist_ok=True)
                shutil.copyfile(path, dst)

    exec_jobs([(validate_file, inverse[s], converter) for s in mine], max_workers, "Validating")
    results = convert_files([(inverse[s], save_directory / s) for s in mine], converter, max_workers)
    total, new_map = 0, {}
    for t, m in results:
        total += t
        new_map.update(m)

    if is_distributed():
        import torch.distributed as dist

        with open(save_directory / f".index_fragment_rank{rank}.json", "w") as f:
            json.dump({"total_size": total, "weight_map": new_map}, f)
        dist.barrier()  # control only: the fragments must exist before rank 0 merges them
        if rank == 0:
            total, new_map = 0, {}
            for r in range(world):
                frag = save_directory / f".index_fragment_rank{r}.json"
                with open(frag) as f:
                    d = json.load(f)
                total += d["total_size"]
                new_map.update(d["weight_map"])
                frag.unlink()
    if rank == 0:
        write_checkpoint_quantization_config(save_directory, converter)
        update_safetensors_index(save_directory, total, new_map)
    if is_distributed():
        import torch.distributed as dist

        dist.barrier()
