"""developer script: end-to-end rate of the model-free path (safetensors -> GPU decompress -> safetensors) on a
synthetic TinyLlama-1.1B-shaped W4A16 checkpoint kept in /dev/shm (no disk in the way): the PCIe / host-copy
inclusive figure that DESIGN.md quotes next to the HBM-resident kernel rates."""
import json, os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import save_file
import bench as B
import compressed_tensors_amd as cta
from compressed_tensors_amd import codec
from compressed_tensors_amd.entrypoints.convert import CompressedTensorsDequantizer, convert_checkpoint

dev = torch.device("cuda:0")
root = "/dev/shm/ct_convert_bench"
shutil.rmtree(root, ignore_errors=True)
src, dst = os.path.join(root, "src"), os.path.join(root, "dst")
os.makedirs(src)
args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
qcfg = {"quant_method": "compressed-tensors", "format": "pack-quantized", "ignore": [],
        "config_groups": {"group_0": {"targets": ["Linear"], "weights": {"num_bits": 4, "type": "int", "symmetric": True, "strategy": "group", "group_size": 128}}}}
json.dump({"quantization_config": qcfg}, open(os.path.join(src, "config.json"), "w"))
wm, in_bytes, out_bytes = {}, 0, 0
for layer0 in range(0, 22, 6):  # 4 shards
    tensors = {}
    for l in range(layer0, min(layer0 + 6, 22)):
        for name, r, c in B.TINYLLAMA_LAYER:
            w = torch.randn(r, c, dtype=torch.bfloat16, device=dev)
            s, z = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
            comp = cta.PackedQuantizationCompressor.compress({"weight": w, "weight_scale": s, "weight_zero_point": z}, scheme)
            for k, v in comp.items():
                tensors[f"model.layers.{l}.{name}.{k}"] = v.cpu().contiguous()
            out_bytes += r * c * 2
    fn = f"model-{layer0 // 6 + 1:05d}-of-00004.safetensors"
    save_file(tensors, os.path.join(src, fn))
    in_bytes += sum(t.numel() * t.element_size() for t in tensors.values())
    wm.update({k: fn for k in tensors})
json.dump({"metadata": {"total_size": in_bytes}, "weight_map": wm}, open(os.path.join(src, "model.safetensors.index.json"), "w"))
conv = CompressedTensorsDequantizer(src, dtype=torch.bfloat16, device=dev)
for workers in (1, 4):
    for rep in range(2):
        shutil.rmtree(dst, ignore_errors=True)
        t0 = time.perf_counter()
        convert_checkpoint(src, dst, conv, max_workers=workers)
        dt = time.perf_counter() - t0
    print(f"max_workers={workers}: {dt * 1e3:8.1f} ms  in {in_bytes / 1e6:.0f} MB  out {out_bytes / 1e6:.0f} MB  -> {(in_bytes + out_bytes) / dt / 1e9:.2f} GB/s end to end (host files in /dev/shm)")
# where the time of one shard goes
from compressed_tensors_amd.entrypoints.convert import converters as C
from compressed_tensors_amd.entrypoints.convert.converters import build_inverse_weight_maps
from compressed_tensors_amd.entrypoints.convert.safetensors_io import (get_checkpoint_files, get_weight_map, load_tensors_from_inverse_weight_map,
                                                                       write_safetensors)
files = get_checkpoint_files(src)
inv = build_inverse_weight_maps(get_weight_map(files), files, [conv])
shard = sorted(inv)[0]
os.makedirs(dst, exist_ok=True)
stage_ms = []
_orig_stage = C._stage_to_device
def _timed_stage(*a, **k):
    t = time.perf_counter(); r = _orig_stage(*a, **k); torch.cuda.synchronize(); stage_ms.append(1e3 * (time.perf_counter() - t)); return r
C._stage_to_device = _timed_stage
for rep in range(3):
    del stage_ms[:]
    t0 = time.perf_counter(); tensors = load_tensors_from_inverse_weight_map(inv[shard]); t1 = time.perf_counter()
    out = conv.process(tensors); t2 = time.perf_counter()
    write_safetensors(out, os.path.join(dst, shard)); t3 = time.perf_counter()
C._stage_to_device = _orig_stage
nb = sum(t.numel() * t.element_size() for t in out.values())
print(f"one shard ({nb / 1e6:.0f} MB out): load {1e3 * (t1 - t0):.1f} ms, process {1e3 * (t2 - t1):.1f} ms (of which mapped file -> pinned -> device "
      f"{sum(stage_ms):.1f} ms), write {1e3 * (t3 - t2):.1f} ms")
# the per-tensor pageable copy the staging replaced
for rep in range(2):
    tensors = load_tensors_from_inverse_weight_map(inv[shard])
    t0 = time.perf_counter()
    moved = [t.to(dev, non_blocking=True) for k, t in tensors.items() if not k.endswith("weight_shape")]
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"  the same shard's compressed tensors with one pageable .to(device) each: {1e3 * (t1 - t0):.1f} ms")
shutil.rmtree(root, ignore_errors=True)
