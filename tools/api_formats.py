"""ModelCompressor.compress_model + decompress_model wall time for FP8 / NVFP4 / MXFP4 / W4 schemes on TinyLlama- and Llama-3-8B-shaped trees:
what the plug-in API costs per module for the formats that have no table launch"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import compressed_tensors_amd as cta
from compressed_tensors_amd import codec
dev = torch.device("cuda:0")
if os.environ.get("NATIVE", "1") == "0":  # the interpreter's module loops alone (what every format ran before its C++ loop existed)
    from compressed_tensors_amd import _lib
    _lib.hostpath()
    _lib._HOSTPATH[:] = [None]
TINY = (("q_proj", 2048, 2048), ("k_proj", 256, 2048), ("v_proj", 256, 2048), ("o_proj", 2048, 2048), ("gate_proj", 5632, 2048), ("up_proj", 5632, 2048), ("down_proj", 2048, 5632))
L8B = (("q_proj", 4096, 4096), ("k_proj", 1024, 4096), ("v_proj", 1024, 4096), ("o_proj", 4096, 4096), ("gate_proj", 14336, 4096), ("up_proj", 14336, 4096), ("down_proj", 4096, 14336))
F8 = torch.float8_e4m3fn

def build(layer_shapes, nlayers, fmt):
    root = torch.nn.Module(); root.layers = torch.nn.ModuleList()
    g = torch.Generator(device=dev).manual_seed(1)
    if fmt == "fp8":
        args = cta.QuantizationArgs(num_bits=8, type="float", strategy="channel", symmetric=True)
    elif fmt == "fp8blk":  # block 128 x 128 (the FP8-block checkpoints' layout)
        args = cta.QuantizationArgs(num_bits=8, type="float", strategy="block", block_structure=[128, 128], symmetric=True)
    elif fmt == "w4asym":
        args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=False, strategy="group")
    elif fmt == "mxfp8":
        args = cta.QuantizationArgs(num_bits=8, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=torch.uint8)
    elif fmt == "w4act":  # activation ordering (GPTQ actorder="group"): the modules carry weight_g_idx
        args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group", actorder="group")
    elif fmt == "w8a16":  # the W8A16 preset: weight-only int8, channel-wise, stored pack-quantized
        args = cta.QuantizationArgs(num_bits=8, symmetric=True, strategy="channel")
    elif fmt == "w3":
        args = cta.QuantizationArgs(num_bits=3, group_size=128, symmetric=True, strategy="group")
    elif fmt == "nvfp4":
        args = cta.QuantizationArgs(num_bits=4, type="float", strategy="tensor_group", symmetric=True, group_size=16, scale_dtype=F8)
    elif fmt == "mxfp4":
        args = cta.QuantizationArgs(num_bits=4, type="float", strategy="group", symmetric=True, group_size=32, scale_dtype=torch.uint8)
    else:
        args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
    scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
    if fmt in ("nvfp4", "mxfp4"):
        scheme.format = fmt + "-pack-quantized"
    if fmt == "mxfp8":
        scheme.format = "mxfp8-quantized"
    alg = 0
    for l in range(nlayers):
        blk = torch.nn.Module(); root.layers.append(blk)
        for (proj, r, c) in layer_shapes:
            w = torch.randn(r, c, dtype=torch.bfloat16, device=dev, generator=g)
            lin = torch.nn.Linear(c, r, bias=False, device="meta")
            lin.weight = torch.nn.Parameter(w, requires_grad=False)
            if fmt == "fp8":
                s = codec.minmax_qparams_float(w, kind="fp8"); z = torch.zeros(s.shape, dtype=F8, device=dev)
                alg += 2 * (3 * r * c)
            elif fmt == "fp8blk":
                s = (w.float().reshape(r // 128, 128, c // 128, 128).abs().amax(dim=(1, 3)) / 448.0).to(torch.bfloat16); z = torch.zeros(s.shape, dtype=F8, device=dev)
                alg += 2 * (3 * r * c)
            elif fmt == "w4asym":
                s, z = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=False)
                alg += 2 * int(2.5 * r * c)
            elif fmt == "mxfp8":
                amax = w.float().reshape(r, -1, 32).abs().amax(-1).clamp(min=1e-4)
                s = torch.exp2(torch.floor(torch.log2(amax)) - 8).to(torch.bfloat16); z = torch.zeros(s.shape, dtype=F8, device=dev)
                alg += 2 * int((3 + 3 / 32) * r * c)
            elif fmt == "w4act":
                s, z = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
                lin.weight_g_idx = torch.nn.Parameter((torch.randperm(c, device=dev, generator=g) // 128).to(torch.int32), requires_grad=False)
                alg += 2 * int(2.5 * r * c)
            elif fmt == "w8a16":
                s, z = codec.minmax_qparams(w, num_bits=8, group_size=None, symmetric=True)
                alg += 2 * (3 * r * c)
            elif fmt == "w3":
                s, z = codec.minmax_qparams(w, num_bits=3, group_size=128, symmetric=True)
                alg += 2 * int((2 + 3 / 8) * r * c)
            elif fmt == "nvfp4":
                gs = codec.generate_gparam(w); s = codec.minmax_qparams_float(w, kind="nvfp4", group_size=16, global_scale=gs); z = None
                lin.weight_global_scale = torch.nn.Parameter(gs, requires_grad=False)
                alg += 2 * int(2.5 * r * c)
            elif fmt == "mxfp4":
                s = codec.minmax_qparams_float(w, kind="mxfp4", group_size=32); z = None
                alg += 2 * int(2.5 * r * c)
            else:
                s, z = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
                alg += 2 * int(2.5 * r * c)
            lin.weight_scale = torch.nn.Parameter(s, requires_grad=False)
            if z is not None: lin.weight_zero_point = torch.nn.Parameter(z, requires_grad=False)
            lin.quantization_scheme = scheme
            setattr(blk, proj, lin)
    return root, alg

for name, shapes, nl in (("tinyllama 154 modules", TINY, 22), ("llama-8B-shaped 112 modules", L8B, 16)):
    for fmt in os.environ.get("FORMATS", "w4,w4asym,w8a16,fp8,fp8blk,mxfp8,nvfp4,mxfp4").split(","):
        model, alg = build(shapes, nl, fmt)
        mc = cta.ModelCompressor()
        def cycle():
            mc.compress_model(model); mc.decompress_model(model)
        try:
            cycle(); cycle()
        except Exception as e:
            print(name, fmt, "FAILED", repr(e)[:300]); continue
        both, host = [], []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); cycle(); t1 = time.perf_counter(); torch.cuda.synchronize()
            both.append(time.perf_counter() - t0); host.append(t1 - t0)
        both.sort(); host.sort()
        n = nl * 7
        print(f"{name:30s} {fmt:6s}: wall {both[2]*1e3:8.3f} ms ({alg/both[2]/8e12:.3f} of HBM peak), host returns at {host[2]*1e3:8.3f} ms = {host[2]*1e6/n/2:.1f} us per module and direction", flush=True)
        del model
        torch.cuda.empty_cache()
