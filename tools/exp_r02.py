"""round-2 experiments (dev helper): python tools/exp_r02.py w4d|bitmask   (knobs come from the environment)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
what = sys.argv[1]
if what == "w4d":
    out = {}
    for key, kw in (("bf16_sym", dict(n=B.N)), ("bf16_asym", dict(n=B.N, symmetric=False))):
        r = B.w4_kernel_point(dev, **kw)
        out[key] = (r["compress_us"], r["decompress_us"], r["round_trip_equals_fake_quantize"])
        torch.cuda.empty_cache()
    print(json.dumps({"rowlead": os.environ.get("CT_W4D_ROWLEAD"), **out}))
elif what == "bitmask":
    r = B.bitmask_leg(dev)
    print(json.dumps({"chunk_mb": os.environ.get("CT_BITMASK_CHUNK_MB"), "compress_us": r["compress_us"], "decompress_us": r["decompress_us"], "ok": r["round_trip_bit_exact"]}))
elif what == "rtn8":
    from compressed_tensors_amd import _lib
    lib = _lib.load(); stream = torch.cuda.current_stream(dev).cuda_stream
    N = B.N; nsets = 10
    g = torch.Generator(device=dev).manual_seed(3)
    ws = [torch.randn(N, N, dtype=torch.bfloat16, device=dev, generator=g) for _ in range(nsets)]
    q8 = [torch.empty(N, N, dtype=torch.uint8, device=dev) for _ in range(nsets)]
    sc = torch.empty(N, 1, dtype=torch.bfloat16, device=dev); zp = torch.empty(N, 1, dtype=torch.int8, device=dev)
    out = {}
    for name, fp8, sym in (("fp8", 1, 1), ("int8_sym", 0, 1), ("int8_asym", 0, 0)):
        f = lambda i: lib.ct_rtn_quant_channel8(ws[i % nsets].data_ptr(), _lib.BF16, N, N, fp8, sym, q8[i % nsets].data_ptr(), sc.data_ptr(), zp.data_ptr(), stream)
        out[name] = round(B.time_kernel(f, 40), 2)
    print(json.dumps({"wave": os.environ.get("CT_RTN8_WAVE"), **out}))
elif what == "qp":
    r = B.qparams_leg(dev)
    print(json.dumps({"U": os.environ.get("CT_QP_U"), "us": r["us"], "fused_us": r["fused_with_compress"]["us"]}))
