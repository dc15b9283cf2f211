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
