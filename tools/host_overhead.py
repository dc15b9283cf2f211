"""developer script: the HOST cost of one plug-in call — tiny tensors, so that the Python / ctypes path and not the kernel is
what is timed.  `python tools/host_overhead.py [profile]`.  The figure matters where a kernel is as short as the host glue
(marlin-24 at 8192^2: a 34 us kernel; every 4096^2 launch: 9 us)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import compressed_tensors_amd as cta
from compressed_tensors_amd import _lib, codec

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def per_call(fn, n=20000, reps=5):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
        torch.cuda.synchronize()
    return best


g = torch.Generator(device=dev).manual_seed(1)
w = torch.randn(64, 256, dtype=torch.bfloat16, device=dev, generator=g)
w24 = w * codec.sparse24_mask(w).to(w.dtype)
args = cta.QuantizationArgs(num_bits=4, group_size=128, symmetric=True, strategy="group")
scheme = cta.QuantizationScheme(targets=["Linear"], weights=args)
scale, zp = codec.minmax_qparams(w24, num_bits=4, group_size=128, symmetric=True)
sd24 = {"weight": w24, "weight_scale": scale, "weight_zero_point": zp}
s2, z2 = codec.minmax_qparams(w, num_bits=4, group_size=128, symmetric=True)
sd = {"weight": w, "weight_scale": s2, "weight_zero_point": z2}
M, P = cta.Marlin24Compressor, cta.PackedQuantizationCompressor
packed = P.compress(sd, scheme)
q8 = torch.randint(-8, 8, (64, 256), dtype=torch.int8, device=dev)
lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
out = {}
out["torch.empty (one device tensor)"] = per_call(lambda: torch.empty((8, 128), dtype=torch.int32, device=dev))
out["torch.cuda.current_stream(dev).cuda_stream"] = per_call(lambda: torch.cuda.current_stream(dev).cuda_stream)
out["_lib.stream_of(tensor)"] = per_call(lambda: _lib.stream_of(w))
out["torch.cuda.current_device()"] = per_call(lambda: torch.cuda.current_device())
out["tensor.data_ptr()"] = per_call(lambda: w.data_ptr())
flag = torch.zeros(1, dtype=torch.int32, device=dev)
bufs = (torch.empty(8, 128, dtype=torch.int32, device=dev), torch.empty(8, 128, dtype=torch.int16, device=dev), torch.empty(2, 64, dtype=torch.float16, device=dev))
a = (w24.data_ptr(), _lib.BF16, scale.data_ptr(), _lib.BF16, zp.data_ptr(), codec.DT[zp.dtype], 64, 256, 128, 1, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), flag.data_ptr(), 0, stream)
out["raw ctypes ct_marlin24_compress_w4_full (16 args, tiny launch)"] = per_call(lambda: lib.ct_marlin24_compress_w4_full(*a))
out["codec.pack_to_int32 (64x256 int8)"] = per_call(lambda: codec.pack_to_int32(q8, 4))
out["PackedQuantizationCompressor.compress"] = per_call(lambda: P.compress(sd, scheme))
out["PackedQuantizationCompressor.decompress"] = per_call(lambda: P.decompress(packed, scheme))
with M.deferred_structure_check():
    out["Marlin24Compressor.compress (deferred check)"] = per_call(lambda: M.compress(sd24, scheme), n=1000, reps=20)
out["Marlin24Compressor.compress (check per call)"] = per_call(lambda: M.compress(sd24, scheme), n=2000, reps=3)
for k, v in out.items():
    print(f"{v:7.2f} us  {k}")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, io, pstats
    for name, fn in (("marlin", lambda: M.compress(sd24, scheme)), ("packed.compress", lambda: P.compress(sd, scheme)), ("packed.decompress", lambda: P.decompress(packed, scheme))):
        pr = cProfile.Profile()
        with M.deferred_structure_check():
            pr.enable()
            for _ in range(1000):
                fn()
            pr.disable()
        torch.cuda.synchronize()
        st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(16)
        print("=====", name); print(st.getvalue()[:3800])
