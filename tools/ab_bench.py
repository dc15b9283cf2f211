"""dev helper (run U): bench.py under one perturbation, to find what moves `tinyllama_checkpoint.api` between contexts.
python tools/ab_bench.py asis|pywait|lateimport|timed|timed_late|cpufirst|nogc|nocpu [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv.pop(1)
import bench

if mode == "pywait":  # the two waiting plug-in calls take their Python host path; the model path keeps the C++ loop
    from compressed_tensors_amd import _lib

    hp = _lib.hostpath()

    class Hidden:
        def __getattr__(self, name):
            if name in ("bitmask_compress", "marlin24_compress_default"):
                return lambda *a: None
            return getattr(hp, name)

    _lib._HOSTPATH[0] = Hidden()
elif mode == "lateimport":  # as pywait, and the extension itself is not imported before the model leg asks for it
    from compressed_tensors_amd import _lib

    class Lazy:
        def __getattr__(self, name):
            if name in ("bitmask_compress", "marlin24_compress_default"):
                return lambda *a: None
            _lib._HOSTPATH.clear()
            return getattr(_lib.hostpath(), name)

    _lib._HOSTPATH[:] = [Lazy()]
elif mode in ("timed", "timed_late"):  # where the host time of the model path goes, per function, summed over the whole run (stderr)
    import atexit, collections, json, time

    from compressed_tensors_amd import _lib, codec
    from compressed_tensors_amd.compressors import base as cbase
    from compressed_tensors_amd.compressors.model_compressors import model_compressor as mcm

    acc = collections.defaultdict(lambda: [0, 0.0])

    def timed(label, fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            e = acc[label]
            e[0] += 1
            e[1] += time.perf_counter() - t0
            return r
        return w

    class Timed:
        def __init__(self, late):
            self.hp = None if late else _lib.hostpath()

        def __getattr__(self, name):
            if self.hp is None:
                if name in ("bitmask_compress", "marlin24_compress_default"):
                    return lambda *a: None
                _lib._HOSTPATH.clear()
                self.hp = _lib.hostpath()
                _lib._HOSTPATH[0] = self
            fn = getattr(self.hp, name)
            return timed("hp." + name, fn) if callable(fn) else fn

    _lib._HOSTPATH[:] = [Timed(mode == "timed_late")]
    codec.launch_w4_words = timed("codec.launch_w4_words", codec.launch_w4_words)
    codec._upload_table = timed("codec._upload_table", codec._upload_table)
    cbase._by_format = timed("_by_format", cbase._by_format)
    for n in ("compress_model", "decompress_model", "_finish_compress", "remove_decompression_hook"):
        setattr(mcm.ModelCompressor, n, timed("ModelCompressor." + n, getattr(mcm.ModelCompressor, n)))
    atexit.register(lambda: print(json.dumps({k: [v[0], round(v[1] / v[0] * 1e6, 1)] for k, v in acc.items()}), file=sys.stderr))
elif mode == "cpufirst":  # the order of runs Q-W: extension imported, then the CPU baseline leg, then everything else
    import torch

    from compressed_tensors_amd import _lib

    _lib.hostpath()
    torch.cuda.set_device(0)
    bench.cpu_baseline(torch.device("cuda:0"))
    sys.argv.append("--no-cpu-baseline")
elif mode == "nogc":
    import gc

    gc.disable()
elif mode == "nocpu":
    sys.argv.append("--no-cpu-baseline")
bench.main()
