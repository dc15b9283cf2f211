"""dev helper (run U): bench.py under one perturbation, to find what moves `tinyllama_checkpoint.api` between contexts.
python tools/ab_bench.py asis|pywait|nogc|nocpu [bench args]"""
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
elif mode == "nogc":
    import gc

    gc.disable()
elif mode == "nocpu":
    sys.argv.append("--no-cpu-baseline")
bench.main()
