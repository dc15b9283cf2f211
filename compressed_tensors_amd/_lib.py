"""ctypes binding of libct_hip.so (the C ABI declared in include/ct_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950) next to
this file.  There is NO fallback: if the shared object is missing, or no MI355X is visible
when a compute entry point is called, the call raises.  Nothing in this package computes the
hot path on the CPU.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libct_hip.so")
# the diagnostics build of the same library (ct_sparse.hip compiled -DCT_DIAG: the CT_BITMASK_RESIDENT* knobs and the per-workgroup
# time stamps).  Never loaded by the package: a test or a dev tool that wants the knobs sets `_lib.LIB_PATH = _lib.DIAG_LIB_PATH`
# in its own process before the first load().
DIAG_LIB_PATH = os.path.join(_HERE, "libct_hip_diag.so")

# element type codes of include/ct_hip.h
F32, F16, BF16, I8, I32, U8, I16, I64, F8 = range(9)
DT = {
    torch.float32: F32,
    torch.float16: F16,
    torch.bfloat16: BF16,
    torch.int8: I8,
    torch.int32: I32,
    torch.uint8: U8,
    torch.bool: U8,
    torch.int16: I16,
    torch.int64: I64,
}
if hasattr(torch, "float8_e4m3fn"):
    DT[torch.float8_e4m3fn] = F8  # FLOAT 8-bit quantization type (the copy codecs view fp8 payloads as raw bytes first)
if hasattr(torch, "float8_e5m2"):
    DT[torch.float8_e5m2] = I8  # raw bytes only

CT_OK, CT_ERR_INVALID_ARG, CT_ERR_UNSUPPORTED, CT_ERR_HIP = range(4)

_c = ctypes
_P, _I, _L, _S = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_void_p
_Q = [_P, _I, _P, _I, _P, _I, _L, _L, _L, _L, _L, _P]  # x,xdt,scale,sdt,zp,zdt,rows,cols,rdiv,cdiv,scale_cols,col_group

_PROTOTYPES = {
    "ct_abi_version": ([], _I),
    "ct_last_error": ([], _c.c_char_p),
    "ct_mailbox_alloc": ([_L, _P, _P], _I),
    "ct_mailbox_free": ([_P], _I),
    "ct_mailbox_wait_i64": ([_P, _L, _S, _P], _I),
    "ct_stream_wait": ([_S], _I),
    "ct_pack_int32": ([_P, _L, _L, _I, _P, _L, _S], _I),
    "ct_unpack_int32": ([_P, _L, _L, _L, _L, _I, _P, _S], _I),
    "ct_pack_int32_dim0": ([_P, _L, _L, _I, _P, _S], _I),
    "ct_unpack_int32_dim0": ([_P, _L, _L, _L, _I, _P, _S], _I),
    "ct_quantize": (_Q + [_I, _I, _P, _I, _S], _I),
    "ct_dequantize": (_Q + [_P, _I, _S], _I),
    "ct_fake_quantize": (_Q + [_I, _I, _P, _I, _S], _I),
    "ct_quantize_fp8": (_Q + [_I, _P, _I, _S], _I),
    "ct_fake_quantize_fp8": (_Q + [_I, _P, _I, _S], _I),
    "ct_quantize_fp4": (_Q + [_P, _I, _P, _I, _S], _I),
    "ct_fake_quantize_fp4": (_Q + [_P, _I, _P, _I, _S], _I),
    "ct_dequantize_gs": (_Q + [_P, _P, _I, _S], _I),
    "ct_quant_pack": (_Q + [_I, _I, _P, _S], _I),
    "ct_rtn_quant_channel8": ([_P, _I, _L, _L, _I, _I, _P, _P, _P, _S], _I),
    "ct_rtn_quant_pack_w4": ([_P, _I, _L, _L, _L, _I, _P, _P, _P, _S], _I),
    "ct_unpack_dequant": ([_P, _L, _L, _L, _I, _P, _I, _P, _I, _L, _L, _L, _P, _P, _I, _S], _I),
    "ct_quant_pack_w4_zp": ([_P, _I, _P, _P, _L, _L, _L, _P, _P, _S], _I),
    "ct_unpack_dequant_w4_zp": ([_P, _P, _I, _P, _L, _L, _L, _P, _P, _S], _I),
    "ct_gidx_col_group": ([_P, _L, _L, _P, _P, _S], _I),
    "ct_w4_batch_plan": ([_P, _I, _I], _L),
    "ct_quant_pack_batch": ([_P, _I, _L, _I, _S], _I),
    "ct_unpack_dequant_batch": ([_P, _I, _L, _I, _S], _I),
    "ct_zp4_batch_plan": ([_P, _I], _L),
    "ct_zp4_pack_dim0_batch": ([_P, _I, _L, _I, _S], _I),
    "ct_q8_batch_plan": ([_P, _I, _I], _L),
    "ct_q8_quant_batch": ([_P, _I, _L, _I, _I, _I, _S], _I),
    "ct_q8_dequant_batch": ([_P, _I, _L, _I, _I, _S], _I),
    "ct_fp4_quant_pack": ([_P, _I, _P, _I, _P, _L, _L, _L, _P, _S], _I),
    "ct_fp4_unpack_dequant": ([_P, _L, _L, _P, _I, _I, _P, _L, _P, _I, _S], _I),
    "ct_fp4_quant_pack_stored": ([_P, _I, _P, _I, _P, _L, _L, _L, _P, _P, _P, _S], _I),
    "ct_fp4_unpack_dequant_scale": ([_P, _L, _L, _P, _I, _I, _P, _L, _P, _I, _P, _S], _I),
    "ct_fp4_batch_plan": ([_P, _I, _I], _L),
    "ct_fp4_quant_pack_batch": ([_P, _I, _L, _I, _I, _I, _P, _S], _I),
    "ct_fp4_unpack_dequant_batch": ([_P, _I, _L, _I, _I, _S], _I),
    "ct_mx_scale_compress": ([_P, _I, _L, _P, _P, _S], _I),
    "ct_mx_scale_batch_plan": ([_P, _I], _L),
    "ct_mx_scale_batch": ([_P, _I, _L, _I, _I, _P, _S], _I),
    "ct_mx_scale_decompress": ([_P, _L, _P, _S], _I),
    "ct_rtn_mxfp4_quant_pack": ([_P, _I, _L, _L, _P, _P, _P, _S], _I),
    "ct_rtn_nvfp4_quant_pack": ([_P, _I, _L, _L, _P, _P, _P, _P, _S], _I),
    "ct_fp4_cast": ([_P, _I, _P, _L, _S], _I),
    "ct_fp4_pack": ([_P, _I, _P, _L, _S], _I),
    "ct_fp4_unpack": ([_P, _L, _P, _I, _S], _I),
    "ct_selftest_fp4_div": ([_I, _c.c_uint32, _c.c_uint32, _P, _S], _I),
    "ct_minmax_qparams": ([_P, _I, _L, _L, _L, _I, _I, _P, _P, _S], _I),
    "ct_minmax_qparams_float": ([_P, _I, _L, _L, _L, _I, _P, _P, _S], _I),
    "ct_generate_gparam": ([_P, _I, _L, _L, _P, _P, _S], _I),
    "ct_pack_bitmasks": ([_P, _L, _L, _P, _S], _I),
    "ct_unpack_bitmasks": ([_P, _L, _L, _P, _S], _I),
    "ct_bitmask_count": ([_P, _I, _L, _L, _P, _P, _S], _I),
    "ct_exclusive_scan_i64": ([_P, _L, _P, _P, _S], _I),
    "ct_bitmask_scatter": ([_P, _I, _L, _L, _P, _P, _S], _I),
    "ct_bitmask_decompress": ([_P, _L, _P, _P, _L, _I, _L, _L, _P, _S], _I),
    "ct_bitmask_compress_workspace_bytes": ([_L, _L], _L),
    "ct_bitmask_compress": ([_P, _I, _L, _L, _P, _L, _P, _P, _P, _P, _L, _S], _I),
    "ct_bitmask_batch_plan": ([_P, _I, _P], _L),
    "ct_bitmask_compress_batch": ([_P, _I, _L, _I, _P, _L, _S], _I),
    "ct_bitmask_decompress_batch_plan": ([_P, _I], _L),
    "ct_bitmask_decompress_batch": ([_P, _I, _L, _I, _S], _I),
    "ct_copy_batch_plan": ([_P, _I], _L),
    "ct_copy_batch": ([_P, _I, _L, _S], _I),
    "ct_bitmask_row_popcount": ([_P, _L, _L, _P, _S], _I),
    "ct_sparse24_compress": ([_P, _I, _L, _L, _P, _P, _S], _I),
    "ct_sparse24_mask": ([_P, _I, _L, _P, _S], _I),
    "ct_cutlass24_from_dense": ([_P, _I, _L, _L, _P, _P, _S], _I),
    "ct_cutlass24_to_dense": ([_P, _I, _P, _I, _L, _L, _P, _S], _I),
    "ct_marlin24_quant_compress": ([_P, _I, _P, _I, _P, _I, _L, _L, _L, _I, _P, _P, _P, _S], _I),
    "ct_marlin24_compress_w4": ([_P, _I, _P, _I, _P, _I, _L, _L, _L, _P, _P, _P, _S], _I),
    "ct_marlin24_compress_w4_full": ([_P, _I, _P, _I, _P, _I, _L, _L, _L, _I, _P, _P, _P, _P, _I, _S], _I),
    "ct_marlin24_compress_w4_verdict": ([_P, _I, _P, _I, _P, _I, _L, _L, _L, _I, _P, _P, _P, _P, _P, _I, _S], _I),
    "ct_selftest_m24_div": ([_I, _c.c_uint32, _c.c_uint32, _P, _S], _I),
    "ct_marlin24_pack_weights": ([_P, _I, _I, _I, _L, _L, _I, _P, _S], _I),
    "ct_marlin24_pack_scales": ([_P, _I, _L, _L, _I, _P, _S], _I),
    "ct_marlin24_pack_scales_f16": ([_P, _I, _L, _L, _I, _P, _S], _I),
    "ct_selftest_bf16_div": ([_c.c_uint32, _c.c_uint32, _P, _S], _I),
    "ct_selftest_f16_div": ([_c.c_uint32, _c.c_uint32, _P, _S], _I),
}



class W4Item(ctypes.Structure):
    """struct ct_w4_item of include/ct_hip.h"""
    _fields_ = [("src", _P), ("scale", _P), ("zp", _P), ("dst", _P), ("rows", _L), ("cols", _L), ("group", _L),
                ("first_block", _L), ("units", _L), ("upg_shift", _c.c_int32), ("upg", _c.c_int32),
                ("zp_packed", _P), ("main_blocks", _L), ("g_magic", _c.c_uint32), ("g_shift", _c.c_int32)]


class BitmaskItem(ctypes.Structure):
    """struct ct_bitmask_item of include/ct_hip.h (a row of ct_bitmask_compress_batch's table)"""
    _fields_ = [("x", _P), ("values", _P), ("bitmask", _P), ("row_offsets", _P), ("total", _P), ("rows", _L), ("cols", _L), ("values_capacity", _L),
                ("dt", _c.c_int32), ("is_float", _c.c_int32), ("first_block", _L), ("units", _L), ("upr", _L), ("slots_offset", _L),
                ("nwg", _c.c_int32), ("tpw", _c.c_int32), ("mask_dwords", _c.c_int32), ("gen", _c.c_uint32)]


class BitmaskDItem(ctypes.Structure):
    """struct ct_bitmask_ditem of include/ct_hip.h (a row of ct_bitmask_decompress_batch's table)"""
    _fields_ = [("values", _P), ("bitmask", _P), ("row_offsets", _P), ("out", _P), ("rows", _L), ("cols", _L), ("values_len", _L),
                ("dt", _c.c_int32), ("single", _c.c_int32), ("first_block", _L)]


class CopyItem(ctypes.Structure):
    """struct ct_copy_item of include/ct_hip.h"""
    _fields_ = [("src", _P), ("dst", _P), ("bytes", _L), ("first_block", _L)]


ITEM_WORDS = ctypes.sizeof(W4Item) // 8  # 13: every host table of ct_w4_item rows is a flat array of this many 64-bit words per item


EXPORTED_SYMBOLS = tuple(sorted(_PROTOTYPES))

_lib = None
_lock = threading.Lock()


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Load libct_hip.so (once).  Raises HipExtensionMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise HipExtensionMissing(
                    f"{LIB_PATH} not found: build the HIP extension first "
                    "(python -c 'import __graft_entry__ as g; g.build()').  "
                    "compressed_tensors_amd has no CPU fallback."
                )
            lib = ctypes.CDLL(LIB_PATH)
            for name, (argtypes, restype) in _PROTOTYPES.items():
                fn = getattr(lib, name)  # AttributeError here == ABI mismatch with include/ct_hip.h
                fn.argtypes = argtypes
                fn.restype = restype
            _lib = lib
    return _lib


def last_error() -> str:
    msg = load().ct_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int):
    """Map a ct_status onto the exception types the reference raises."""
    if status == CT_OK:
        return
    msg = last_error()
    if status == CT_ERR_INVALID_ARG:
        raise ValueError(msg)
    if status == CT_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


class StreamHandle(int):
    """a raw hipStream_t plus the index of the device it belongs to.  The handle of PyTorch's default stream is 0 — the
    null stream — which HIP resolves against the CURRENT device, not the tensors' device: `call` switches the current
    device for the launch when they differ (e.g. an HF `device_map` model spread over several GPUs in one process)."""
    device_index = None


def call(name: str, *args):
    """launch one C-ABI entry; the trailing argument is the stream (`stream_of(tensor)`), whose device becomes the
    current device for the duration of the call"""
    fn = getattr(load(), name)
    dev = getattr(args[-1], "device_index", None) if args else None
    if dev is not None and dev != torch.cuda.current_device():
        with torch.cuda.device(dev):
            check(fn(*args))
    else:
        check(fn(*args))


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor):
    """the caller's current HIP stream on the tensor's device, as an integer handle that remembers its device"""
    return stream_on(t.device)


def stream_of_device(device):
    return stream_on(device)


# the raw handle of the current stream without building a torch.cuda.Stream object (1.9 -> 0.3 us of the ~15 us a plug-in
# call spends on the host, `tools/host_overhead.py`); the public accessor is the fallback if a PyTorch build lacks it
_raw_current_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_on(device, handle=None):
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    if handle is not None:
        h = StreamHandle(int(handle))
    elif _raw_current_stream is not None:
        h = StreamHandle(_raw_current_stream(index))
    else:
        h = StreamHandle(torch.cuda.current_stream(device).cuda_stream)
    h.device_index = index
    return h


class Mailbox:
    """A few 64-bit words of pinned, device-mapped host memory (ct_mailbox_alloc): where a kernel leaves the one number the host
    has to wait for — the sparse-bitmask codec's nnz, marlin-24's structure verdict — without a D2H copy.  One per (thread,
    device): a call fills its word, launches, waits and reads before it returns, so a thread never has two waits in flight."""

    WORDS = 264  # 0: nnz of a sparse-bitmask compress, 1: marlin-24's verdict, 8..263: the nnz words of a batch (codec.bitmask_compress_many:
    BATCH_WORD0, BATCH_WORDS = 8, 256  # two windows of 128 tensors in flight)
    M24_VERDICT_WORKSPACE_BYTES = 8320  # CT_M24_VERDICT_WORKSPACE_BYTES of include/ct_hip.h

    def __init__(self, device_index: int):
        self.device_index = device_index
        self._verdict_ws = None
        lib = load()
        h, d = ctypes.c_void_p(), ctypes.c_void_p()
        with torch.cuda.device(device_index):
            check(lib.ct_mailbox_alloc(8 * self.WORDS, ctypes.byref(h), ctypes.byref(d)))
        self.host, self.dev = h.value, d.value
        self.words = (ctypes.c_int64 * self.WORDS).from_address(self.host)
        self._out = ctypes.c_int64()
        self._free = lib.ct_mailbox_free

    def wait_word(self, index: int, pending: int, stream) -> int:
        """the word once the device has replaced `pending` (or the stream has drained)"""
        check(load().ct_mailbox_wait_i64(self.host + 8 * index, pending, stream, ctypes.byref(self._out)))
        return self._out.value

    def verdict_workspace(self) -> int:
        """device address of this (thread, device)'s ticket tree for ct_marlin24_compress_w4_verdict: zeroed once, left zero by every
        launch that delivers its verdict — and the caller of that entry reads the verdict before it returns, so the thread never has
        two launches on it.  `drop_verdict_workspace` after a call that failed: the next one gets a fresh, zeroed tree."""
        ws = self._verdict_ws
        if ws is None:
            ws = self._verdict_ws = torch.zeros(self.M24_VERDICT_WORKSPACE_BYTES // 4, dtype=torch.int32, device=torch.device("cuda", self.device_index))
            if ws.data_ptr() % 128:
                raise RuntimeError("the caching allocator returned a block that is not 128-byte aligned")
            torch.cuda.synchronize(self.device_index)  # the zeros are written on the CURRENT stream; the tree may be used on any
        return ws.data_ptr()

    def drop_verdict_workspace(self) -> None:
        self._verdict_ws = None

    def __del__(self):
        try:
            self._free(self.host)
        except Exception:
            pass


_tls = threading.local()


def mailbox(device_index: int) -> Mailbox:
    boxes = getattr(_tls, "boxes", None)
    if boxes is None:
        boxes = _tls.boxes = {}
    mb = boxes.get(device_index)
    if mb is None:
        mb = boxes[device_index] = Mailbox(device_index)
    return mb


_HOSTPATH = []  # [module or None], resolved on first use


def hostpath():
    """the C++ host extension (csrc/host/ct_hostpath.cpp, built by __graft_entry__.build()) with the C-ABI entries it launches bound
    by address, or None when it has not been built — every caller keeps a Python path that does the same work more slowly"""
    if not _HOSTPATH:
        try:
            from . import _hostpath as hp
        except ImportError as e:
            hp = None
            # "file absent" is the documented optional case and stays quiet; anything else (a torch / CPython ABI mismatch of the
            # prebuilt binary, a missing libtorch_python symbol) would silently cost the model API its C++ loop: say so, once
            built = any(f.startswith("_hostpath.") and f.endswith(".so") for f in os.listdir(os.path.dirname(os.path.abspath(__file__))))
            if built:
                import warnings

                warnings.warn(f"compressed_tensors_amd: _hostpath.so is present but did not import ({e!r}); the Python host loop is used "
                              "instead (same results, more host time per module) — rebuild with __graft_entry__.build()", RuntimeWarning)
        if hp is not None:
            lib = load()
            # lib[name]: the symbol itself — `lib.name` may have been replaced by a launch-counting wrapper (tests/ref_suite)
            abi = {name: ctypes.cast(lib[name], ctypes.c_void_p).value
                   for name in ("ct_bitmask_compress", "ct_bitmask_compress_workspace_bytes", "ct_mailbox_wait_i64", "ct_stream_wait",
                                "ct_marlin24_compress_w4_full", "ct_marlin24_compress_w4_verdict", "ct_bitmask_batch_plan", "ct_bitmask_compress_batch",
                                "ct_copy_batch_plan", "ct_copy_batch", "ct_bitmask_decompress_batch_plan", "ct_bitmask_decompress_batch")}
            try:  # the HIP runtime libct_hip.so is linked against, only if it is already in the process (RTLD_NOLOAD: never a second copy)
                hip = ctypes.CDLL("libamdhip64.so", mode=os.RTLD_NOLOAD | os.RTLD_NOW)
                abi["hipStreamSynchronize"] = ctypes.cast(hip.hipStreamSynchronize, ctypes.c_void_p).value
            except (OSError, AttributeError):
                pass  # the marlin-24 default mode then waits through ct_stream_wait
            hp.bind_abi(abi)
            if hasattr(hp, "bind_pack"):  # the loop over pack-quantized modules of the word widths without a table launches these two by address
                hp.bind_pack(ctypes.cast(lib["ct_quant_pack"], ctypes.c_void_p).value, ctypes.cast(lib["ct_unpack_dequant"], ctypes.c_void_p).value,
                             ctypes.cast(lib["ct_gidx_col_group"], ctypes.c_void_p).value, ctypes.cast(lib["ct_pack_int32_dim0"], ctypes.c_void_p).value,
                             ctypes.cast(lib["ct_unpack_int32_dim0"], ctypes.c_void_p).value)
        _HOSTPATH.append(hp)
    return _HOSTPATH[0]


def host_path_kind() -> str:
    """"native" when the C++ host loop is the one the plug-in classes run, "python" otherwise (bench.py prints it in its line)"""
    return "native" if hostpath() is not None else "python"


def stream_wait(stream) -> None:
    """block until everything queued on `stream` (a StreamHandle) has completed: a spin on hipStreamQuery, no copy, no event"""
    check(load().ct_stream_wait(stream))


def require_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "compressed_tensors_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm; "
            "no GPU is available and there is no CPU fallback."
        )
    return torch.device("cuda", torch.cuda.current_device())
