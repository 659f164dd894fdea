"""ctypes binding of libmfa_hip.so (include/mfa_hip.h).

Thin by design: the product is the C-ABI library; Python is used by the tests and the
bench to allocate device memory (torch) and to launch ranks (torch.distributed).
There is no fallback: if the library is missing, loading raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MFA_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libmfa_hip.so")      # (the override: development builds)

OK = 0
KERNEL_NONE, KERNEL_WALK, KERNEL_SPECIALISED, KERNEL_TABLE = 0, 1, 2, 3
ERR_INVALID_ARG, ERR_BAD_BLOB, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_TOO_LONG, ERR_JIT = -1, -2, -3, -4, -5, -6, -7, -8

EXPORTS = ["mfa_image_create", "mfa_image_destroy", "mfa_image_get_info", "mfa_image_prepare", "mfa_image_specialize", "mfa_match_batch",
           "mfa_match_batch_regions", "mfa_region_scan", "mfa_match_batch_host", "mfa_last_kernel_ms", "mfa_last_region_ms",
           "mfa_device_count", "mfa_last_hip_error", "mfa_strerror", "mfa_version",
           "mfa_mixed_create", "mfa_mixed_destroy", "mfa_match_mixed", "mfa_match_mixed_sized", "mfa_match_mixed_host", "mfa_mixed_last_ms", "mfa_mixed_timing",
           "mfa_mixed_last_launches", "mfa_pack_result_bitmap"]

REGION_WORDS, REGION_MAX, REGION_OVERFLOW, REGION_MIN_LEN = 16, 15, 0x100, 64


class MfaError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().mfa_strerror(code).decode()
        if code == ERR_HIP:
            msg += " (hipError %d)" % lib().mfa_last_hip_error()
        super().__init__("%s: %s [%d]" % (where, msg, code))


class ImageInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in
                ("kind", "is_reversed", "n_nodes", "n_edges", "n_cells", "dfa_states", "byte_classes", "last_kernel")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmfa_hip.so is not built (run python -c 'import __graft_entry__ as g; g.build()'): "
                              + LIB_PATH)
        # A process must hold ONE HIP runtime.  torch bundles its own libamdhip64 (same soname as
        # /opt/rocm's): load torch first so the dynamic loader resolves our NEEDED entry to the copy
        # torch already mapped instead of mapping a second runtime that cannot see the device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        vp, u64, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
        L.mfa_image_create.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
        L.mfa_image_create.restype = i32
        L.mfa_image_destroy.argtypes = [vp]
        L.mfa_image_destroy.restype = None
        L.mfa_image_get_info.argtypes = [vp, ctypes.POINTER(ImageInfo)]
        L.mfa_image_prepare.argtypes = [vp, i32]
        L.mfa_image_specialize.argtypes = [vp]
        L.mfa_match_batch.argtypes = [vp, vp, vp, u64, vp, i32, vp]
        L.mfa_match_batch_regions.argtypes = [vp, vp, vp, u64, vp, vp, i32, vp]
        L.mfa_region_scan.argtypes = [vp, vp, u64, vp, i32, vp]
        L.mfa_last_region_ms.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float)]
        L.mfa_match_batch_host.argtypes = [vp, vp, vp, u64, vp, i32]
        L.mfa_last_kernel_ms.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float)]
        L.mfa_mixed_create.argtypes = [ctypes.POINTER(vp), ctypes.c_uint32, ctypes.POINTER(vp)]
        L.mfa_mixed_destroy.argtypes = [vp]
        L.mfa_mixed_destroy.restype = None
        L.mfa_match_mixed.argtypes = [vp, vp, vp, u64, ctypes.POINTER(u64), vp, i32, vp]
        L.mfa_match_mixed_sized.argtypes = [vp, vp, vp, u64, u64, ctypes.POINTER(u64), vp, i32, vp]
        L.mfa_mixed_last_launches.argtypes = [vp, i32] + [ctypes.POINTER(ctypes.c_uint32)] * 4
        L.mfa_pack_result_bitmap.argtypes = [vp, u64, vp, vp]
        L.mfa_match_mixed_host.argtypes = [vp, vp, vp, u64, ctypes.POINTER(u64), vp, i32]
        L.mfa_mixed_last_ms.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.mfa_mixed_timing.argtypes = [vp, i32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.mfa_strerror.argtypes = [i32]
        L.mfa_strerror.restype = ctypes.c_char_p
        L.mfa_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc, where):
    if rc != OK:
        raise MfaError(rc, where)


class Image:
    """An automaton image handle (mfa_image_t*)."""

    def __init__(self, blob):
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(blob), len(blob))
        _check(lib().mfa_image_create(buf, len(blob), ctypes.byref(self._h)), "mfa_image_create")

    def info(self):
        out = ImageInfo()
        _check(lib().mfa_image_get_info(self._h, ctypes.byref(out)), "mfa_image_get_info")
        return {n: getattr(out, n) for n, _ in ImageInfo._fields_}

    def specialize(self):
        """Generate and compile the automaton-specific kernel into the on-disk cache (no GPU needed).
        Returns False when the automaton is too large for a specialised kernel."""
        rc = lib().mfa_image_specialize(self._h)
        if rc == ERR_UNSUPPORTED:
            return False
        _check(rc, "mfa_image_specialize")
        return True

    def prepare(self, device=0):
        _check(lib().mfa_image_prepare(self._h, device), "mfa_image_prepare")

    def match_device(self, d_bytes_ptr, d_offsets_ptr, n, d_results_ptr, device=0, stream=0):
        """Raw device pointers (ints); asynchronous on `stream` (a hipStream_t value, 0 = default)."""
        _check(lib().mfa_match_batch(self._h, d_bytes_ptr, d_offsets_ptr, n, d_results_ptr, device,
                                     ctypes.c_void_p(stream)), "mfa_match_batch")

    def match_tensors(self, d_bytes, d_offsets, d_results=None, stream=None):
        """torch CUDA tensors: uint8 bytes, int64 offsets (n+1) -> uint8 results (n).  Enqueued on
        `stream` (torch stream; default: the current one)."""
        import torch
        n = d_offsets.numel() - 1
        dev = d_offsets.device.index or 0
        if d_results is None:
            d_results = torch.empty(max(n, 1), dtype=torch.uint8, device=d_offsets.device)
        s = stream if stream is not None else torch.cuda.current_stream(d_offsets.device)
        self.match_device(d_bytes.data_ptr(), d_offsets.data_ptr(), n, d_results.data_ptr(), dev, s.cuda_stream)
        return d_results[:n]

    def match_tensors_regions(self, d_bytes, d_offsets, d_table, d_results=None, stream=None):
        """mfa_match_batch_regions: like match_tensors, with a region table the caller has filled (region_scan) for
        exactly these strings; d_table None = no table (every step is executed)."""
        import torch
        n = d_offsets.numel() - 1
        dev = d_offsets.device.index or 0
        if d_results is None:
            d_results = torch.empty(max(n, 1), dtype=torch.uint8, device=d_offsets.device)
        s = stream if stream is not None else torch.cuda.current_stream(d_offsets.device)
        _check(lib().mfa_match_batch_regions(self._h, d_bytes.data_ptr(), d_offsets.data_ptr(), n, d_results.data_ptr(),
                                             d_table.data_ptr() if d_table is not None else None, dev,
                                             ctypes.c_void_p(s.cuda_stream)), "mfa_match_batch_regions")
        return d_results[:n]

    def last_region_ms(self, device=0):
        ms = ctypes.c_float()
        _check(lib().mfa_last_region_ms(self._h, device, ctypes.byref(ms)), "mfa_last_region_ms")
        return ms.value

    def match_host(self, data, offsets, device=0):
        """numpy uint8 data + uint64 offsets(n+1) in host memory -> numpy uint8 results."""
        import numpy as np
        n = len(offsets) - 1
        res = np.zeros(max(n, 1), dtype=np.uint8)
        _check(lib().mfa_match_batch_host(self._h, data.ctypes.data, offsets.ctypes.data, n, res.ctypes.data, device),
               "mfa_match_batch_host")
        return res[:n]

    def last_kernel_ms(self, device=0):
        ms = ctypes.c_float()
        _check(lib().mfa_last_kernel_ms(self._h, device, ctypes.byref(ms)), "mfa_last_kernel_ms")
        return ms.value

    def close(self):
        if self._h:
            lib().mfa_image_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mixed:
    """A mixed-batch handle (mfa_mixed_t*): several memory automata matched against the segments of ONE batch."""

    def __init__(self, images):
        self._images = list(images)                      # the images must outlive the handle
        self._h = ctypes.c_void_p()
        arr = (ctypes.c_void_p * len(self._images))(*[im._h for im in self._images])
        _check(lib().mfa_mixed_create(arr, len(self._images), ctypes.byref(self._h)), "mfa_mixed_create")

    def match_tensors(self, d_bytes, d_offsets, seg_first, d_results=None, stream=None, total_bytes=0):
        """seg_first: host sequence of len(images) + 1 string indices.  Asynchronous on `stream` (default: the current one).
        total_bytes: the batch's size in bytes if the caller knows it (mfa_match_mixed_sized: nothing is read back)."""
        import torch
        n = d_offsets.numel() - 1
        dev = d_offsets.device.index or 0
        if d_results is None:
            d_results = torch.empty(max(n, 1), dtype=torch.uint8, device=d_offsets.device)
        s = stream if stream is not None else torch.cuda.current_stream(d_offsets.device)
        sf = (ctypes.c_uint64 * len(seg_first))(*[int(x) for x in seg_first])
        if total_bytes:
            _check(lib().mfa_match_mixed_sized(self._h, d_bytes.data_ptr(), d_offsets.data_ptr(), n, int(total_bytes), sf, d_results.data_ptr(), dev,
                                               ctypes.c_void_p(s.cuda_stream)), "mfa_match_mixed_sized")
        else:
            _check(lib().mfa_match_mixed(self._h, d_bytes.data_ptr(), d_offsets.data_ptr(), n, sf, d_results.data_ptr(), dev,
                                         ctypes.c_void_p(s.cuda_stream)), "mfa_match_mixed")
        return d_results[:n]

    def last_launches(self, device=0):
        """what the last call launched: {"region_launches", "walk_launches", "groups", "gated"}"""
        v = [ctypes.c_uint32() for _ in range(4)]
        _check(lib().mfa_mixed_last_launches(self._h, device, *[ctypes.byref(x) for x in v]), "mfa_mixed_last_launches")
        return {"region_launches": v[0].value, "walk_launches": v[1].value, "groups": v[2].value, "gated": bool(v[3].value)}

    def last_ms(self, device=0, back=0):
        """(region launches, first region launch to last walk) of the call `back` calls ago (0 = the last one), in ms"""
        r, sp = ctypes.c_float(), ctypes.c_float()
        _check(lib().mfa_mixed_timing(self._h, device, back, ctypes.byref(r), ctypes.byref(sp)), "mfa_mixed_timing")
        return r.value, sp.value

    def close(self):
        if self._h:
            lib().mfa_mixed_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def region_scan(d_bytes, d_offsets, d_table=None, stream=None):
    """mfa_region_scan on torch CUDA tensors: returns the region table, int64 [n, REGION_WORDS]."""
    import torch
    n = d_offsets.numel() - 1
    dev = d_offsets.device.index or 0
    if d_table is None:
        d_table = torch.empty((max(n, 1), REGION_WORDS), dtype=torch.int64, device=d_offsets.device)
    s = stream if stream is not None else torch.cuda.current_stream(d_offsets.device)
    _check(lib().mfa_region_scan(d_bytes.data_ptr(), d_offsets.data_ptr(), n, d_table.data_ptr(), dev,
                                 ctypes.c_void_p(s.cuda_stream)), "mfa_region_scan")
    return d_table


def pack_result_bitmap(d_results, d_bitmap=None, stream=None):
    """mfa_pack_result_bitmap on a torch CUDA uint8 result vector: uint8 bitmap ((n + 7) // 8), bit k % 8 of byte k // 8 = string k accepted."""
    import torch
    n = d_results.numel()
    if d_bitmap is None:
        d_bitmap = torch.empty((n + 7) // 8, dtype=torch.uint8, device=d_results.device)
    s = stream if stream is not None else torch.cuda.current_stream(d_results.device)
    _check(lib().mfa_pack_result_bitmap(d_results.data_ptr(), n, d_bitmap.data_ptr(), ctypes.c_void_p(s.cuda_stream)), "mfa_pack_result_bitmap")
    return d_bitmap


def device_count():
    return lib().mfa_device_count()
