"""zstd_amd — MI355X (gfx950) Zstandard block-compression core.

Python is plumbing only: this module binds the C ABI of zstd_amd/libzstd_hip.so (include/zstd_hip.h) with ctypes
and uses torch solely for device buffers / streams.  There is NO CPU fallback: if the HIP library is missing or no
GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZHIP_LIB") or os.path.join(_HERE, "libzstd_hip.so")     # $ZHIP_LIB: another build of the same library (A/B timing)
UNIT_SIZE_MAX = 131072
_lib = None


class ZhipError(RuntimeError):
    pass


def lib():
    """load libzstd_hip.so (must have been built: python -m zstd_amd.build or __graft_entry__.build())"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZhipError(f"{LIB_PATH} is missing: build it with `python zstd_amd/build.py` (no CPU fallback exists)")
        try:
            import torch  # noqa: F401  (torch bundles its own libamdhip64: load it FIRST so the process has ONE HIP runtime)
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.zhip_device_count.restype = C.c_int
        L.zhip_create.restype = C.c_void_p
        L.zhip_create.argtypes = [C.c_int, C.c_size_t]
        L.zhip_destroy.argtypes = [C.c_void_p]
        L.zhip_last_error.restype = C.c_char_p
        L.zhip_last_error.argtypes = [C.c_void_p]
        L.zhip_isError.restype = C.c_uint
        L.zhip_isError.argtypes = [C.c_size_t]
        L.zhip_getErrorName.restype = C.c_char_p
        L.zhip_getErrorName.argtypes = [C.c_size_t]
        L.zhip_compressBound.restype = C.c_size_t
        L.zhip_compressBound.argtypes = [C.c_size_t, C.c_size_t]
        L.zhip_getCParams.restype = C.c_int
        L.zhip_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p]
        L.zhip_parse_device.restype = C.c_size_t
        L.zhip_parse_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]
        L.zhip_get_sequences.restype = C.c_size_t
        L.zhip_get_sequences.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.zhip_last_timing.restype = None
        L.zhip_last_timing.argtypes = [C.c_void_p, C.c_void_p]
        L.zhip_last_hc_timing.restype = None
        L.zhip_last_hc_timing.argtypes = [C.c_void_p, C.c_void_p]
        L.zhip_last_stats.restype = C.c_size_t
        L.zhip_last_stats.argtypes = [C.c_void_p, C.c_void_p]
        for name, res, args in [
            ("zhip_compress", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p]),
            ("zhip_compress_device", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]),
            ("zhip_prepare_sequences", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
            ("zhip_set_frame_checksum", C.c_int, [C.c_void_p, C.c_int]),
            ("zhip_seek_table_bound", C.c_size_t, [C.c_size_t, C.c_int]),
            ("zhip_write_seek_table", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
            ("zhip_compress_seekable", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t]),
            ("zhip_create_cdict", C.c_void_p, [C.c_int, C.c_void_p, C.c_size_t, C.c_int]),
            ("zhip_free_cdict", None, [C.c_void_p]),
            ("zhip_create_for_records", C.c_void_p, [C.c_int, C.c_size_t, C.c_size_t]),
            ("zhip_records_bound", C.c_size_t, [C.c_void_p, C.c_size_t]),
            ("zhip_compress_records_device", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
            ("zhip_compress_records", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
            ("zhip_create_dctx", C.c_void_p, [C.c_int]),
            ("zhip_free_dctx", None, [C.c_void_p]),
            ("zhip_dctx_last_error", C.c_char_p, [C.c_void_p]),
            ("zhip_create_ddict", C.c_void_p, [C.c_int, C.c_void_p, C.c_size_t]),
            ("zhip_free_ddict", None, [C.c_void_p]),
            ("zhip_ddict_id", C.c_uint, [C.c_void_p]),
            ("zhip_find_frames", C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
            ("zhip_decompress_frames_device", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
            ("zhip_decompress", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
            ("zhip_dctx_last_timing", None, [C.c_void_p, C.c_void_p]),
            ("zhip_dctx_set_bigframe_min", None, [C.c_void_p, C.c_ulonglong]),
            ("zhip_dctx_last_bigframe", None, [C.c_void_p, C.c_void_p]),
            ("zhip_seekable_read", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_ulonglong]),
        ]:
            if hasattr(L, name):
                getattr(L, name).restype = res
                getattr(L, name).argtypes = args
        _lib = L
    return _lib


def get_cparams(level, src_size):
    out = (C.c_uint * 7)()
    if lib().zhip_getCParams(level, src_size, out) != 0:
        raise ZhipError(f"level {level} / size {src_size}: strategy not implemented on device")
    return list(out)


def compress_bound(src_size, unit_size=UNIT_SIZE_MAX):
    return lib().zhip_compressBound(src_size, unit_size)


_WL = None


def datagen(size, match_pct=50, seed=0, stream_mode=True, lit_proba=0.0):
    """bench / test input: programs/datagen.c restated on the host (zstd_amd/workloads_src/zhip_datagen.h, built into
    libzhip_workloads.so — not part of the product library): `datagen -g<size> -P<pct> -s<seed>`"""
    global _WL
    if _WL is None:
        from . import build as _b
        _b.build()
        _WL = C.CDLL(_b.WORKLOADS)
        _WL.zhip_wl_datagen.restype = None
        _WL.zhip_wl_datagen.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_uint, C.c_int]
    a = np.empty(max(size, 1), dtype=np.uint8)
    _WL.zhip_wl_datagen(a.ctypes.data_as(C.c_void_p), size, match_pct / 100.0, lit_proba, seed, 1 if stream_mode else 0)
    return a[:size]


class MultiContext:
    """host buffers over several devices in one process (zhip_compress_multi): pinned double-buffered lanes, ordered host gather"""

    def __init__(self, devices, chunk_units=0):
        L = lib()
        L.zhip_multi_create.restype = C.c_void_p
        L.zhip_multi_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.zhip_multi_destroy.restype = None
        L.zhip_multi_destroy.argtypes = [C.c_void_p]
        L.zhip_compress_multi.restype = C.c_size_t
        L.zhip_compress_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.zhip_multi_last_error.restype = C.c_char_p
        L.zhip_multi_last_error.argtypes = [C.c_void_p]
        L.zhip_multi_last_seconds.restype = C.c_double
        L.zhip_multi_last_seconds.argtypes = [C.c_void_p]
        L.zhip_multi_last_stages.restype = None
        L.zhip_multi_last_stages.argtypes = [C.c_void_p, C.c_void_p]
        L.zhip_multi_set_frame_checksum.argtypes = [C.c_void_p, C.c_int]
        arr = (C.c_int * len(devices))(*devices)
        self._h = L.zhip_multi_create(arr, len(devices), chunk_units)
        if not self._h:
            raise ZhipError("zhip_multi_create failed")

    def compress_into(self, dst, data, level=1, unit_size=UNIT_SIZE_MAX, cparams=None, sizes=None):
        """data, dst: numpy uint8 arrays (dst >= compress_bound); returns the compressed size"""
        L = lib()
        cp = (C.c_uint * 7)(*cparams) if cparams is not None else None
        r = L.zhip_compress_multi(self._h, dst.ctypes.data_as(C.c_void_p), dst.nbytes, data.ctypes.data_as(C.c_void_p), data.nbytes, level, cp, unit_size,
                                  sizes.ctypes.data_as(C.c_void_p) if sizes is not None else None)
        if L.zhip_isError(r):
            raise ZhipError(f"zhip_compress_multi: {L.zhip_getErrorName(r).decode()} ({L.zhip_multi_last_error(self._h).decode()})")
        return int(r)

    def compress(self, data, level=1, unit_size=UNIT_SIZE_MAX, cparams=None):
        data = np.ascontiguousarray(data)
        dst = np.empty(compress_bound(data.nbytes, unit_size), dtype=np.uint8)
        return dst[: self.compress_into(dst, data, level, unit_size, cparams)].tobytes()

    def compress_frame_mt_into(self, dst, data, level=1, cparams=None, job_size=0, overlap_log=0):
        """ONE frame with the reference's ZSTD_c_nbWorkers >= 1 bytes, its jobs spread over the lanes (zhip_compress_frame_mt_multi)"""
        L = lib()
        L.zhip_compress_frame_mt_multi.restype = C.c_size_t
        L.zhip_compress_frame_mt_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
        cp = (C.c_uint * 7)(*cparams) if cparams is not None else None
        r = L.zhip_compress_frame_mt_multi(self._h, dst.ctypes.data_as(C.c_void_p), dst.nbytes, data.ctypes.data_as(C.c_void_p), data.nbytes, level, cp,
                                           job_size, overlap_log)
        if L.zhip_isError(r):
            raise ZhipError(f"zhip_compress_frame_mt_multi: {L.zhip_getErrorName(r).decode()} ({L.zhip_multi_last_error(self._h).decode()})")
        return int(r)

    def compress_frame_mt(self, data, level=1, cparams=None, job_size=0, overlap_log=0):
        data = np.ascontiguousarray(data)
        dst = np.empty(compress_bound(data.nbytes, UNIT_SIZE_MAX) + 64, dtype=np.uint8)
        return dst[: self.compress_frame_mt_into(dst, data, level, cparams, job_size, overlap_log)].tobytes()

    def set_checksum(self, enable=True):
        lib().zhip_multi_set_frame_checksum(self._h, 1 if enable else 0)

    def last_seconds(self):
        return float(lib().zhip_multi_last_seconds(self._h))

    def last_stages(self):
        """seconds summed over chunks and lanes of the most recent compress call, per stage (zhip_multi_last_stages)"""
        out = (C.c_double * 7)()
        lib().zhip_multi_last_stages(self._h, out)
        keys = ("host_copy_in_s", "h2d_s", "kernels_s", "d2h_s", "ordered_gather_wait_s", "host_copy_out_s", "chunks")
        return {k: (int(out[i]) if k == "chunks" else round(out[i], 5)) for i, k in enumerate(keys)}

    def close(self):
        if self._h:
            lib().zhip_multi_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CDict:
    """a dictionary digested for one GPU (the role ZSTD_CDict plays): parameters + tagged hash tables built on the host
    like ZSTD_createCDict builds them, uploaded once"""

    def __init__(self, dict_bytes, level=3, device=0):
        a = np.frombuffer(dict_bytes, dtype=np.uint8) if not isinstance(dict_bytes, np.ndarray) else dict_bytes
        self._keep = np.ascontiguousarray(a)
        self._h = lib().zhip_create_cdict(device, self._keep.ctypes.data_as(C.c_void_p), self._keep.size, level)
        if not self._h:
            raise ZhipError(f"zhip_create_cdict(level={level}, {self._keep.size} B): not supported on device (ZDICT entropy tables, or a "
                            "CDict strategy above dfast) or no GPU")
        self.level, self.device = level, device

    def close(self):
        if self._h:
            lib().zhip_free_cdict(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """owns the device-side state for one GPU (the role ZSTD_CCtx plays for the CPU library)"""

    def __init__(self, device=0, max_units=1024, records_total_bytes=None):
        """max_units 128 KB units per call; or, for the dictionary path, max_units small records of records_total_bytes in all"""
        L = lib()
        if L.zhip_device_count() <= 0:
            raise ZhipError("no HIP device visible (zstd_amd has no CPU path)")
        self._h = L.zhip_create(device, max_units) if records_total_bytes is None else L.zhip_create_for_records(device, max_units, records_total_bytes)
        if not self._h:
            raise ZhipError(f"zhip_create(device={device}, max_units={max_units}) failed")
        self.device = device
        self.max_units = max_units

    def close(self):
        if self._h:
            lib().zhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, r, what):
        L = lib()
        if L.zhip_isError(r):
            raise ZhipError(f"{what}: {L.zhip_getErrorName(r).decode()} ({L.zhip_last_error(self._h).decode()})")
        return r

    def timing(self):
        t = (C.c_double * 4)()
        lib().zhip_last_timing(self._h, t)
        return {"parse_ms": t[0], "entropy_ms": t[1], "gather_ms": t[2], "total_ms": t[3]}

    def set_checksum(self, enable=True):
        """ZSTD_c_checksumFlag: every frame carries XXH64's low 32 bits of its content"""
        lib().zhip_set_frame_checksum(self._h, 1 if enable else 0)

    def hc_timing(self):
        """hash-chain levels: the match-finder stage of the last call split by kernel (ms)"""
        t = (C.c_double * 3)()
        lib().zhip_last_hc_timing(self._h, t)
        return {"chain_ms": t[0], "search_ms": t[1], "parse_ms": t[2]}

    def stats(self):
        s = (C.c_ulonglong * 5)()
        self._check(lib().zhip_last_stats(self._h, s), "zhip_last_stats")
        return {"units": s[0], "src_bytes": s[1], "dst_bytes": s[2], "sequences": s[3], "literals": s[4]}

    # ---- stage 1 only (sequence-producer path)
    def parse_device(self, src_ptr, src_size, level=1, unit_size=UNIT_SIZE_MAX, stream=None):
        return self._check(lib().zhip_parse_device(self._h, src_ptr, src_size, level, unit_size, stream), "zhip_parse_device")

    def get_sequences(self, unit_index, cap=UNIT_SIZE_MAX // 3 + 8):
        out = np.zeros((cap, 4), dtype=np.uint32)
        n = self._check(lib().zhip_get_sequences(self._h, unit_index, out.ctypes.data_as(C.c_void_p), cap), "zhip_get_sequences")
        return out[:n]

    def compress_seekable(self, data, level=1, unit_size=UNIT_SIZE_MAX):
        """host bytes -> frames + seek table (contrib/seekable_format); checksums go into the table when set_checksum is on"""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = a.size
        nunits = max(1, -(-n // unit_size))
        cap = compress_bound(n, unit_size) + lib().zhip_seek_table_bound(nunits, 1)
        dst = np.empty(cap, dtype=np.uint8)
        r = self._check(lib().zhip_compress_seekable(self._h, dst.ctypes.data_as(C.c_void_p), cap, a.ctypes.data_as(C.c_void_p) if n else None,
                                                     n, level, unit_size), "zhip_compress_seekable")
        return dst[:r].tobytes()

    # ---- dictionary path: many small records, one frame each
    def compress_records(self, cdict, records, return_sizes=False):
        """records: list of bytes / uint8 arrays -> concatenated frames (bytes)"""
        arrs = [np.frombuffer(r, dtype=np.uint8) if not isinstance(r, np.ndarray) else r for r in records]
        offs = np.concatenate([[0], np.cumsum([a.size for a in arrs])]).astype(np.uint64)
        flat = np.concatenate(arrs + [np.zeros(8, np.uint8)]) if arrs else np.zeros(8, np.uint8)
        cap = lib().zhip_records_bound(offs.ctypes.data_as(C.c_void_p), len(arrs)) + 64
        dst = np.empty(cap, dtype=np.uint8)
        sizes = np.zeros(max(1, len(arrs)), dtype=np.uint64)
        r = self._check(lib().zhip_compress_records(self._h, cdict._h, dst.ctypes.data_as(C.c_void_p), cap, flat.ctypes.data_as(C.c_void_p),
                                                    offs.ctypes.data_as(C.c_void_p), len(arrs), sizes.ctypes.data_as(C.c_void_p)), "zhip_compress_records")
        out = dst[:r].tobytes()
        return (out, sizes[:len(arrs)]) if return_sizes else out

    def compress_records_device(self, cdict, dst_ptr, dst_cap, src_ptr, offsets, sizes_ptr=None, stream=None):
        """offsets: host np.uint64 array with nRec+1 entries; src/dst are device pointers"""
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        return self._check(lib().zhip_compress_records_device(self._h, cdict._h, dst_ptr, dst_cap, src_ptr, offs.ctypes.data_as(C.c_void_p),
                                                              len(offs) - 1, sizes_ptr, stream), "zhip_compress_records_device")

    # ---- full pipeline
    def set_row_matcher(self, mode):
        """greedy / lazy / lazy2: 0 auto = the reference's default (row-hash matcher when windowLog > 14, salt of a fresh CCtx),
        2 = hash-chain matcher (ZSTD_c_useRowMatchFinder = ZSTD_ps_disable)"""
        L = lib()
        L.zhip_set_row_matcher.argtypes = [C.c_void_p, C.c_int]
        if L.zhip_set_row_matcher(self._h, int(mode)) != 0:
            raise ZhipError("zhip_set_row_matcher: bad mode")

    def set_prediction(self, units=None, frames=None):
        """the row matcher's two-pass prediction for units / for multi-block frames (same bytes either way; units: off by default, frames: on behind a 32 KB probe — DESIGN.md 4.2b "the live rows", 4.7c)"""
        L = lib()
        L.zhip_set_prediction.restype = C.c_int
        L.zhip_set_prediction.argtypes = [C.c_void_p, C.c_int, C.c_int]
        if L.zhip_set_prediction(self._h, -1 if units is None else int(bool(units)), -1 if frames is None else int(bool(frames))) != 0:
            raise ZhipError("zhip_set_prediction: bad value")

    def set_live_rows(self, on=True):
        """the frame kernels' live rows: on (default) / off = their live searches walk the links (zhip_set_live_rows); same bytes; the unit kernels keep no rows"""
        L = lib()
        L.zhip_set_live_rows.restype = C.c_int
        L.zhip_set_live_rows.argtypes = [C.c_void_p, C.c_int]
        L.zhip_set_live_rows(self._h, int(bool(on)))

    def compress_device(self, dst_ptr, dst_cap, src_ptr, src_size, level=1, unit_size=UNIT_SIZE_MAX, sizes_ptr=None, stream=None):
        return self._check(lib().zhip_compress_device(self._h, dst_ptr, dst_cap, src_ptr, src_size, level, unit_size,
                                                      sizes_ptr, stream), "zhip_compress_device")

    def compress(self, data, level=1, unit_size=UNIT_SIZE_MAX, return_sizes=False):
        """host bytes -> concatenated frames (bytes)"""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        n = a.size
        cap = compress_bound(n, unit_size)
        dst = np.empty(cap, dtype=np.uint8)
        nunits = max(1, -(-n // unit_size))
        sizes = np.zeros(nunits, dtype=np.uint64)
        r = self._check(lib().zhip_compress(self._h, dst.ctypes.data_as(C.c_void_p), cap,
                                            a.ctypes.data_as(C.c_void_p) if n else None, n, level, unit_size,
                                            sizes.ctypes.data_as(C.c_void_p)), "zhip_compress")
        out = dst[:r].tobytes()
        return (out, sizes) if return_sizes else out

    def compress_frames(self, buffers, level=1, cparams=None, workers=0, job_size=0, overlap_log=0):
        """each buffer -> ONE multi-block frame, byte-identical to the reference's ZSTD_compress of it (zhip_compress_frames;
        strategies ZSTD_fast ... ZSTD_lazy2: levels -N .. 12).  Returns the list of frames (bytes).
        workers >= 1: the frames ZSTD_compress2 emits with ZSTD_c_nbWorkers >= 1 instead (zhip_compress_frames_mt: inputs above
        512 KB are cut into independent jobs of job_size with overlap_log's prefix — one large input then fills the GPU)."""
        L = lib()
        L.zhip_frames_bound.restype = C.c_size_t
        L.zhip_frames_bound.argtypes = [C.c_void_p, C.c_size_t]
        L.zhip_compress_frames.restype = C.c_size_t
        L.zhip_compress_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        arrs = [np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b for b in buffers]
        offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([a.size for a in arrs])
        src = np.concatenate(arrs + [np.zeros(1, dtype=np.uint8)])
        cap = L.zhip_frames_bound(offs.ctypes.data_as(C.c_void_p), len(arrs))
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        sizes = np.zeros(len(arrs), dtype=np.uint64)
        cp = (C.c_uint * 7)(*cparams) if cparams is not None else None
        if workers:
            L.zhip_compress_frames_mt.restype = C.c_size_t
            L.zhip_compress_frames_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                                  C.c_size_t, C.c_int, C.c_void_p]
            r = self._check(L.zhip_compress_frames_mt(self._h, dst.ctypes.data_as(C.c_void_p), cap, src.ctypes.data_as(C.c_void_p),
                                                      offs.ctypes.data_as(C.c_void_p), len(arrs), level, cp, job_size, overlap_log,
                                                      sizes.ctypes.data_as(C.c_void_p)), "zhip_compress_frames_mt")
        else:
            r = self._check(L.zhip_compress_frames(self._h, dst.ctypes.data_as(C.c_void_p), cap, src.ctypes.data_as(C.c_void_p),
                                                   offs.ctypes.data_as(C.c_void_p), len(arrs), level, cp, sizes.ctypes.data_as(C.c_void_p)),
                            "zhip_compress_frames")
        out, pos = [], 0
        for z in sizes:
            out.append(dst[pos: pos + int(z)].tobytes()); pos += int(z)
        assert pos == r
        return out


# ---------------------------------------------------------------------------------------------------------------- decompression
def find_frames(data):
    """walk concatenated frames in host bytes -> dict of uint64 arrays: src_off, src_size, content (2**64-1 = not stated), bound"""
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    L = lib()
    p = a.ctypes.data_as(C.c_void_p) if a.size else None
    n = L.zhip_find_frames(p, a.size, None, None, None, None, 0)
    if L.zhip_isError(n):
        raise ZhipError(f"zhip_find_frames: {L.zhip_getErrorName(n).decode()} (code {(1 << 64) - n})")
    out = {k: np.zeros(max(n, 1), dtype=np.uint64) for k in ("src_off", "src_size", "content", "bound")}
    if n:
        L.zhip_find_frames(p, a.size, *[out[k].ctypes.data_as(C.c_void_p) for k in ("src_off", "src_size", "content", "bound")], n)
    return {k: v[:n] for k, v in out.items()}


class DDict:
    """a dictionary digested for decoding on one GPU (the role ZSTD_DDict plays): content + entropy tables in decoding form"""

    def __init__(self, dict_bytes, device=0):
        a = np.frombuffer(dict_bytes, dtype=np.uint8) if not isinstance(dict_bytes, np.ndarray) else dict_bytes
        a = np.ascontiguousarray(a)
        self._h = lib().zhip_create_ddict(device, a.ctypes.data_as(C.c_void_p), a.size)
        if not self._h:
            raise ZhipError(f"zhip_create_ddict({a.size} B): malformed dictionary or no GPU")
        self.device = device
        self.dict_id = lib().zhip_ddict_id(self._h)

    def close(self):
        if self._h:
            lib().zhip_free_ddict(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DContext:
    """owns the decoder's device state for one GPU (the role ZSTD_DCtx plays)"""

    def __init__(self, device=0):
        L = lib()
        if L.zhip_device_count() <= 0:
            raise ZhipError("no HIP device visible (zstd_amd has no CPU path)")
        self._h = L.zhip_create_dctx(device)
        if not self._h:
            raise ZhipError(f"zhip_create_dctx(device={device}) failed")
        self.device = device

    def close(self):
        if self._h:
            lib().zhip_free_dctx(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def timing(self):
        t = (C.c_double * 2)()
        lib().zhip_dctx_last_timing(self._h, t)
        return {"decode_ms": t[0], "verify_ms": t[1]}

    def set_bigframe_min(self, min_content):
        """frames stating at least this much content are decoded block-parallel (zhip_decode_big.h); 0 = never"""
        lib().zhip_dctx_set_bigframe_min(self._h, int(min_content))

    def last_bigframe(self):
        t = (C.c_uint * 4)()
        lib().zhip_dctx_last_bigframe(self._h, t)
        return {"block_parallel": t[0], "fell_back": t[1], "jump_rounds": t[2], "blocks": t[3]}

    def _err(self, r, what):
        L = lib()
        raise ZhipError(f"{what}: zstd error {(1 << 64) - r} ({L.zhip_dctx_last_error(self._h).decode()})")

    def decompress(self, data, capacity=None, ddict=None):
        """host bytes holding concatenated frames -> their contents back to back (= ZSTD_decompress)"""
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        if capacity is None:
            fr = find_frames(a)
            capacity = int(sum(int(c) if c != np.uint64(2**64 - 1) else int(b) for c, b in zip(fr["content"], fr["bound"])))
        dst = np.empty(max(capacity, 1), dtype=np.uint8)
        L = lib()
        r = L.zhip_decompress(self._h, ddict._h if ddict else None, dst.ctypes.data_as(C.c_void_p), capacity,
                              a.ctypes.data_as(C.c_void_p) if a.size else None, a.size)
        if L.zhip_isError(r):
            self._err(r, "zhip_decompress")
        return dst[:r].tobytes()

    def seekable_read(self, blob, offset, length):
        """random access into a seekable file (host bytes): only the frames overlapping [offset, offset+length) are decoded"""
        a = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else blob
        dst = np.empty(max(length, 1), dtype=np.uint8)
        L = lib()
        r = L.zhip_seekable_read(self._h, dst.ctypes.data_as(C.c_void_p), length, a.ctypes.data_as(C.c_void_p), a.size, offset)
        if L.zhip_isError(r):
            self._err(r, "zhip_seekable_read")
        return dst[:r].tobytes()

    def decompress_frames_device(self, dst_ptr, dst_offsets, dst_caps, src_ptr, src_offsets, src_sizes, ddict=None, stream=None, check=True):
        """device buffers; descriptor arrays are host uint64 arrays. returns (total, status[uint32], sizes[uint64])"""
        n = len(src_offsets)
        arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (dst_offsets, dst_caps, src_offsets, src_sizes)]
        status = np.zeros(max(n, 1), dtype=np.uint32)
        sizes = np.zeros(max(n, 1), dtype=np.uint64)
        L = lib()
        r = L.zhip_decompress_frames_device(self._h, ddict._h if ddict else None, dst_ptr, arrs[0].ctypes.data_as(C.c_void_p),
                                            arrs[1].ctypes.data_as(C.c_void_p), src_ptr, arrs[2].ctypes.data_as(C.c_void_p),
                                            arrs[3].ctypes.data_as(C.c_void_p), n, status.ctypes.data_as(C.c_void_p),
                                            sizes.ctypes.data_as(C.c_void_p), stream)
        if check and L.zhip_isError(r):
            self._err(r, "zhip_decompress_frames_device")
        return r, status[:n], sizes[:n]
