"""HDF5 demonstration files for the labelling path -- no h5py.

The reference opens its demonstration file with ``h5py.File(data_path, "a")`` (arp_dt/label_reward.py:69), reads
``g[img_key][traj, -1]`` per trajectory (:268) and writes two gzip-chunked float32 datasets (:273-289); the file is produced
by data/PPG/trajectory_recorder.py:148-176 (``ob`` uint8 ``[len, num_frames, H, W, 3]``, gzip, chunks ``(1, num_frames, H, W, 3)``,
attr ``env_name``).  h5py is a wrapper around the HDF5 C library; this module is a ctypes wrapper around the same library
(``libhdf5.so``, 1.10.3 or newer) exposing the small h5py-shaped surface ``arp_amd.label_reward`` uses -- ``H5Store`` stands where
``h5py.File`` stands -- plus the reader the reference does not have:

``H5Dataset.read_last_frames(r0, r1)``: the last stacked frame of rows ``[r0, r1)`` WITHOUT inflating every row's chunk.  A row's
chunk holds ``num_frames`` consecutive frames of its trajectory (trajectory_recorder.py:103-115: a deque of the last ``num_frames``
observations, the first one left-padded), so within a trajectory the chunk of row ``i`` already contains the last frames of rows
``i-num_frames+1 .. i``: one chunk in ``num_frames`` is read raw (``H5Dread_chunk``) and inflated, the chunks inflate in a thread
pool (zlib releases the GIL), and each yields ``num_frames`` output frames.  The reference's ``g["ob"][traj, -1]`` inflates all of
them -- 8x the bytes for ``num_frames = 8``.  The first group of every call is cross-checked against the plain per-row read; a
file that does not obey the recorder's stacking falls back to per-row reads.

The library is looked up lazily (``ARP_HDF5_LIB``, the loader's search path, ``/opt/conda/lib``); everything else in ``arp_amd``
works without it (``store=`` mappings).
"""
import ctypes as C
import ctypes.util
import os
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int
H5P_DEFAULT = 0
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
H5S_SELECT_SET = 0
H5S_SCALAR = 0
H5D_CHUNKED = 2
H5Z_FILTER_DEFLATE, H5Z_FILTER_SHUFFLE = 1, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5F_SCOPE_GLOBAL = 1

_lib = None
_lock = threading.RLock()  # the HDF5 library is not thread-safe unless built so: every call goes through this lock


class H5Error(OSError):
    pass


class _H5G_info(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint)]


def _candidates():
    if os.environ.get("ARP_HDF5_LIB"):
        yield os.environ["ARP_HDF5_LIB"]
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/local/lib", "/usr/lib64"):
        for n in ("libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so.103"):
            yield os.path.join(d, n)


def lib():
    """The HDF5 C library, loaded on first use."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        errs = []
        h = None
        for p in _candidates():
            try:
                h = C.CDLL(p)
                break
            except OSError as e:
                errs.append(f"{p}: {e}")
        if h is None:
            raise ImportError("libhdf5 not found (set ARP_HDF5_LIB=/path/to/libhdf5.so); tried:\n  " + "\n  ".join(errs[:8]))
        sig = {
            "H5open": (herr_t, []),
            "H5get_libversion": (herr_t, [C.POINTER(C.c_uint)] * 3),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
            "H5free_memory": (herr_t, [C.c_void_p]),
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
            "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]),
            "H5Fflush": (herr_t, [hid_t, C.c_int]),
            "H5Gget_info": (herr_t, [hid_t, C.POINTER(_H5G_info)]),
            "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Ldelete": (herr_t, [hid_t, C.c_char_p, hid_t]),
            "H5Pcreate": (hid_t, [hid_t]),
            "H5Pclose": (herr_t, [hid_t]),
            "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]),
            "H5Pget_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
            "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]),
            "H5Pget_layout": (C.c_int, [hid_t]),
            "H5Pget_nfilters": (C.c_int, [hid_t]),
            "H5Pget_filter2": (C.c_int, [hid_t, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_size_t), C.POINTER(C.c_uint), C.c_size_t,
                                         C.c_char_p, C.POINTER(C.c_uint)]),
            "H5Screate": (hid_t, [C.c_int]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sclose": (herr_t, [hid_t]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dclose": (herr_t, [hid_t]),
            "H5Dget_space": (hid_t, [hid_t]),
            "H5Dget_type": (hid_t, [hid_t]),
            "H5Dget_create_plist": (hid_t, [hid_t]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dset_extent": (herr_t, [hid_t, C.POINTER(hsize_t)]),
            "H5Dread_chunk": (herr_t, [hid_t, hid_t, C.POINTER(hsize_t), C.POINTER(C.c_uint32), C.c_void_p]),
            "H5Dget_chunk_storage_size": (herr_t, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Tget_class": (C.c_int, [hid_t]),
            "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tget_sign": (C.c_int, [hid_t]),
            "H5Tget_super": (hid_t, [hid_t]),
            "H5Tis_variable_str": (C.c_int, [hid_t]),
            "H5Tcopy": (hid_t, [hid_t]),
            "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
            "H5Tset_cset": (herr_t, [hid_t, C.c_int]),
            "H5Tclose": (herr_t, [hid_t]),
            "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
            "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
            "H5Aget_type": (hid_t, [hid_t]),
            "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aclose": (herr_t, [hid_t]),
            "H5Adelete": (herr_t, [hid_t, C.c_char_p]),
        }
        for name, (res, args) in sig.items():
            try:
                fn = getattr(h, name)
            except AttributeError as e:
                raise ImportError(f"{h._name}: missing {name} (HDF5 >= 1.10.3 is required)") from e
            fn.restype, fn.argtypes = res, args
        for opt in ("H5Pset_small_data_block_size", "H5Pset_meta_block_size"):  # allocation block sizes of a writer (H5Store.__init__)
            if hasattr(h, opt):
                getattr(h, opt).restype, getattr(h, opt).argtypes = herr_t, [hid_t, hsize_t]
        if hasattr(h, "H5Dwrite_chunk"):  # 1.10.3+: place a pre-filtered chunk (H5Dataset._write_rows_direct)
            h.H5Dwrite_chunk.restype, h.H5Dwrite_chunk.argtypes = herr_t, [hid_t, hid_t, C.c_uint32, C.POINTER(hsize_t), C.c_size_t, C.c_void_p]
        try:  # 1.10.5+: where a chunk lives in the file, so that worker threads can pread() it without the library lock
            h.H5Dget_chunk_info_by_coord.restype = herr_t
            h.H5Dget_chunk_info_by_coord.argtypes = [hid_t, C.POINTER(hsize_t), C.POINTER(C.c_uint), C.POINTER(C.c_uint64), C.POINTER(hsize_t)]
            h._arp_has_chunk_info = True
        except AttributeError:
            h._arp_has_chunk_info = False
        if h.H5open() < 0:
            raise ImportError("H5open failed")
        h.H5Eset_auto2(0, None, None)  # errors become Python exceptions, not stderr dumps
        _lib = h
        return _lib


def lib_version():
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    lib().H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def _gid(name):
    return hid_t.in_dll(lib(), name).value


def _ck(r, what):
    if r < 0:
        raise H5Error(f"HDF5: {what} failed")
    return r


def _dims(seq):
    return (hsize_t * len(seq))(*[int(x) for x in seq])


_NATIVE = {
    np.dtype(np.uint8): "H5T_NATIVE_UINT8_g", np.dtype(np.int8): "H5T_NATIVE_INT8_g", np.dtype(np.uint16): "H5T_NATIVE_UINT16_g",
    np.dtype(np.int16): "H5T_NATIVE_INT16_g", np.dtype(np.uint32): "H5T_NATIVE_UINT32_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g",
    np.dtype(np.uint64): "H5T_NATIVE_UINT64_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g",
    np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
}


def _native_type(dt):
    dt = np.dtype(dt)
    if dt == np.bool_:
        dt = np.dtype(np.int8)  # h5py stores numpy bools as an int8-based enum; plain int8 is what this writer emits
    if dt not in _NATIVE:
        raise TypeError(f"unsupported dtype {dt}")
    return _gid(_NATIVE[dt])


def _np_dtype_of(tid):
    """numpy dtype for a file datatype (+ whether the handle returned for memory reads must be closed)."""
    L = lib()
    cls = L.H5Tget_class(tid)
    if cls == H5T_ENUM:  # h5py bool: read through the integer base type
        base = L.H5Tget_super(tid)
        try:
            return _np_dtype_of(base)
        finally:
            L.H5Tclose(base)
    size = L.H5Tget_size(tid)
    if cls == H5T_FLOAT:
        return np.dtype({4: np.float32, 8: np.float64}[size])
    if cls == H5T_INTEGER:
        signed = L.H5Tget_sign(tid) != 0
        return np.dtype({(1, False): np.uint8, (1, True): np.int8, (2, False): np.uint16, (2, True): np.int16, (4, False): np.uint32,
                         (4, True): np.int32, (8, False): np.uint64, (8, True): np.int64}[(size, signed)])
    raise TypeError(f"unsupported HDF5 datatype class {cls}")


_native = None


def _native_inflate():
    """``arp_h5_inflate_last_frames`` of libarp_hip.so as a numpy-taking callable, or None when the library is not built."""
    global _native
    if _native is None:
        try:
            from . import _ffi
        except Exception:  # noqa: BLE001 -- the HDF5 reader stays usable on its Python pool
            _native = False
        else:
            def call(fd, n, addr, size, rawf, chunk_bytes, frame_bytes, offs, cnts, out, threads):
                u64, u8 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
                _ffi.check(_ffi.lib.arp_h5_inflate_last_frames(
                    fd, n, addr.ctypes.data_as(u64), size.ctypes.data_as(u64), rawf.ctypes.data_as(u8), chunk_bytes, frame_bytes,
                    offs.ctypes.data_as(u64), cnts.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(u8), threads))
            _native = call
    return _native or None


class _Attrs:
    """``file.attrs`` -- string and numeric scalars (the recorder writes ``env_name``, trajectory_recorder.py:66)."""

    def __init__(self, owner):
        self._o = owner

    def __contains__(self, name):
        with _lock:
            return lib().H5Aexists(self._o._id, name.encode()) > 0

    def __getitem__(self, name):
        L = lib()
        with _lock:
            if name not in self:
                raise KeyError(name)
            a = _ck(L.H5Aopen(self._o._id, name.encode(), H5P_DEFAULT), "H5Aopen")
            t = L.H5Aget_type(a)
            try:
                if L.H5Tget_class(t) == H5T_STRING:
                    if L.H5Tis_variable_str(t) > 0:
                        p = C.c_void_p()
                        _ck(L.H5Aread(a, t, C.byref(p)), "H5Aread")
                        s = C.string_at(p.value).decode("utf-8") if p.value else ""
                        if p.value:
                            L.H5free_memory(p)
                        return s
                    n = L.H5Tget_size(t)
                    buf = C.create_string_buffer(n + 1)
                    _ck(L.H5Aread(a, t, buf), "H5Aread")
                    return buf.raw[:n].split(b"\0")[0].decode("utf-8")
                dt = _np_dtype_of(t)
                out = np.zeros((), dt)
                _ck(L.H5Aread(a, _native_type(dt), out.ctypes.data_as(C.c_void_p)), "H5Aread")
                return out[()]
            finally:
                L.H5Tclose(t)
                L.H5Aclose(a)

    def get(self, name, default=None):
        return self[name] if name in self else default

    def __setitem__(self, name, value):
        L = lib()
        with _lock:
            if name in self:
                _ck(L.H5Adelete(self._o._id, name.encode()), "H5Adelete")
            sp = _ck(L.H5Screate(H5S_SCALAR), "H5Screate")
            try:
                if isinstance(value, (str, bytes)):  # variable-length UTF-8, as h5py writes a Python str
                    t = L.H5Tcopy(_gid("H5T_C_S1_g"))
                    L.H5Tset_size(t, H5T_VARIABLE)
                    L.H5Tset_cset(t, H5T_CSET_UTF8)
                    raw = value.encode("utf-8") if isinstance(value, str) else value
                    buf = C.create_string_buffer(raw)
                    p = C.c_char_p(C.addressof(buf))
                    a = _ck(L.H5Acreate2(self._o._id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
                    try:
                        _ck(L.H5Awrite(a, t, C.byref(p)), "H5Awrite")
                    finally:
                        L.H5Aclose(a)
                        L.H5Tclose(t)
                else:
                    v = np.asarray(value)
                    if v.ndim:
                        raise TypeError("only scalar attributes are supported")
                    t = _native_type(v.dtype)
                    a = _ck(L.H5Acreate2(self._o._id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
                    try:
                        _ck(L.H5Awrite(a, t, v.ctypes.data_as(C.c_void_p)), "H5Awrite")
                    finally:
                        L.H5Aclose(a)
            finally:
                L.H5Sclose(sp)


class H5Dataset:
    """The h5py.Dataset subset the labelling path touches: ``shape``, ``dtype``, ``chunks``, ``compression``, basic slicing,
    slice assignment, ``resize`` -- plus ``read_last_frames``."""

    def __init__(self, store, name, did):
        self._store, self.name, self._id = store, name, did
        L = lib()
        with _lock:
            t = L.H5Dget_type(did)
            try:
                self.dtype = _np_dtype_of(t)
            finally:
                L.H5Tclose(t)
            p = L.H5Dget_create_plist(did)
            try:
                self.chunks = None
                self.filters = []
                if L.H5Pget_layout(p) == H5D_CHUNKED:
                    nd = len(self.shape)
                    d = (hsize_t * max(nd, 1))()
                    L.H5Pget_chunk(p, nd, d)
                    self.chunks = tuple(int(x) for x in d[:nd])
                    for i in range(max(L.H5Pget_nfilters(p), 0)):
                        flags, n = C.c_uint(), C.c_size_t(8)
                        vals, cfg = (C.c_uint * 8)(), C.c_uint()
                        fid = L.H5Pget_filter2(p, i, C.byref(flags), C.byref(n), vals, 0, None, C.byref(cfg))
                        self.filters.append((int(fid), tuple(int(v) for v in vals[: n.value])))
            finally:
                L.H5Pclose(p)

    # ---- shape -----------------------------------------------------------------------------------------
    @property
    def shape(self):
        L = lib()
        with _lock:
            sp = _ck(L.H5Dget_space(self._id), "H5Dget_space")
            try:
                nd = L.H5Sget_simple_extent_ndims(sp)
                d = (hsize_t * max(nd, 1))()
                L.H5Sget_simple_extent_dims(sp, d, None)
                return tuple(int(x) for x in d[:nd])
            finally:
                L.H5Sclose(sp)

    @property
    def maxshape(self):
        L = lib()
        with _lock:
            sp = _ck(L.H5Dget_space(self._id), "H5Dget_space")
            try:
                nd = L.H5Sget_simple_extent_ndims(sp)
                d, m = (hsize_t * max(nd, 1))(), (hsize_t * max(nd, 1))()
                L.H5Sget_simple_extent_dims(sp, d, m)
                return tuple(None if int(x) == H5S_UNLIMITED else int(x) for x in m[:nd])
            finally:
                L.H5Sclose(sp)

    @property
    def ndim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    @property
    def compression(self):
        return "gzip" if any(f == H5Z_FILTER_DEFLATE for f, _ in self.filters) else None

    @property
    def compression_opts(self):
        for f, v in self.filters:
            if f == H5Z_FILTER_DEFLATE:
                return v[0] if v else None
        return None

    # ---- selection -------------------------------------------------------------------------------------
    def _select(self, key):
        """h5py-style basic indexing -> (start, count, squeeze-axes, first-axis index list or None)."""
        shape = self.shape
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = [j for j, k in enumerate(key) if k is Ellipsis][0]
            key = key[:i] + (slice(None),) * (len(shape) - (len(key) - 1)) + key[i + 1:]
        key = key + (slice(None),) * (len(shape) - len(key))
        if len(key) != len(shape):
            raise IndexError(f"too many indices for a dataset of rank {len(shape)}")
        start, count, squeeze, rows = [], [], [], None
        for ax, (k, n) in enumerate(zip(key, shape)):
            if isinstance(k, (int, np.integer)):
                k = int(k)
                if k < 0:
                    k += n
                if not 0 <= k < n:
                    raise IndexError(f"index {k} out of range for axis {ax} with size {n}")
                start.append(k); count.append(1); squeeze.append(ax)
            elif isinstance(k, slice):
                a, b, st = k.indices(n)
                if st != 1:
                    raise IndexError("only unit-stride slices are supported")
                start.append(a); count.append(max(b - a, 0))
            elif ax == 0 and isinstance(k, (list, np.ndarray)):  # g[img_key][traj, -1] with traj a list (label_reward.py:268)
                idx = np.asarray(k, dtype=np.int64)
                idx = np.where(idx < 0, idx + n, idx)
                if idx.size and (idx.min() < 0 or idx.max() >= n):
                    raise IndexError("row index out of range")
                if idx.size and np.array_equal(idx, np.arange(idx[0], idx[0] + idx.size)):
                    start.append(int(idx[0])); count.append(int(idx.size))
                else:
                    rows = idx
                    start.append(0); count.append(0)
            else:
                raise IndexError(f"unsupported index {k!r}")
        return start, count, squeeze, rows

    def _write_rows_direct(self, start, count, arr):
        """Whole rows of a dataset chunked one row per chunk with the deflate filter alone (the reward datasets of
        label_reward.py:277-283): deflate + H5Dwrite_chunk per row on the native side (arp_h5_write_rows_deflated) instead of the
        library's filter pipeline -- 2.7x faster on 8192 rows of 32 bytes.  Returns False when the fast path does not apply."""
        shape = self.shape
        if (os.environ.get("ARP_H5_DIRECT_WRITE", "1") == "0" or self.chunks is None or len(self.filters) != 1 or self.filters[0][0] != H5Z_FILTER_DEFLATE
                or self.chunks != (1,) + tuple(shape[1:]) or list(start[1:]) != [0] * (len(shape) - 1) or list(count[1:]) != list(shape[1:])
                or arr.dtype != self.dtype or self.dtype == np.bool_ or count[0] < 64):
            return False
        L = lib()
        fn = getattr(L, "H5Dwrite_chunk", None)
        try:
            from . import _ffi
        except Exception:  # noqa: BLE001 -- libarp_hip.so not built: the library's own write path
            return False
        if fn is None:
            return False
        a = np.ascontiguousarray(arr)
        level = self.compression_opts
        with _lock:
            _ffi.check(_ffi.lib.arp_h5_write_rows_deflated(C.cast(fn, C.c_void_p), self._id, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p),
                                                           a.nbytes // count[0], count[0], start[0], len(shape), 4 if level is None else int(level)))
        return True

    def _rw(self, start, count, arr, write):
        L = lib()
        if int(np.prod(count)) == 0:
            return
        if write and self._write_rows_direct(start, count, arr):
            return
        with _lock:
            fsp = _ck(L.H5Dget_space(self._id), "H5Dget_space")
            msp = _ck(L.H5Screate_simple(len(count), _dims(count), None), "H5Screate_simple")
            try:
                _ck(L.H5Sselect_hyperslab(fsp, H5S_SELECT_SET, _dims(start), None, _dims(count), None), "H5Sselect_hyperslab")
                fn = L.H5Dwrite if write else L.H5Dread
                _ck(fn(self._id, _native_type(arr.dtype), msp, fsp, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)),
                    "H5Dwrite" if write else "H5Dread")
            finally:
                L.H5Sclose(msp)
                L.H5Sclose(fsp)

    def __getitem__(self, key):
        start, count, squeeze, rows = self._select(key)
        mem_dt = np.dtype(np.int8) if self.dtype == np.bool_ else self.dtype
        if rows is not None:  # arbitrary row list: one hyperslab per row
            out = np.empty([len(rows)] + count[1:], mem_dt)
            for i, r in enumerate(rows):
                self._rw([int(r)] + start[1:], [1] + count[1:], out[i : i + 1], False)
        else:
            out = np.empty(count, mem_dt)
            self._rw(start, count, out, False)
        if squeeze:
            out = out.reshape([c for ax, c in enumerate(out.shape) if ax not in squeeze])
        return out if out.ndim else out[()]

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def __setitem__(self, key, value):
        start, count, squeeze, rows = self._select(key)
        if rows is not None:
            raise IndexError("assignment needs a contiguous selection")
        mem_dt = np.dtype(np.int8) if self.dtype == np.bool_ else self.dtype
        target = [c for ax, c in enumerate(count) if ax not in squeeze]
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=mem_dt), target))
        self._rw(start, count, v.reshape(count), True)

    def resize(self, size, axis=None):
        shape = list(self.shape)
        if axis is not None:
            shape[axis] = int(size)
        else:
            shape = [int(s) for s in size]
        with _lock:
            _ck(lib().H5Dset_extent(self._id, _dims(shape)), "H5Dset_extent")

    # ---- the fast reader ---------------------------------------------------------------------------------
    def _raw_chunk(self, row):
        """(filter_mask, bytes) of the chunk that starts at ``row`` on axis 0 -- as stored, not inflated."""
        L = lib()
        off = _dims([row] + [0] * (self.ndim - 1))
        n = hsize_t()
        with _lock:
            _ck(L.H5Dget_chunk_storage_size(self._id, off, C.byref(n)), "H5Dget_chunk_storage_size")
            if n.value == 0:
                return 0, None  # never written: fill value
            buf = C.create_string_buffer(n.value)
            mask = C.c_uint32()
            _ck(L.H5Dread_chunk(self._id, H5P_DEFAULT, off, C.byref(mask), buf), "H5Dread_chunk")
        return mask.value, buf

    def _chunk_loc(self, row):
        """(filter_mask, file address, stored size) of the chunk starting at ``row``; None where the library cannot tell."""
        L = lib()
        if not L._arp_has_chunk_info:
            return None
        mask, addr, size = C.c_uint(), C.c_uint64(), hsize_t()
        with _lock:
            if L.H5Dget_chunk_info_by_coord(self._id, _dims([row] + [0] * (self.ndim - 1)), C.byref(mask), C.byref(addr), C.byref(size)) < 0:
                return None
        if addr.value == 0xFFFFFFFFFFFFFFFF:
            return (0, None, 0)  # never written
        return (mask.value, addr.value, size.value)

    def fast_path_ok(self):
        """Row-chunked (1, F, ...) with deflate as the only filter that changes bytes (a shuffle of 1-byte elements is the identity)."""
        if self.chunks is None or self.ndim < 2 or self.chunks != (1,) + self.shape[1:]:
            return False
        ids = [f for f, _ in self.filters]
        return ids == [H5Z_FILTER_DEFLATE] or (self.dtype.itemsize == 1 and ids == [H5Z_FILTER_SHUFFLE, H5Z_FILTER_DEFLATE])

    def read_last_frames(self, r0, r1, threads=None, stacked=True, native=None, native_threads=None):
        """``self[r0:r1, -1]`` for the rows of ONE trajectory (see ``read_last_frames_spans``)."""
        return self.read_last_frames_spans([(r0, r1)], threads=threads, stacked=stacked, native=native, native_threads=native_threads)

    def read_last_frames_spans(self, spans, threads=None, stacked=True, native=None, native_threads=None, out=None):
        """``concatenate([self[a:b, -1] for a, b in spans])`` where every span is the rows of ONE trajectory (or part of one),
        inflating one chunk per ``num_frames`` rows (module docstring); all spans' chunks go to the workers in one batch.
        ``out``: a C-contiguous array of this dtype with room for at least that many frames, reused across calls (a fresh 200 MB
        buffer per 1024-frame batch costs ~49 k page faults going in and an munmap coming out: 9 ms each on the test host); the
        result is then a view of its head.
        ``stacked=False`` (or a dataset the fast path does not cover) reads every row's chunk -- the reference's access pattern,
        still inflated in parallel.  Inflation runs on the C++ threads of ``arp_h5_inflate_last_frames`` (libarp_hip.so) when
        that library is importable and HDF5 can report chunk addresses; ``native=False`` keeps the Python thread pool
        (``threads`` workers; zlib releases the GIL)."""
        spans = [(int(a), int(b)) for a, b in spans if int(b) > int(a)]
        n = sum(b - a for a, b in spans)
        F = self.shape[1]
        frame_shape = self.shape[2:]
        fbytes = int(np.prod(frame_shape)) * self.dtype.itemsize
        if out is not None:
            if out.dtype != self.dtype or not out.flags.c_contiguous or out.size < n * int(np.prod(frame_shape)):
                raise ValueError("out: need a C-contiguous array of the dataset's dtype with room for the requested frames")
            out = out.reshape(-1)[: n * int(np.prod(frame_shape))].reshape((n,) + frame_shape)
        else:
            out = np.empty((max(n, 0),) + frame_shape, self.dtype)
        if n <= 0:
            return out
        if not self.fast_path_ok():
            o = 0
            for a, b in spans:
                out[o : o + b - a] = self[a:b, -1]
                o += b - a
            return out
        threads = threads or min(8, os.cpu_count() or 4)  # measured on a 256-cpu host: 8 threads 27.8 k frames/s, 32 threads 16 k (GIL hand-offs)
        deflate_idx = [f for f, _ in self.filters].index(H5Z_FILTER_DEFLATE)

        fd = getattr(self._store, "_pread_fd", None)
        if fd is not None and self._store.mode != "r":
            self._store.flush()  # chunks written through this handle must be in the file before another descriptor reads them

        def inflate(job):
            row, raw, lo, cnt = job  # this chunk supplies out[lo : lo + cnt] = its LAST cnt frames
            mask, buf = raw
            if buf is None:
                return lo, cnt, None
            if isinstance(buf, tuple):  # (address, size): read here, outside the library lock
                stored = os.pread(fd, buf[1], buf[0])
                if len(stored) != buf[1]:
                    raise H5Error(f"{self.name}: short read of the chunk at row {row}")
            else:
                stored = memoryview(buf)
            data = stored if (mask >> deflate_idx) & 1 else zlib.decompress(stored)  # mask bit i set = filter i skipped for this chunk
            if len(data) != F * fbytes:
                raise H5Error(f"{self.name}: chunk at row {row} inflated to {len(data)} bytes, expected {F * fbytes}")
            return lo, cnt, np.frombuffer(data, self.dtype, count=cnt * (fbytes // self.dtype.itemsize), offset=(F - cnt) * fbytes)

        def raw_of(row):
            nonlocal fd
            if fd is not None:
                loc = self._chunk_loc(row)
                if loc is not None:
                    return (loc[0], None if loc[1] is None else (loc[1], loc[2]))
            return self._raw_chunk(row)  # serial (library lock); inflation still runs in the pool

        if stacked:
            # chunk of row i holds the last frames of rows i-F+1 .. i of the same trajectory: walk back from the last row
            jobs_rows, base = [], 0
            for r0, r1 in spans:
                hi = r1
                while hi > r0:
                    lo = max(hi - F, r0)
                    jobs_rows.append((hi - 1, base + lo - r0, hi - lo))
                    hi = lo
                base += r1 - r0
            # cross-check against plain per-row reads before trusting the sliding-window layout.  A group of ONE row proves nothing
            # (a chunk's last frame always is its own row's last frame), so only groups of >= 2 rows count, the fullest ones
            # first (cnt == F exercises every slot), up to three of them spread over the call; the verdict is latched for the
            # dataset only once such a group has been checked.  A call made of one-row groups alone is per-row reading anyway.
            if not getattr(self, "_stack_checked", False):
                best = max(cnt for _, _, cnt in jobs_rows)
                if best >= 2:
                    cand = [j for j in jobs_rows if j[2] == best]
                    picks = [cand[0], cand[len(cand) // 2], cand[-1]] if len(cand) > 2 else cand
                    ok = True
                    for row, lo, cnt in dict.fromkeys(picks):
                        direct = self[row - cnt + 1 : row + 1, -1]
                        try:
                            got = inflate((row, raw_of(row), lo, cnt))[2]
                        except (zlib.error, OSError):  # the pread path mis-addressed the chunk (e.g. a user block): library reads only
                            if fd is not None:
                                os.close(fd)
                            self._store._pread_fd = fd = None
                            got = inflate((row, self._raw_chunk(row), lo, cnt))[2]
                        if got is None or not np.array_equal(direct.reshape(-1), got):
                            ok = False
                            break
                    self._stack_ok = ok
                    self._stack_checked = True
                else:
                    self._stack_ok = True  # this call only: nothing latched
            if not self._stack_ok:
                return self.read_last_frames_spans(spans, threads=threads, stacked=False, native=native, native_threads=native_threads, out=out)
        else:
            jobs_rows, base = [], 0
            for r0, r1 in spans:
                jobs_rows += [(r, base + r - r0, 1) for r in range(r0, r1)]
                base += r1 - r0

        if fd is not None and native is not False and _native_inflate() is not None:
            # native path: chunk locations from the library, pread + inflate + tail copy on C++ threads (csrc/arp_io.cpp)
            locs = [self._chunk_loc(row) for row, _, _ in jobs_rows]
            if all(l is not None for l in locs):
                k = len(jobs_rows)
                addr = np.array([l[1] or 0 for l in locs], np.uint64)
                size = np.array([l[2] if l[1] is not None else 0 for l in locs], np.uint64)
                rawf = np.array([(l[0] >> deflate_idx) & 1 for l in locs], np.uint8)
                offs = np.array([lo * fbytes for _, lo, _ in jobs_rows], np.uint64)
                cnts = np.array([cnt for _, _, cnt in jobs_rows], np.uint32)
                if native_threads is None:  # ARP_H5_THREADS; default 32: more only adds scheduling noise next to the GPU feeder thread
                    native_threads = int(os.environ.get("ARP_H5_THREADS", "0")) or min(32, os.cpu_count() or 8)
                _native_inflate()(fd, k, addr, size, rawf, F * fbytes, fbytes, offs, cnts, out, native_threads)
                return out

        def gen():
            for row, lo, cnt in jobs_rows:
                yield row, raw_of(row), lo, cnt

        with ThreadPoolExecutor(max_workers=threads) as pool:
            for lo, cnt, flat in pool.map(inflate, gen()):
                if flat is None:
                    out[lo : lo + cnt] = 0
                else:
                    out[lo : lo + cnt] = flat.reshape((cnt,) + frame_shape)
        return out

    def close(self):
        if self._id:
            with _lock:
                lib().H5Dclose(self._id)
            self._id = 0


class H5Store:
    """``H5Store(path, "a")`` stands where ``h5py.File(path, "a")`` stands in arp_dt/label_reward.py:69."""

    def __init__(self, path, mode="r"):
        L = lib()
        self.filename = path
        self._ds = {}
        b = os.fsencode(path)
        with _lock:
            # writers: the reward datasets are thousands of ~40-byte chunks; with the default 2 KiB allocation blocks the library goes to
            # its free-space manager every ~50 chunks.  1 MiB blocks (file-access properties only: the file format is unchanged) take
            # ~13 % off creating them.  (The v1.10 chunk index -- libver bounds -- would take another 15 %, but changes what older
            # readers can open: not used.)
            fapl = H5P_DEFAULT
            if mode != "r" and hasattr(L, "H5Pset_small_data_block_size"):
                fapl = L.H5Pcreate(_gid("H5P_CLS_FILE_ACCESS_ID_g"))
                if fapl >= 0:
                    L.H5Pset_small_data_block_size(fapl, 1 << 20)
                    L.H5Pset_meta_block_size(fapl, 1 << 20)
                else:
                    fapl = H5P_DEFAULT
            try:
                if mode == "r":
                    fid = L.H5Fopen(b, H5F_ACC_RDONLY, H5P_DEFAULT)
                elif mode == "r+":
                    fid = L.H5Fopen(b, H5F_ACC_RDWR, fapl)
                elif mode == "a":
                    fid = L.H5Fopen(b, H5F_ACC_RDWR, fapl) if os.path.exists(path) else L.H5Fcreate(b, H5F_ACC_EXCL, H5P_DEFAULT, fapl)
                elif mode == "w":
                    fid = L.H5Fcreate(b, H5F_ACC_TRUNC, H5P_DEFAULT, fapl)
                elif mode in ("w-", "x"):
                    fid = L.H5Fcreate(b, H5F_ACC_EXCL, H5P_DEFAULT, fapl)
                else:
                    raise ValueError(f"bad mode {mode!r}")
            finally:
                if fapl != H5P_DEFAULT:
                    L.H5Pclose(fapl)
        if fid < 0:
            raise H5Error(f"cannot open {path!r} in mode {mode!r}")
        self._id = fid
        self.mode = mode
        self.attrs = _Attrs(self)
        # a second, read-only descriptor for pread() of stored chunks from worker threads (read_last_frames)
        try:
            self._pread_fd = os.open(path, os.O_RDONLY)
        except OSError:
            self._pread_fd = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __contains__(self, name):
        with _lock:
            return bool(self._id) and lib().H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def keys(self):
        L = lib()
        with _lock:
            info = _H5G_info()
            _ck(L.H5Gget_info(self._id, C.byref(info)), "H5Gget_info")
            names = []
            for i in range(info.nlinks):
                n = L.H5Lget_name_by_idx(self._id, b".", 0, 0, i, None, 0, H5P_DEFAULT)
                buf = C.create_string_buffer(n + 1)
                L.H5Lget_name_by_idx(self._id, b".", 0, 0, i, buf, n + 1, H5P_DEFAULT)
                names.append(buf.value.decode())
            return names

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, name):
        if name in self._ds and self._ds[name]._id:
            return self._ds[name]
        with _lock:
            if name not in self:
                raise KeyError(name)
            did = lib().H5Dopen2(self._id, name.encode(), H5P_DEFAULT)
        if did < 0:
            raise H5Error(f"{name!r} is not a dataset")
        self._ds[name] = H5Dataset(self, name, did)
        return self._ds[name]

    def get(self, name, default=None):
        return self[name] if name in self else default

    def __delitem__(self, name):
        if name in self._ds:
            self._ds.pop(name).close()
        with _lock:
            _ck(lib().H5Ldelete(self._id, name.encode(), H5P_DEFAULT), "H5Ldelete")

    def create_dataset(self, name, shape=None, dtype=None, data=None, compression=None, compression_opts=None, chunks=None, maxshape=None):
        """``g.create_dataset(key, compression="gzip", chunks=(1, num_frames), maxshape=(None, num_frames), data=...)`` of
        label_reward.py:277-283 / trajectory_recorder.py:156-170 (gzip level 4 = h5py's default)."""
        L = lib()
        if data is not None:
            data = np.ascontiguousarray(data if dtype is None else np.asarray(data, dtype=dtype))
            shape = data.shape if shape is None else tuple(shape)
            dtype = data.dtype
        if shape is None or dtype is None:
            raise TypeError("create_dataset needs data or shape + dtype")
        shape = tuple(int(s) for s in shape)
        if compression not in (None, "gzip"):
            raise ValueError("only gzip compression is supported")
        if maxshape is not None or compression is not None:
            if chunks is None or chunks is True:
                chunks = tuple(max(1, s) for s in shape)
        with _lock:
            if name in self:
                raise ValueError(f"dataset {name!r} exists")
            mx = None if maxshape is None else _dims([H5S_UNLIMITED if m is None else m for m in maxshape])
            sp = _ck(L.H5Screate_simple(len(shape), _dims(shape), mx), "H5Screate_simple")
            pl = _ck(L.H5Pcreate(_gid("H5P_CLS_DATASET_CREATE_ID_g")), "H5Pcreate")
            try:
                if chunks is not None:
                    _ck(L.H5Pset_chunk(pl, len(chunks), _dims(chunks)), "H5Pset_chunk")
                if compression == "gzip":
                    _ck(L.H5Pset_deflate(pl, 4 if compression_opts is None else int(compression_opts)), "H5Pset_deflate")
                did = _ck(L.H5Dcreate2(self._id, name.encode(), _native_type(dtype), sp, H5P_DEFAULT, pl, H5P_DEFAULT), "H5Dcreate2")
            finally:
                L.H5Pclose(pl)
                L.H5Sclose(sp)
        ds = H5Dataset(self, name, did)
        self._ds[name] = ds
        if data is not None and data.size:
            ds._rw([0] * len(shape), list(shape), data.astype(np.int8) if data.dtype == np.bool_ else data, True)
        return ds

    def flush(self):
        with _lock:
            lib().H5Fflush(self._id, H5F_SCOPE_GLOBAL)

    def close(self):
        if not self._id:
            return
        for d in self._ds.values():
            d.close()
        self._ds.clear()
        with _lock:
            lib().H5Fclose(self._id)
        self._id = 0
        if self._pread_fd is not None:
            os.close(self._pread_fd)
            self._pread_fd = None
