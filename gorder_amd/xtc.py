"""ctypes binding of the host-side XTC reader (include/gorder_xtc.h, gorder_amd/csrc/xtc_reader.cpp)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from .abi import load_library

_bound = False


def _lib():
    global _bound
    lib = load_library()
    if not _bound:
        vp = C.c_void_p
        lib.gorder_xtc_open.argtypes = [C.c_char_p, vp, C.c_uint32, C.POINTER(vp)]
        lib.gorder_xtc_close.argtypes = [vp]
        lib.gorder_xtc_close.restype = None
        lib.gorder_xtc_n_atoms_file.argtypes = [vp]
        lib.gorder_xtc_n_atoms_file.restype = C.c_uint32
        lib.gorder_xtc_n_atoms_out.argtypes = [vp]
        lib.gorder_xtc_n_atoms_out.restype = C.c_uint32
        lib.gorder_xtc_next.argtypes = [vp, vp, vp, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.gorder_xtc_read_window.argtypes = [vp, C.c_float, C.c_float, C.c_uint32, C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_double), vp, vp, vp, C.c_uint64]
        lib.gorder_xtc_read_window.restype = C.c_int64
        lib.gorder_xtc_read_window_mt.argtypes = [vp, C.c_float, C.c_float, C.c_uint32, C.POINTER(C.c_uint64),
                                                  C.POINTER(C.c_double), vp, vp, vp, C.c_uint64, C.c_uint32]
        lib.gorder_xtc_read_window_mt.restype = C.c_int64
        lib.gorder_xtc_pack_window.argtypes = [vp, C.c_float, C.c_float, C.c_uint32, C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_double), vp, C.c_uint64, C.POINTER(C.c_uint64), vp, vp, vp,
                                               C.c_uint64, C.c_uint32]
        lib.gorder_xtc_pack_window.restype = C.c_int64
        lib.gorder_xtc_skip_window.argtypes = [vp, C.c_float, C.c_float, C.c_uint32, C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_double), C.c_uint64]
        lib.gorder_xtc_skip_window.restype = C.c_int64
        lib.gorder_xtc_pool_create.argtypes = [C.c_uint32, C.POINTER(vp)]
        lib.gorder_xtc_pool_wait.argtypes = [vp]
        lib.gorder_xtc_pool_destroy.argtypes = [vp]
        lib.gorder_xtc_pool_destroy.restype = None
        lib.gorder_xtc_pack_window_pool.argtypes = [vp, C.c_float, C.c_float, C.c_uint32, C.POINTER(C.c_uint64),
                                                    C.POINTER(C.c_double), vp, C.c_uint64, C.POINTER(C.c_uint64), vp, vp, vp,
                                                    C.c_uint64, vp]
        lib.gorder_xtc_pack_window_pool.restype = C.c_int64
        lib.gorder_xtc_pack_window_ex.argtypes = [vp, C.c_float, C.c_float, C.c_uint32, C.POINTER(C.c_uint64),
                                                  C.POINTER(C.c_double), vp, C.c_uint64, C.POINTER(C.c_uint64), vp, vp, vp,
                                                  C.c_uint64, C.c_uint32, vp, C.c_uint32, vp]
        lib.gorder_xtc_pack_window_ex.restype = C.c_int64
        lib.gorder_xtc_is_xtc.argtypes = [vp]
        lib.gorder_xtc_probe.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        lib.gorder_xtc_n_atoms_needed.argtypes = [vp]
        lib.gorder_xtc_n_atoms_needed.restype = C.c_uint32
        lib.gorder_xtc_writer_open.argtypes = [C.c_char_p, C.c_uint32, C.c_float, C.POINTER(vp)]
        lib.gorder_xtc_writer_add.argtypes = [vp, vp, vp, C.c_int64, C.c_float]
        lib.gorder_xtc_writer_close.argtypes = [vp]
        lib.gorder_xtc_writer_close.restype = None
        _bound = True
    return lib


def write_trajectory(path: str, xyz: np.ndarray, box: np.ndarray, times: Optional[Sequence[float]] = None,
                     precision: float = 1000.0, append_to=None):
    """Write frames [F, N, 3] (+ boxes [F, 3, 3], times in ps) as a compressed XTC file (tooling: tests, end-to-end
    benchmark).  Reading it back gives round(x * precision) / precision."""
    lib = _lib()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    box = np.ascontiguousarray(box, dtype=np.float32).reshape(xyz.shape[0], 9)
    if times is None:
        times = np.arange(xyz.shape[0], dtype=np.float32)
    w = C.c_void_p()
    st = lib.gorder_xtc_writer_open(path.encode(), xyz.shape[1], precision, C.byref(w))
    if st != 0:
        raise IOError(f"cannot create {path}: status {st}")
    try:
        for f in range(xyz.shape[0]):
            st = lib.gorder_xtc_writer_add(w, xyz[f].ctypes.data, box[f].ctypes.data, f, float(times[f]))
            if st != 0:
                raise IOError(f"{path}: XTC write error {st} in frame {f}")
    finally:
        lib.gorder_xtc_writer_close(w)


def read_trajectory(paths: Sequence[str], group: Optional[np.ndarray] = None, begin: float = 0.0, end: float = -1.0,
                    step: int = 1, chunk: int = 64, return_precision: bool = False, threads: int = 1):
    """Read (and concatenate) XTC files like gorder's `read_trajectory`
    (/root/reference/src/analysis/common.rs:239-342): time window in ps, every `step`-th frame,
    duplicate boundary frames of consecutive files dropped.
    -> xyz [F, n_out, 3] f32, box [F, 3, 3] f32, time [F] f32."""
    lib = _lib()
    grp = None if group is None else np.ascontiguousarray(group, dtype=np.uint32)
    state, last = C.c_uint64(0), C.c_double(float("-inf"))
    xs, bs, ts = [], [], []
    prec = 0.0
    for path in paths:
        r = C.c_void_p()
        st = lib.gorder_xtc_open(path.encode(), None if grp is None else grp.ctypes.data, 0 if grp is None else grp.size,
                                 C.byref(r))
        if st != 0:
            raise IOError(f"cannot open {path}: status {st}")
        try:
            n = lib.gorder_xtc_n_atoms_out(r)
            if return_precision and prec == 0.0:
                p, t = C.c_float(), C.c_float()
                lib.gorder_xtc_next(r, None, None, None, C.byref(t), C.byref(p))
                prec = p.value
                lib.gorder_xtc_close(r)
                r = C.c_void_p()
                lib.gorder_xtc_open(path.encode(), None if grp is None else grp.ctypes.data,
                                    0 if grp is None else grp.size, C.byref(r))
            while True:
                x = np.empty((chunk, n, 3), dtype=np.float32)
                b = np.empty((chunk, 3, 3), dtype=np.float32)
                t = np.empty(chunk, dtype=np.float32)
                got = lib.gorder_xtc_read_window_mt(r, begin, end, step, C.byref(state), C.byref(last), x.ctypes.data,
                                                    b.ctypes.data, t.ctypes.data, chunk, threads)
                if got < 0:
                    raise IOError(f"{path}: XTC read error {got}")
                if got == 0:
                    break
                xs.append(x[:got]); bs.append(b[:got]); ts.append(t[:got])
        finally:
            lib.gorder_xtc_close(r)
    if not xs:
        n = 0 if grp is None else grp.size
        out = (np.zeros((0, n, 3), np.float32), np.zeros((0, 3, 3), np.float32), np.zeros(0, np.float32))
    else:
        out = (np.concatenate(xs), np.concatenate(bs), np.concatenate(ts))
    return out + (prec,) if return_precision else out


def pack_trajectory(paths: Sequence[str], group: Optional[np.ndarray] = None, begin: float = 0.0, end: float = -1.0,
                    step: int = 1, chunk: int = 64, blob_capacity: int = 0, threads: int = 1, pool: bool = False):
    """The frames `read_trajectory` would return, still compressed: windows of at most `chunk` frames as
    gorder_xtc_pack_window packs them for the device decoder (gorder_hip_xtc_decode).
    `pool`: the block copies go through a pool of `threads` copying threads (gorder_xtc_pack_window_pool), waited for
    after every window.
    -> list of dicts {blob: uint8 [bytes], frames: structured array of CXtcFrame, box [n, 3, 3], time [n],
                      n_atoms_file, n_stop, slot_of (int32 [n_atoms_file] or None)}."""
    from .abi import CXtcFrame
    lib = _lib()
    grp = None if group is None else np.ascontiguousarray(group, dtype=np.uint32)
    state, last = C.c_uint64(0), C.c_double(float("-inf"))
    out = []
    cpool = C.c_void_p()
    if pool and lib.gorder_xtc_pool_create(threads, C.byref(cpool)) != 0:
        raise IOError("cannot create the copy pool")
    try:
        return _pack_files(lib, paths, grp, begin, end, step, chunk, blob_capacity, threads, cpool if pool else None, state,
                           last, out)
    finally:
        if pool:
            lib.gorder_xtc_pool_destroy(cpool)


def _pack_files(lib, paths, grp, begin, end, step, chunk, blob_capacity, threads, cpool, state, last, out):
    from .abi import CXtcFrame
    for path in paths:
        r = C.c_void_p()
        st = lib.gorder_xtc_open(path.encode(), None if grp is None else grp.ctypes.data, 0 if grp is None else grp.size,
                                 C.byref(r))
        if st != 0:
            raise IOError(f"cannot open {path}: status {st}")
        try:
            if not lib.gorder_xtc_is_xtc(r):
                raise IOError(f"{path}: not an XTC file")
            n_file, n_stop = lib.gorder_xtc_n_atoms_file(r), lib.gorder_xtc_n_atoms_needed(r)
            slot_of = None
            if grp is not None:
                slot_of = np.full(n_file, -1, dtype=np.int32)
                slot_of[grp] = np.arange(grp.size, dtype=np.int32)
            cap = blob_capacity or (chunk * (n_file * 12 + 256) + 4096)
            while True:
                blob = np.empty(cap, dtype=np.uint8)
                frames = (CXtcFrame * chunk)()
                b = np.empty((chunk, 3, 3), dtype=np.float32)
                t = np.empty(chunk, dtype=np.float32)
                used = C.c_uint64(0)
                if cpool is not None:
                    got = lib.gorder_xtc_pack_window_pool(r, begin, end, step, C.byref(state), C.byref(last), blob.ctypes.data,
                                                          cap, C.byref(used), C.cast(frames, C.c_void_p), b.ctypes.data,
                                                          t.ctypes.data, chunk, cpool)
                    if got >= 0 and lib.gorder_xtc_pool_wait(cpool) != 0:
                        got = -2
                else:
                    got = lib.gorder_xtc_pack_window(r, begin, end, step, C.byref(state), C.byref(last), blob.ctypes.data,
                                                     cap, C.byref(used), C.cast(frames, C.c_void_p), b.ctypes.data,
                                                     t.ctypes.data, chunk, threads)
                if got < 0:
                    raise IOError(f"{path}: XTC pack error {got}")
                if got == 0:
                    break
                fr = np.frombuffer(frames, dtype=np.dtype(CXtcFrame))[:got].copy()
                out.append({"blob": blob[:used.value].copy(), "frames": fr, "box": b[:got].copy(), "time": t[:got].copy(),
                            "n_atoms_file": n_file, "n_stop": n_stop, "slot_of": slot_of})
        finally:
            lib.gorder_xtc_close(r)
    return out


def count_frames(paths: Sequence[str], begin: float = 0.0, end: float = -1.0, step: int = 1) -> int:
    """How many frames `read_trajectory` would return, from the headers alone (gorder_xtc_skip_window)."""
    lib = _lib()
    state, last = C.c_uint64(0), C.c_double(float("-inf"))
    total = 0
    for path in paths:
        r = C.c_void_p()
        st = lib.gorder_xtc_open(path.encode(), None, 0, C.byref(r))
        if st != 0:
            raise IOError(f"cannot open {path}: status {st}")
        try:
            n = lib.gorder_xtc_skip_window(r, begin, end, step, C.byref(state), C.byref(last), 2 ** 64 - 1)
            if n < 0:
                raise IOError(f"{path}: XTC read error {n}")
            total += n
        finally:
            lib.gorder_xtc_close(r)
    return total
