"""ctypes wrapper of oracle/libgorder_oracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by gorder_amd/.
Takes the same python-side ``Tables`` description as the HIP engine so parity tests are symmetric.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from gorder_amd.abi import OK, CTables, Results, Tables

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgorder_oracle.so")
TRIG_LIBM, TRIG_MIRROR, TRIG_DIRECT = 0, 1, 2
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    lib.gorder_oracle_create.argtypes = [C.POINTER(CTables), i32, i32, C.POINTER(vp)]
    lib.gorder_oracle_destroy.argtypes = [vp]
    lib.gorder_oracle_destroy.restype = None
    lib.gorder_oracle_n_accumulators.argtypes = [vp]
    lib.gorder_oracle_n_accumulators.restype = u32
    lib.gorder_oracle_ordermap_dims.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    lib.gorder_oracle_ordermap_dims.restype = u32
    lib.gorder_oracle_submit.argtypes = [vp, vp, vp, vp, u32]
    lib.gorder_oracle_submit_passes.argtypes = [vp, vp, vp, vp, u32, u32]
    lib.gorder_oracle_ua_fast_fidelity.argtypes = [vp, vp, vp, u32, vp, vp]
    lib.gorder_oracle_prime_leaflets.argtypes = [vp, vp, vp, u64]
    lib.gorder_oracle_set_manual_leaflets.argtypes = [vp, vp, u64]
    lib.gorder_oracle_finish.argtypes = [vp, vp, vp, vp, vp, C.POINTER(u64)]
    lib.gorder_oracle_timewise.argtypes = [vp, vp, vp, u64]
    lib.gorder_oracle_leaflets.argtypes = [vp, vp, vp, C.POINTER(u64)]
    lib.gorder_oracle_normals.argtypes = [vp, vp, vp]
    lib.gorder_oracle_set_normals.argtypes = [vp, vp, u32]
    lib.gorder_oracle_dynamic_normal.argtypes = [vp, vp, u32, u32, f32, vp, i32, vp]
    lib.gorder_oracle_last_error_index.argtypes = [vp]
    lib.gorder_oracle_last_error_index.restype = u64
    lib.gorder_oracle_vector_to.argtypes = [vp, vp, vp, i32, vp]
    lib.gorder_oracle_calc_sch.argtypes = [vp, vp, i32]
    lib.gorder_oracle_calc_sch.restype = f32
    lib.gorder_oracle_tick.argtypes = [f32]
    lib.gorder_oracle_tick.restype = C.c_int64
    lib.gorder_oracle_calc_order.argtypes = [C.c_int64, u64, u64]
    lib.gorder_oracle_calc_order.restype = f32
    lib.gorder_oracle_trig_batch.argtypes = [i32, i32, u32, u32, u32, vp]
    lib.gorder_oracle_trig_batch.restype = None
    lib.gorder_oracle_mirror_acosf.argtypes = [f32]
    lib.gorder_oracle_mirror_acosf.restype = f32
    lib.gorder_oracle_mirror_cosf.argtypes = [f32]
    lib.gorder_oracle_mirror_cosf.restype = f32
    lib.gorder_oracle_mirror_sinf.argtypes = [f32]
    lib.gorder_oracle_mirror_sinf.restype = f32
    lib.gorder_oracle_predict_hydrogens.argtypes = [u32, vp, vp, i32, vp]
    lib.gorder_oracle_estimate_error.argtypes = [vp, vp, u64, u64]
    lib.gorder_oracle_estimate_error.restype = f32
    lib.gorder_oracle_prefix_average.argtypes = [vp, vp, u64, vp]
    lib.gorder_oracle_prefix_average.restype = None
    lib.gorder_oracle_center.argtypes = [vp, vp, u32, vp, i32, vp]
    _lib = lib
    return lib


class OracleError(RuntimeError):
    def __init__(self, status, index=0):
        self.status, self.index = status, index
        super().__init__(f"oracle status {status} (index {index})")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- scalar building blocks ----------------------------------------------------------------
def vector_to(p1, p2, box, pbc=True):
    lib = load()
    p1, p2, box = _f32(p1), _f32(p2), _f32(box)
    out = np.zeros(3, dtype=np.float32)
    bad = lib.gorder_oracle_vector_to(p1.ctypes.data, p2.ctypes.data, box.ctypes.data, 1 if pbc else 0,
                                      out.ctypes.data)
    if bad:
        raise OracleError(103)
    return out


def trig_batch(fn: str, which: str, first_bits: int, stride: int, n: int) -> np.ndarray:
    """acos / cos / sin of the floats with bit patterns first_bits + i * stride: which = 'mirror' (the restatement of
    glibc's algorithms = what the device computes) or 'libm' (the host's)."""
    out = np.empty(n, dtype=np.float32)
    load().gorder_oracle_trig_batch({"acos": 0, "cos": 1, "sin": 2}[fn], {"mirror": 0, "libm": 1}[which], first_bits, stride, n,
                                    out.ctypes.data)
    return out


def calc_sch(v, n, trig=TRIG_LIBM):
    v, n = _f32(v), _f32(n)
    return float(np.float32(load().gorder_oracle_calc_sch(v.ctypes.data, n.ctypes.data, trig)))


def tick(s):
    return int(load().gorder_oracle_tick(float(np.float32(s))))


def calc_order(total, n, min_samples=1):
    return float(np.float32(load().gorder_oracle_calc_order(int(total), int(n), int(min_samples))))


def mirror_acosf(x):
    return float(np.float32(load().gorder_oracle_mirror_acosf(float(np.float32(x)))))


def mirror_cosf(x):
    return float(np.float32(load().gorder_oracle_mirror_cosf(float(np.float32(x)))))


def mirror_sinf(x):
    return float(np.float32(load().gorder_oracle_mirror_sinf(float(np.float32(x)))))


def predict_hydrogens(kind, positions, box, pbc=True):
    """positions: [3 or 4][3] in the order of the reference's index tuple -> [n_h][3]."""
    pos = np.zeros((4, 3), dtype=np.float32)
    p = _f32(positions)
    pos[:p.shape[0]] = p
    box = _f32(box)
    out = np.zeros((3, 3), dtype=np.float32)
    n = load().gorder_oracle_predict_hydrogens(int(kind), pos.ctypes.data, box.ctypes.data, 1 if pbc else 0,
                                               out.ctypes.data)
    if n < 0:
        raise OracleError(103)
    return out[:n].copy()


def estimate_error(sums, counts, n_blocks):
    s = np.ascontiguousarray(sums, dtype=np.int64)
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    return float(np.float32(load().gorder_oracle_estimate_error(s.ctypes.data, c.ctypes.data, s.size, n_blocks)))


def prefix_average(sums, counts):
    s = np.ascontiguousarray(sums, dtype=np.int64)
    c = np.ascontiguousarray(counts, dtype=np.uint64)
    out = np.zeros(s.size, dtype=np.float32)
    load().gorder_oracle_prefix_average(s.ctypes.data, c.ctypes.data, s.size, out.ctypes.data)
    return out


def center(xyz, idx, box, pbc=True):
    xyz, box = _f32(xyz), _f32(box)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    out = np.zeros(3, dtype=np.float32)
    load().gorder_oracle_center(xyz.ctypes.data, idx.ctypes.data, idx.size, box.ctypes.data, 1 if pbc else 0,
                                out.ctypes.data)
    return out


# ---- the engine ----------------------------------------------------------------------------
class OracleEngine:
    def __init__(self, tables: Tables, trig: int = TRIG_LIBM, n_threads: int = 1):
        self.lib = load()
        self.tables = tables
        ct, self._keep = tables.as_ctypes()
        self._h = C.c_void_p()
        st = self.lib.gorder_oracle_create(C.byref(ct), trig, n_threads, C.byref(self._h))
        if st != OK:
            raise OracleError(st)
        self.n_acc = self.lib.gorder_oracle_n_accumulators(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gorder_oracle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != OK:
            raise OracleError(st, self.lib.gorder_oracle_last_error_index(self._h))

    def submit(self, xyz, box, frame_index=None, passes=1):
        """`passes` > 1 (benchmark aid): the worker threads walk the batch that many times before the one reduce."""
        xyz = _f32(xyz)
        n_frames = xyz.shape[0]
        assert xyz.shape[1:] == (self.tables.n_atoms, 3)
        if frame_index is None:
            frame_index = np.arange(n_frames)
        fi = np.ascontiguousarray(frame_index, dtype=np.uint64)
        bp = None
        if box is not None:
            box = _f32(box).reshape(n_frames, 9)
            bp = box.ctypes.data
        if passes > 1:
            self._check(self.lib.gorder_oracle_submit_passes(self._h, xyz.ctypes.data, bp, fi.ctypes.data, n_frames, int(passes)))
        else:
            self._check(self.lib.gorder_oracle_submit(self._h, xyz.ctypes.data, bp, fi.ctypes.data, n_frames))

    submit_host = submit

    def ua_fast_fidelity(self, xyz, box):
        """Per-sample comparison of the GORDER_FLAG_UA_FAST_NORMALISE construction with the reference's arithmetic on the
        given frames (gorder_oracle_ua_fast_fidelity) -> dict."""
        xyz = _f32(xyz)
        n_frames = xyz.shape[0]
        bp = None
        if box is not None:
            box = _f32(box).reshape(n_frames, 9)
            bp = box.ctypes.data
        hist = np.zeros(16, dtype=np.uint64)
        out = np.zeros(5, dtype=np.uint64)
        self._check(self.lib.gorder_oracle_ua_fast_fidelity(self._h, xyz.ctypes.data, bp, n_frames, hist.ctypes.data, out.ctypes.data))
        n = int(out[0])
        return {"samples": n, "tick_difference_histogram": [int(v) for v in hist],
                "fraction_moved": float(1.0 - hist[0] / max(1, n)), "fraction_moved_by_more_than_one_tick": float(hist[2:].sum() / max(1, n)),
                "mean_shift_ticks": float((int(out[3]) - (1 << 62)) / max(1, n)),
                "carbons_sent_to_the_literal_loops": int(out[1]), "bond_positions_in_another_ordermap_tile": int(out[2]),
                "fraction_moved_default_path": float(int(out[4]) / max(1, n))}

    def prime_leaflets(self, xyz, box, frame_index):
        xyz = _f32(xyz)
        bp = _f32(box).reshape(9).ctypes.data if box is not None else None
        self._check(self.lib.gorder_oracle_prime_leaflets(self._h, xyz.ctypes.data, bp, frame_index))

    def set_manual_leaflets(self, flags, frame_index=0):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(self.lib.gorder_oracle_set_manual_leaflets(self._h, flags.ctypes.data, frame_index))

    def finish(self) -> Results:
        n = self.n_acc
        sums = np.zeros((3, n), dtype=np.int64)
        counts = np.zeros((3, n), dtype=np.uint64)
        nx, ny = C.c_uint32(), C.c_uint32()
        nt = self.lib.gorder_oracle_ordermap_dims(self._h, C.byref(nx), C.byref(ny))
        ms = mc = None
        msp = mcp = None
        if nt:
            ms = np.zeros((3, n, nx.value, ny.value), dtype=np.int64)
            mc = np.zeros((3, n, nx.value, ny.value), dtype=np.uint64)
            msp, mcp = ms.ctypes.data, mc.ctypes.data
        nf = C.c_uint64()
        self._check(self.lib.gorder_oracle_finish(self._h, sums.ctypes.data, counts.ctypes.data, msp, mcp,
                                                  C.byref(nf)))
        return Results(sums, counts, int(nf.value), ms, mc)

    def timewise(self, n_frames):
        n = self.n_acc
        s = np.zeros((n_frames, 3, n), dtype=np.int64)
        c = np.zeros((n_frames, 3, n), dtype=np.uint64)
        self._check(self.lib.gorder_oracle_timewise(self._h, s.ctypes.data, c.ctypes.data, n_frames))
        return s, c

    def set_normals(self, normals):
        n = np.ascontiguousarray(normals, dtype=np.float32)
        self._check(self.lib.gorder_oracle_set_normals(self._h, n.ctypes.data, n.shape[0]))

    def normals(self):
        """Dynamic membrane normals of the last analysed frame -> (normals [n_mol, 3], n_points [n_mol])."""
        n = np.zeros((self.tables.n_molecules_total, 3), dtype=np.float32)
        k = np.zeros(self.tables.n_molecules_total, dtype=np.uint32)
        self._check(self.lib.gorder_oracle_normals(self._h, n.ctypes.data, k.ctypes.data))
        return n, k

    def leaflets(self):
        n = self.tables.n_molecules_total
        flags = np.zeros(n, dtype=np.uint8)
        dist = np.zeros(n, dtype=np.float32)
        fr = C.c_uint64()
        self._check(self.lib.gorder_oracle_leaflets(self._h, flags.ctypes.data, dist.ctypes.data, C.byref(fr)))
        return flags, dist, int(fr.value)
