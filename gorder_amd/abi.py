"""ctypes view of include/gorder_hip.h plus a thin engine wrapper.

Python is harness only (tests, bench, multi-GPU launcher): the product is the C-ABI library
``gorder_amd/libgorder_hip.so`` built from ``gorder_amd/csrc`` (hand-written HIP for gfx950).
There is NO CPU fallback here: if the library or a GPU is missing every call raises.

The table classes mirror what gorder's ``SystemTopology`` holds for this path
(/root/reference/src/analysis/topology/mod.rs:34-65, topology/bond.rs:220-246,
leaflets.rs:575-590, 744-775).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GORDER_HIP_LIB") or os.path.join(_HERE, "libgorder_hip.so")   # env: A/B builds

# ---- status codes (gorder_status_t) ---------------------------------------------------------
OK = 0
ERR_UNDEFINED_BOX = 1
ERR_NOT_ORTHOGONAL_BOX = 2
ERR_ZERO_BOX = 3
ERR_UNDEFINED_POSITION = 4
ERR_INVALID_GLOBAL_MEMBRANE_CENTER = 5
ERR_INVALID_LOCAL_MEMBRANE_CENTER = 6
ERR_INVALID_ARGUMENT = 100
ERR_DEVICE = 101
ERR_NO_DEVICE = 102
ERR_BOX_RANGE = 103
ERR_LEAFLETS_NOT_PRIMED = 104
ERR_OVERFLOW = 105
ERR_TRAJECTORY_FORMAT = 106

LEAFLETS_NONE, LEAFLETS_GLOBAL, LEAFLETS_LOCAL, LEAFLETS_INDIVIDUAL, LEAFLETS_MANUAL = range(5)
FLAG_TRIG_ACOS_COS = 1
FLAG_UA_FAST_NORMALISE = 2      # united atoms: tolerance-bounded hydrogen construction (include/gorder_hip.h)
UA_CH1_SAT, UA_CH2, UA_CH3, UA_CH1_UNSAT = 1, 2, 3, 4
UA_N_H = {UA_CH1_SAT: 1, UA_CH2: 2, UA_CH3: 3, UA_CH1_UNSAT: 1}

_u32p = C.POINTER(C.c_uint32)


class CLeaflets(C.Structure):
    _fields_ = [("method", C.c_uint32), ("normal_dim", C.c_uint32), ("frequency", C.c_uint32),
                ("flip", C.c_uint32), ("radius", C.c_float), ("n_membrane", C.c_uint32),
                ("membrane", _u32p)]


class COrderMap(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("plane", C.c_uint32), ("span_x", C.c_float * 2),
                ("span_y", C.c_float * 2), ("bin", C.c_float * 2)]


class CUaAtom(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("indices", _u32p)]


class CMolType(C.Structure):
    _fields_ = [("n_molecules", C.c_uint32), ("n_bond_types", C.c_uint32), ("bonds", _u32p),
                ("n_ua_atoms", C.c_uint32), ("ua_atoms", C.POINTER(CUaAtom)), ("heads", _u32p),
                ("n_methyls", C.c_uint32), ("methyls", _u32p), ("normal_heads", _u32p)]


class CDynamicNormal(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("radius", C.c_float), ("n_cloud", C.c_uint32), ("cloud", _u32p)]


class CGeometry(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("invert", C.c_uint32), ("reference", C.c_uint32), ("point", C.c_float * 3),
                ("n_group", C.c_uint32), ("group", _u32p), ("xdim", C.c_float * 2), ("ydim", C.c_float * 2),
                ("zdim", C.c_float * 2), ("radius", C.c_float), ("span", C.c_float * 2), ("orientation", C.c_uint32),
                ("structure_box", C.c_float * 3)]


class CTables(C.Structure):
    _fields_ = [("n_atoms", C.c_uint32), ("n_molecule_types", C.c_uint32),
                ("molecule_types", C.POINTER(CMolType)), ("handle_pbc", C.c_int32),
                ("normal", C.c_float * 3), ("leaflets", CLeaflets), ("ordermap", COrderMap),
                ("timewise", C.c_int32), ("device", C.c_int32), ("flags", C.c_uint32), ("geometry", CGeometry),
                ("dynamic_normal", CDynamicNormal)]


class CTrajectory(C.Structure):
    _fields_ = [("paths", C.POINTER(C.c_char_p)), ("n_paths", C.c_uint32), ("group", _u32p), ("n_group", C.c_uint32),
                ("begin_ps", C.c_float), ("end_ps", C.c_float), ("step", C.c_uint32), ("n_threads", C.c_uint32),
                ("batch_frames", C.c_uint32), ("first_frame_index", C.c_uint64), ("device_decode", C.c_uint32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("reserved", C.c_uint32)]


class CTrajectoryStats(C.Structure):
    _fields_ = [("n_frames", C.c_uint64), ("n_batches", C.c_uint64), ("bytes_h2d", C.c_uint64),
                ("seconds_total", C.c_double), ("seconds_decode", C.c_double), ("seconds_reader_stalled", C.c_double),
                ("seconds_gpu_starved", C.c_double), ("batch_frames", C.c_uint32), ("decoder_threads", C.c_uint32),
                ("device_decode", C.c_uint32), ("frames_decoded_by_host", C.c_uint32), ("shard_first", C.c_uint64),
                ("shard_frames_total", C.c_uint64), ("seconds_setup", C.c_double)]


class CXtcFrame(C.Structure):
    """gorder_xtc_frame_t (include/gorder_xtc.h): one still-compressed frame for the device decoder."""
    _fields_ = [("offset", C.c_uint64), ("recip1", C.c_uint64), ("recip2", C.c_uint64), ("n_bytes", C.c_uint32),
                ("kind", C.c_uint32), ("minint", C.c_int32 * 3), ("sizeint", C.c_uint32 * 3), ("smallidx", C.c_int32),
                ("inv_precision", C.c_float), ("bitsize", C.c_uint32), ("bitsizeint", C.c_uint32)]


class CPlan(C.Structure):
    _fields_ = [("n_tiles", C.c_uint32), ("block_threads", C.c_uint32),
                ("max_window_atoms", C.c_uint32), ("n_direct_items", C.c_uint32),
                ("frames_per_stage", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("map_staged", C.c_uint32), ("map_lds_bytes", C.c_uint32), ("leaflets_one_read", C.c_uint32)]


# ---- python-side description of the tables ----------------------------------------------------
def _u32(a, shape=None) -> Optional[np.ndarray]:
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a: Optional[np.ndarray]):
    if a is None or a.size == 0:
        return _u32p()
    return a.ctypes.data_as(_u32p)


@dataclass
class MolType:
    """One molecule type (topology/molecule.rs:146-169) reduced to its index tables."""
    n_molecules: int
    bonds: Optional[np.ndarray] = None        # [n_bond_types, n_molecules, 2] uint32
    ua_atoms: List[tuple] = field(default_factory=list)   # [(kind, indices[n_molecules, 4])]
    heads: Optional[np.ndarray] = None        # [n_molecules]
    methyls: Optional[np.ndarray] = None      # [n_molecules, n_methyls]
    name: str = ""
    normal_heads: Optional[np.ndarray] = None  # [n_molecules] (dynamic membrane normals)

    @property
    def n_bond_types(self) -> int:
        return 0 if self.bonds is None else int(np.asarray(self.bonds).shape[0])

    @property
    def n_slots(self) -> int:
        return self.n_bond_types + sum(UA_N_H[int(k)] for k, _ in self.ua_atoms)


@dataclass
class Leaflets:
    method: int = LEAFLETS_NONE
    normal_dim: int = 2
    frequency: int = 1      # 0 = once; REAL frequency (input frequency * step)
    flip: bool = False
    radius: float = 0.0
    membrane: Optional[np.ndarray] = None


@dataclass
class OrderMap:
    enabled: bool = False
    plane: int = 0
    span_x: Sequence[float] = (0.0, 0.0)
    span_y: Sequence[float] = (0.0, 0.0)
    bin: Sequence[float] = (0.1, 0.1)


GEOM_NONE, GEOM_CUBOID, GEOM_CYLINDER, GEOM_SPHERE = range(4)
GEOMREF_POINT, GEOMREF_BOX_CENTER, GEOMREF_GROUP = range(3)
_INF = float("inf")


@dataclass
class Geometry:
    """input/geometry.rs: cuboid / cylinder / sphere relative to a point, the box centre or a group centre."""
    kind: int = GEOM_NONE
    invert: bool = False
    reference: int = GEOMREF_POINT
    point: Sequence[float] = (0.0, 0.0, 0.0)
    group: Optional[np.ndarray] = None
    xdim: Sequence[float] = (-_INF, _INF)
    ydim: Sequence[float] = (-_INF, _INF)
    zdim: Sequence[float] = (-_INF, _INF)
    radius: float = 0.0
    span: Sequence[float] = (-_INF, _INF)
    orientation: int = 2
    structure_box: Sequence[float] = (0.0, 0.0, 0.0)   # needed for reference = point with PBC


@dataclass
class DynamicNormal:
    """input/membrane_normal.rs DynamicNormal: local normals from the cloud of `heads` atoms within `radius`."""
    enabled: bool = False
    radius: float = 2.0
    cloud: Optional[np.ndarray] = None        # atom indices of group "NormalHeads"


@dataclass
class Tables:
    n_atoms: int
    molecule_types: List[MolType]
    handle_pbc: bool = True
    normal: Sequence[float] = (0.0, 0.0, 1.0)
    leaflets: Leaflets = field(default_factory=Leaflets)
    ordermap: OrderMap = field(default_factory=OrderMap)
    timewise: bool = False
    device: int = 0
    flags: int = 0          # FLAG_TRIG_ACOS_COS: acos->cos round trip like the reference
    geometry: Geometry = field(default_factory=Geometry)
    dynamic_normal: DynamicNormal = field(default_factory=DynamicNormal)

    @property
    def n_acc(self) -> int:
        return sum(m.n_slots for m in self.molecule_types)

    @property
    def n_molecules_total(self) -> int:
        return sum(m.n_molecules for m in self.molecule_types)

    @property
    def n_samples_per_frame(self) -> int:
        return sum(m.n_slots * m.n_molecules for m in self.molecule_types)

    def as_ctypes(self):
        """-> (CTables, keepalive list).  The C side copies what it needs during create()."""
        keep = []
        mts = (CMolType * max(1, len(self.molecule_types)))()
        for k, m in enumerate(self.molecule_types):
            c = mts[k]
            c.n_molecules = m.n_molecules
            b = _u32(m.bonds, (-1, m.n_molecules, 2)) if m.bonds is not None else None
            c.n_bond_types = 0 if b is None else b.shape[0]
            c.bonds = _ptr(b)
            ua = (CUaAtom * max(1, len(m.ua_atoms)))()
            for q, (kind, idx) in enumerate(m.ua_atoms):
                ia = _u32(idx, (m.n_molecules, 4))
                ua[q].kind = int(kind)
                ua[q].indices = _ptr(ia)
                keep.append(ia)
            c.n_ua_atoms = len(m.ua_atoms)
            c.ua_atoms = C.cast(ua, C.POINTER(CUaAtom))
            h = _u32(m.heads, (m.n_molecules,)) if m.heads is not None else None
            c.heads = _ptr(h)
            me = _u32(m.methyls, (m.n_molecules, -1)) if m.methyls is not None else None
            c.n_methyls = 0 if me is None else me.shape[1]
            c.methyls = _ptr(me)
            nh = _u32(m.normal_heads, (m.n_molecules,)) if m.normal_heads is not None else None
            c.normal_heads = _ptr(nh)
            keep += [b, ua, h, me, nh]
        t = CTables()
        t.n_atoms = self.n_atoms
        t.n_molecule_types = len(self.molecule_types)
        t.molecule_types = C.cast(mts, C.POINTER(CMolType))
        t.handle_pbc = 1 if self.handle_pbc else 0
        t.normal[:] = [float(x) for x in self.normal]
        lf = self.leaflets
        mem = _u32(lf.membrane) if lf.membrane is not None else None
        t.leaflets.method = lf.method
        t.leaflets.normal_dim = lf.normal_dim
        t.leaflets.frequency = lf.frequency
        t.leaflets.flip = 1 if lf.flip else 0
        t.leaflets.radius = lf.radius
        t.leaflets.n_membrane = 0 if mem is None else mem.size
        t.leaflets.membrane = _ptr(mem)
        om = self.ordermap
        t.ordermap.enabled = 1 if om.enabled else 0
        t.ordermap.plane = om.plane
        t.ordermap.span_x[:] = [float(x) for x in om.span_x]
        t.ordermap.span_y[:] = [float(x) for x in om.span_y]
        t.ordermap.bin[:] = [float(x) for x in om.bin]
        t.timewise = 1 if self.timewise else 0
        t.device = self.device
        t.flags = self.flags
        ge = self.geometry
        grp = _u32(ge.group) if ge.group is not None else None
        t.geometry.kind = ge.kind
        t.geometry.invert = 1 if ge.invert else 0
        t.geometry.reference = ge.reference
        t.geometry.point[:] = [float(x) for x in ge.point]
        t.geometry.n_group = 0 if grp is None else grp.size
        t.geometry.group = _ptr(grp)
        t.geometry.xdim[:] = [float(x) for x in ge.xdim]
        t.geometry.ydim[:] = [float(x) for x in ge.ydim]
        t.geometry.zdim[:] = [float(x) for x in ge.zdim]
        t.geometry.radius = ge.radius
        t.geometry.span[:] = [float(x) for x in ge.span]
        t.geometry.orientation = ge.orientation
        t.geometry.structure_box[:] = [float(x) for x in ge.structure_box]
        dn = self.dynamic_normal
        cloud = _u32(dn.cloud) if dn.cloud is not None else None
        t.dynamic_normal.enabled = 1 if dn.enabled else 0
        t.dynamic_normal.radius = dn.radius
        t.dynamic_normal.n_cloud = 0 if cloud is None else cloud.size
        t.dynamic_normal.cloud = _ptr(cloud)
        keep += [mts, mem, grp, cloud]
        return t, keep


# ---- the library ------------------------------------------------------------------------------
_EXPORTS = [
    "gorder_hip_create", "gorder_hip_destroy", "gorder_hip_n_accumulators", "gorder_hip_ordermap_dims",
    "gorder_hip_set_stream", "gorder_hip_submit_device", "gorder_hip_submit_host",
    "gorder_hip_prime_leaflets", "gorder_hip_set_manual_leaflets", "gorder_hip_synchronize",
    "gorder_hip_finish", "gorder_hip_timewise", "gorder_hip_leaflets", "gorder_hip_leaflet_distances",
    "gorder_hip_normals", "gorder_hip_export_maps", "gorder_hip_set_normals",
    "gorder_hip_accumulators_device", "gorder_hip_bind_accumulators", "gorder_hip_last_error_index", "gorder_hip_last_error_frame", "gorder_hip_kernel_time_names",
    "gorder_hip_last_error_message", "gorder_hip_strerror", "gorder_hip_kernel_time", "gorder_hip_kernel_time_group", "gorder_hip_plan",
    "gorder_hip_plan_tables", "gorder_hip_selftest_arithmetic", "gorder_hip_selftest_trig", "gorder_hip_run_trajectory",
    "gorder_hip_comm_unique_id", "gorder_hip_comm_create", "gorder_hip_comm_destroy", "gorder_hip_allreduce",
    "gorder_hip_reset", "gorder_hip_xtc_decode", "gorder_hip_release_staging", "gorder_hip_speculation_stats", "gorder_hip_local_decide_stats",
]

_lib = None


class GorderHipError(RuntimeError):
    def __init__(self, status: int, message: str = "", index: int = 0, frame: int = 0):
        self.status = status
        self.index = index
        self.frame = frame      # SystemTopology::frame of the first device error (gorder_hip_last_error_frame)
        super().__init__(f"gorder_hip status {status}: {message}")


def load_library() -> C.CDLL:
    """Load libgorder_hip.so (built by __graft_entry__.build / make -C gorder_amd/csrc). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GorderHipError(ERR_NO_DEVICE, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                                            "g.build()'` (there is no CPU fallback)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME
    # libamdhip64.so.7, same as /opt/rocm's).  Import torch FIRST so that our DT_NEEDED entry binds to
    # the runtime torch already initialised; two runtimes in one process cannot both own the device.
    try:
        import torch  # noqa: F401
        _rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(_rt):
            C.CDLL(_rt, mode=C.RTLD_GLOBAL)
    except ImportError:      # plain C/C++ hosts link against /opt/rocm directly
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    lib.gorder_hip_create.argtypes = [C.POINTER(CTables), C.POINTER(vp)]
    lib.gorder_hip_destroy.argtypes = [vp]
    lib.gorder_hip_destroy.restype = None
    lib.gorder_hip_n_accumulators.argtypes = [vp]
    lib.gorder_hip_n_accumulators.restype = u32
    lib.gorder_hip_ordermap_dims.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    lib.gorder_hip_ordermap_dims.restype = u32
    lib.gorder_hip_set_stream.argtypes = [vp, vp]
    lib.gorder_hip_submit_device.argtypes = [vp, vp, vp, vp, u32]
    lib.gorder_hip_submit_host.argtypes = [vp, vp, vp, vp, u32]
    lib.gorder_hip_prime_leaflets.argtypes = [vp, vp, vp, u64]
    lib.gorder_hip_set_manual_leaflets.argtypes = [vp, vp, u64]
    lib.gorder_hip_synchronize.argtypes = [vp]
    lib.gorder_hip_finish.argtypes = [vp, vp, vp, vp, vp, C.POINTER(u64)]
    lib.gorder_hip_timewise.argtypes = [vp, vp, vp, u64]
    lib.gorder_hip_leaflets.argtypes = [vp, vp, C.POINTER(u64)]
    lib.gorder_hip_leaflet_distances.argtypes = [vp, vp]
    lib.gorder_hip_normals.argtypes = [vp, vp, vp]
    lib.gorder_hip_export_maps.argtypes = [vp, vp, vp, u64]
    lib.gorder_hip_set_normals.argtypes = [vp, vp, u32]
    lib.gorder_hip_accumulators_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
    lib.gorder_hip_bind_accumulators.argtypes = [vp, vp, u64]
    lib.gorder_hip_last_error_index.argtypes = [vp]
    lib.gorder_hip_last_error_index.restype = u64
    lib.gorder_hip_kernel_time_names.argtypes = [vp]
    lib.gorder_hip_kernel_time_names.restype = C.c_char_p
    lib.gorder_hip_last_error_frame.argtypes = [vp]
    lib.gorder_hip_last_error_frame.restype = u64
    lib.gorder_hip_last_error_message.argtypes = [vp]
    lib.gorder_hip_last_error_message.restype = C.c_char_p
    lib.gorder_hip_strerror.argtypes = [i32]
    lib.gorder_hip_strerror.restype = C.c_char_p
    lib.gorder_hip_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u64), i32]
    lib.gorder_hip_kernel_time_group.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(u64)]
    lib.gorder_hip_plan.argtypes = [vp, C.POINTER(CPlan)]
    lib.gorder_hip_speculation_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.gorder_hip_local_decide_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.gorder_hip_plan_tables.argtypes = [C.POINTER(CTables), C.POINTER(CPlan), C.POINTER(i32)]
    lib.gorder_hip_selftest_arithmetic.argtypes = [i32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.gorder_hip_selftest_trig.argtypes = [i32, i32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.gorder_hip_run_trajectory.argtypes = [vp, C.POINTER(CTrajectory), C.POINTER(CTrajectoryStats)]
    lib.gorder_hip_comm_unique_id.argtypes = [vp]
    lib.gorder_hip_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
    lib.gorder_hip_comm_destroy.argtypes = [vp]
    lib.gorder_hip_comm_destroy.restype = None
    lib.gorder_hip_allreduce.argtypes = [vp, vp]
    lib.gorder_hip_reset.argtypes = [vp]
    lib.gorder_hip_xtc_decode.argtypes = [vp, vp, u64, vp, u32, u32, vp, u32, vp, u32]
    lib.gorder_hip_release_staging.argtypes = [vp]
    lib.gorder_hip_release_staging.restype = None
    _lib = lib
    return lib


def selftest_trig(fn: str, first_bits: int, stride: int, n: int, device: int = 0) -> np.ndarray:
    """The device's acos ('acos', 'acos_cores'), cos or sin of the floats with bit patterns first_bits + i * stride."""
    lib = load_library()
    out = np.empty(n, dtype=np.float32)
    st = lib.gorder_hip_selftest_trig(device, {"acos": 0, "acos_cores": 1, "cos": 2, "sin": 3}[fn], first_bits, stride, n, out.ctypes.data)
    if st != 0:
        raise GorderHipError(st, "gorder_hip_selftest_trig")
    return out


def selftest_arithmetic(n: int = 1 << 26, seed: int = 1, device: int = 0):
    """(division mismatches, square-root mismatches) of the kernels' Newton-core forms against the IEEE operations on
    n random operand sets inside the guarded ranges (gorder_hip_selftest_arithmetic); both must be 0."""
    lib = load_library()
    out = (C.c_uint64 * 2)()
    st = lib.gorder_hip_selftest_arithmetic(device, n, seed, out)
    if st != 0:
        raise GorderHipError(st, lib.gorder_hip_strerror(st).decode())
    return int(out[0]), int(out[1])


def plan_tables(tables: Tables) -> dict:
    """Host-only: how the library would tile these tables (no device needed)."""
    lib = load_library()
    ct, keep = tables.as_ctypes()
    plan, check = CPlan(), C.c_int(-1)
    st = lib.gorder_hip_plan_tables(C.byref(ct), C.byref(plan), C.byref(check))
    if st != OK:
        raise GorderHipError(st, lib.gorder_hip_strerror(st).decode())
    d = {name: getattr(plan, name) for name, _ in CPlan._fields_}
    d["selfcheck"] = check.value
    return d


@dataclass
class Results:
    sums: np.ndarray        # int64  [3, n_acc]   total / upper / lower
    counts: np.ndarray      # uint64 [3, n_acc]
    n_frames: int
    map_sums: Optional[np.ndarray] = None
    map_counts: Optional[np.ndarray] = None

    def order_ticks(self, min_samples: int = 1) -> np.ndarray:
        """Mean order parameter in integer ticks of 1e-6 (`OrderValue / usize`, order.rs:34-42);
        INT64_MIN marks "fewer than min_samples" (NaN in `order`)."""
        n = self.counts.astype(np.int64)
        ok = n >= max(1, min_samples)
        q = np.full(self.sums.shape, np.iinfo(np.int64).min, dtype=np.int64)
        q[ok] = (np.abs(self.sums[ok]) // n[ok]) * np.sign(self.sums[ok])
        return q

    def order(self, min_samples: int = 1) -> np.ndarray:
        """AnalysisOrder::calc_order (order.rs:101-107): truncating i64 division, then /1e6 as f32."""
        out = np.full(self.sums.shape, np.nan, dtype=np.float32)
        n = self.counts.astype(np.int64)
        ok = n >= max(1, min_samples)
        # Rust i64 `/` truncates toward zero (numpy // floors)
        q = (np.abs(self.sums[ok]) // n[ok]) * np.sign(self.sums[ok])
        out[ok] = (q.astype(np.float64) / 1e6).astype(np.float32)
        return out


class HipEngine:
    """One `SystemTopology` on one GPU (one handle per rank / stream)."""

    def __init__(self, tables: Tables):
        self.lib = load_library()
        self.tables = tables
        ct, keep = tables.as_ctypes()
        self._h = C.c_void_p()
        st = self.lib.gorder_hip_create(C.byref(ct), C.byref(self._h))
        if st != OK:
            msg = self.lib.gorder_hip_strerror(st).decode()
            if self._h:
                msg += ": " + self.lib.gorder_hip_last_error_message(self._h).decode()
                self.lib.gorder_hip_destroy(self._h)
                self._h = C.c_void_p()
            raise GorderHipError(st, msg)
        self.n_acc = self.lib.gorder_hip_n_accumulators(self._h)
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self.lib.gorder_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != OK:
            msg = self.lib.gorder_hip_strerror(st).decode() + ": " + \
                self.lib.gorder_hip_last_error_message(self._h).decode()
            raise GorderHipError(st, msg, self.lib.gorder_hip_last_error_index(self._h),
                                 self.lib.gorder_hip_last_error_frame(self._h))

    def set_stream(self, stream_ptr: int):
        self._check(self.lib.gorder_hip_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_torch_stream(self):
        """Launch on torch's CURRENT stream, so that torch ops (and torch.distributed collectives) issued on it are
        ordered with the handle's work.  torch's default stream is the NULL stream, which the C ABI reads as "the
        handle's own stream": make a torch.cuda.Stream current first if that ordering matters."""
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _frame_index(frame_index, n_frames):
        if frame_index is None:
            frame_index = np.arange(n_frames, dtype=np.uint64)
        fi = np.ascontiguousarray(frame_index, dtype=np.uint64)
        assert fi.shape == (n_frames,)
        return fi

    def submit_device(self, xyz, box, frame_index=None):
        """xyz: torch float32 CUDA tensor [F, N, 3]; box: [F, 3, 3] (or None when handle_pbc=False)."""
        assert xyz.is_cuda and xyz.is_contiguous() and xyz.dtype.is_floating_point and xyz.element_size() == 4
        n_frames = xyz.shape[0]
        assert xyz.shape[1] == self.tables.n_atoms and xyz.shape[2] == 3
        fi = self._frame_index(frame_index, n_frames)
        bp = None
        if box is not None:
            assert box.is_cuda and box.is_contiguous() and box.numel() == 9 * n_frames
            bp = C.c_void_p(box.data_ptr())
        self._check(self.lib.gorder_hip_submit_device(self._h, C.c_void_p(xyz.data_ptr()), bp,
                                                      fi.ctypes.data_as(C.c_void_p), n_frames))

    def submit_host(self, xyz: np.ndarray, box: Optional[np.ndarray], frame_index=None):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n_frames = xyz.shape[0]
        assert xyz.shape[1:] == (self.tables.n_atoms, 3)
        fi = self._frame_index(frame_index, n_frames)
        bp = None
        if box is not None:
            box = np.ascontiguousarray(box, dtype=np.float32).reshape(n_frames, 9)
            bp = box.ctypes.data_as(C.c_void_p)
        self._check(self.lib.gorder_hip_submit_host(self._h, xyz.ctypes.data_as(C.c_void_p), bp,
                                                    fi.ctypes.data_as(C.c_void_p), n_frames))

    def run_trajectory(self, paths, group=None, begin: float = 0.0, end: float = -1.0, step: int = 1, threads: int = 0,
                       batch_frames: int = 0, first_frame_index: int = 0, device_decode: bool = False,
                       shard=None) -> dict:
        """The reference's `read_trajectory` (common.rs:239-342) as one library call: read (and concatenate) the
        files, apply the time window / step, decode on `threads` host threads and analyse batch by batch with copies
        and kernels overlapped (gorder_hip_run_trajectory).  `device_decode`: the host threads only copy the compressed
        XTC blocks, the device unpacks them (k_xtc_scan + k_xtc_chunks).  `shard` = (i, n): analyse only the i-th of n
        contiguous shares of the selected frames (one rank of a multi-GPU run).  -> the pipeline's statistics."""
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        grp = None if group is None else np.ascontiguousarray(group, dtype=np.uint32)
        t = CTrajectory()
        t.paths = C.cast(arr, C.POINTER(C.c_char_p))
        t.n_paths = len(paths)
        t.group = _ptr(grp)
        t.n_group = 0 if grp is None else grp.size
        t.begin_ps, t.end_ps, t.step = begin, end, step
        t.n_threads, t.batch_frames, t.first_frame_index = threads, batch_frames, first_frame_index
        t.device_decode = 1 if device_decode else 0
        if shard is not None:          # (index, count): this rank's contiguous share of the selected frames
            t.shard_index, t.shard_count = shard
        stats = CTrajectoryStats()
        self._check(self.lib.gorder_hip_run_trajectory(self._h, C.byref(t), C.byref(stats)))
        return {name: getattr(stats, name) for name, _ in CTrajectoryStats._fields_}

    def xtc_decode(self, d_blob: int, blob_bytes: int, d_frames: int, n_frames: int, n_atoms_file: int, d_slot_of: int,
                   n_stop: int, d_xyz: int, n_atoms_out: int):
        """Decompress packed XTC frames on the device (gorder_hip_xtc_decode); all pointers are device addresses
        (d_slot_of may be 0).  Asynchronous on the handle's stream."""
        self._check(self.lib.gorder_hip_xtc_decode(self._h, d_blob, blob_bytes, d_frames, n_frames, n_atoms_file,
                                                   d_slot_of or None, n_stop, d_xyz, n_atoms_out))

    def release_staging(self):
        """Give back the staging buffers run_trajectory keeps between calls (gorder_hip_release_staging)."""
        self.lib.gorder_hip_release_staging(self._h)

    def reset(self):
        """A fresh SystemTopology on the same tables (gorder_hip_reset)."""
        self._check(self.lib.gorder_hip_reset(self._h))

    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the 128-byte RCCL id the other ranks need for comm_create (ship it over any channel)."""
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        st = lib.gorder_hip_comm_unique_id(buf)
        if st != OK:
            raise GorderHipError(st, "ncclGetUniqueId failed (is RCCL loadable?)")
        return bytes(buf)

    def comm_create(self, unique_id: bytes, n_ranks: int, rank: int):
        """An RCCL communicator (ncclComm_t) on this handle's device -> opaque pointer for allreduce()."""
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        comm = C.c_void_p()
        self._check(self.lib.gorder_hip_comm_create(self._h, buf, n_ranks, rank, C.byref(comm)))
        return comm

    def comm_destroy(self, comm):
        self.lib.gorder_hip_comm_destroy(comm)

    def allreduce(self, comm):
        """SystemTopology::reduce across the ranks of `comm`: one group of RCCL all-reduces issued by the library."""
        self._check(self.lib.gorder_hip_allreduce(self._h, comm))

    def prime_leaflets_device(self, xyz, box, frame_index: int):
        bp = C.c_void_p(box.data_ptr()) if box is not None else None
        self._check(self.lib.gorder_hip_prime_leaflets(self._h, C.c_void_p(xyz.data_ptr()), bp, frame_index))

    def set_manual_leaflets(self, flags: np.ndarray, frame_index: int = 0):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        assert flags.size == self.tables.n_molecules_total
        self._check(self.lib.gorder_hip_set_manual_leaflets(self._h, flags.ctypes.data_as(C.c_void_p), frame_index))

    def synchronize(self):
        self._check(self.lib.gorder_hip_synchronize(self._h))

    def ordermap_dims(self):
        """(nx, ny) tiles of every ordermap; (0, 0) when ordermaps are off."""
        nx, ny = C.c_uint32(), C.c_uint32()
        nt = self.lib.gorder_hip_ordermap_dims(self._h, C.byref(nx), C.byref(ny))
        return (int(nx.value), int(ny.value)) if nt else (0, 0)

    def finish(self) -> Results:
        n = self.n_acc
        sums = np.zeros((3, n), dtype=np.int64)
        counts = np.zeros((3, n), dtype=np.uint64)
        nx, ny = C.c_uint32(), C.c_uint32()
        nt = self.lib.gorder_hip_ordermap_dims(self._h, C.byref(nx), C.byref(ny))
        ms = mc = None
        msp = mcp = None
        if nt:
            ms = np.zeros((3, n, nx.value, ny.value), dtype=np.int64)
            mc = np.zeros((3, n, nx.value, ny.value), dtype=np.uint64)
            msp, mcp = ms.ctypes.data_as(C.c_void_p), mc.ctypes.data_as(C.c_void_p)
        nf = C.c_uint64()
        self._check(self.lib.gorder_hip_finish(self._h, sums.ctypes.data_as(C.c_void_p),
                                               counts.ctypes.data_as(C.c_void_p), msp, mcp, C.byref(nf)))
        return Results(sums, counts, int(nf.value), ms, mc)

    def timewise(self, n_frames: int):
        n = self.n_acc
        s = np.zeros((n_frames, 3, n), dtype=np.int64)
        c = np.zeros((n_frames, 3, n), dtype=np.uint64)
        self._check(self.lib.gorder_hip_timewise(self._h, s.ctypes.data_as(C.c_void_p),
                                                 c.ctypes.data_as(C.c_void_p), n_frames))
        return s, c

    def leaflets(self):
        flags = np.zeros(self.tables.n_molecules_total, dtype=np.uint8)
        fr = C.c_uint64()
        self._check(self.lib.gorder_hip_leaflets(self._h, flags.ctypes.data_as(C.c_void_p), C.byref(fr)))
        return flags, int(fr.value)

    def leaflet_distances(self) -> np.ndarray:
        d = np.zeros(self.tables.n_molecules_total, dtype=np.float32)
        self._check(self.lib.gorder_hip_leaflet_distances(self._h, d.ctypes.data_as(C.c_void_p)))
        return d

    def set_normals(self, normals: np.ndarray):
        """Manual membrane normals [n_frames, n_molecules_total, 3] for the NEXT submit call."""
        n = np.ascontiguousarray(normals, dtype=np.float32)
        assert n.ndim == 3 and n.shape[1:] == (self.tables.n_molecules_total, 3)
        self._check(self.lib.gorder_hip_set_normals(self._h, n.ctypes.data_as(C.c_void_p), n.shape[0]))

    def normals(self):
        """Dynamic membrane normals of the last submitted frame -> (normals [n_mol, 3] f32, n_points [n_mol])."""
        n = np.zeros((self.tables.n_molecules_total, 3), dtype=np.float32)
        k = np.zeros(self.tables.n_molecules_total, dtype=np.uint32)
        self._check(self.lib.gorder_hip_normals(self._h, n.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p)))
        return n, k

    def accumulator_words(self) -> int:
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self.lib.gorder_hip_accumulators_device(self._h, C.byref(p), C.byref(n)))
        return int(n.value)

    def flush(self):
        """Make the packed accumulator block complete (stream-ordered) — call before a collective on it."""
        self.accumulator_words()

    def bind_accumulators(self, tensor):
        """Accumulate into a caller-owned torch.int64 CUDA tensor (so that torch.distributed can
        all-reduce it over RCCL)."""
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.element_size() == 8
        self._keep.append(tensor)
        self._check(self.lib.gorder_hip_bind_accumulators(self._h, C.c_void_p(tensor.data_ptr()), tensor.numel()))

    def export_maps(self, sums, counts):
        """Copy the ordermaps (i64 sums, u64 counts, [3][n_acc][nx*ny]) into two caller-owned torch.int64 CUDA
        tensors so that torch.distributed can all-reduce them like the accumulator block."""
        for t in (sums, counts):
            assert t.is_cuda and t.is_contiguous() and t.element_size() == 8
        assert sums.numel() == counts.numel()
        self._check(self.lib.gorder_hip_export_maps(self._h, C.c_void_p(sums.data_ptr()), C.c_void_p(counts.data_ptr()),
                                                     sums.numel()))

    def kernel_time(self, reset: bool = False):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.lib.gorder_hip_kernel_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, int(n.value)

    def kernel_names(self) -> str:
        """The kernel groups gorder_hip_kernel_time has measured since the last reset, joined by ' + '."""
        return self.lib.gorder_hip_kernel_time_names(self._h).decode()

    def kernel_groups(self):
        """[(name, ms, segments)] per kernel group of the timed submits since the last reset (the times add up to
        kernel_time()'s ms).  Call before kernel_time(reset=True)."""
        out, k = [], 0
        while True:
            name, ms, n = C.c_char_p(), C.c_double(), C.c_uint64()
            if self.lib.gorder_hip_kernel_time_group(self._h, k, C.byref(name), C.byref(ms), C.byref(n)) != 0:
                return out
            out.append((name.value.decode(), ms.value, int(n.value)))
            k += 1

    def speculation_stats(self) -> dict:
        """One-read global leaflets (gorder_hip_speculation_stats): batches run that way, pairs moved, frames left to the
        exact kernel, and whether the handle still speculates."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.gorder_hip_speculation_stats(self._h, out))
        return {"batches": int(out[0]), "moved": int(out[1]), "exact_frames": int(out[2]), "enabled": bool(out[3])}

    def local_decide_stats(self) -> dict:
        """Local leaflets (gorder_hip_local_decide_stats): submits that ran the bound kernel, submits that paused it, and the
        last report read back (frames left open, frames seen)."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.gorder_hip_local_decide_stats(self._h, out))
        return {"submits": int(out[0]), "paused": int(out[1]), "open_frames": int(out[2]), "frames": int(out[3])}

    def plan(self) -> dict:
        p = CPlan()
        self._check(self.lib.gorder_hip_plan(self._h, C.byref(p)))
        return {name: getattr(p, name) for name, _ in CPlan._fields_}
