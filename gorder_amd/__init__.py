"""gorder_amd — MI355X-native (gfx950) engine for gorder's per-frame order-parameter path.

The product is the C-ABI shared library built from ``gorder_amd/csrc`` (see ``include/gorder_hip.h``);
this package is the thin Python harness around it (ctypes bindings, synthetic workloads,
multi-GPU launcher).  Nothing here imports ``oracle/``.
"""
from .abi import (Geometry, GorderHipError, HipEngine, Leaflets, MolType, OrderMap, Results, Tables,  # noqa: F401
                  load_library, plan_tables)

__all__ = ["Geometry", "GorderHipError", "HipEngine", "Leaflets", "MolType", "OrderMap", "Results", "Tables",
           "load_library", "plan_tables"]
