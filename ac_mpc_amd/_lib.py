"""ctypes binding of libacmpc.so (the C ABI declared in include/acmpc.h).

There is no fallback: if the shared library is missing or fails to load, importing the engine
raises.  Build it with ``python build_native.py`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libacmpc.so")

# Every symbol include/acmpc.h declares; tests check that the library exports all of them.
EXPORTS = (
    "acmpc_init", "acmpc_destroy", "acmpc_last_error", "acmpc_version", "acmpc_set_products",
    "acmpc_aggregate", "acmpc_aggregate_device", "acmpc_status", "acmpc_set_option", "acmpc_set_profiling",
    "acmpc_last_kernel_ms", "acmpc_lines_geometry", "acmpc_synth_plan", "acmpc_synth_prices",
    "acmpc_synth_generate_device", "acmpc_set_product_codes", "acmpc_ingest_detalle_device", "acmpc_ingest_detalle",
    "acmpc_aggregate_lifecycle", "acmpc_aggregate_lifecycle_device",
    "acmpc_emit_totals_device", "acmpc_emit_stock_device", "acmpc_emit_totals", "acmpc_emit_stock",
    "acmpc_host_alloc", "acmpc_host_free", "acmpc_aggregate_device_noalloc",
    "acmpc_comm_id", "acmpc_comm_init_rank", "acmpc_comm_init_all", "acmpc_comm_destroy", "acmpc_merge_device",
    "acmpc_merge_group_device", "acmpc_aggregate_sharded", "acmpc_shard_plan", "acmpc_shard_partition",
    "acmpc_last_line_order_path",
)

OK, E_INVALID, E_HIP, E_ALIGN, E_DATA, E_STATE, E_NOMEM = 0, -1, -2, -3, -4, -5, -6
ST_ORDER_RANGE, ST_ORDER_UNSORTED, ST_CUSTOMER_RANGE, ST_INGEST_MALFORMED, ST_INGEST_CAPACITY = 1, 2, 4, 8, 16
ST_EMIT_CAPACITY, ST_INTERNAL = 32, 64
EMIT_SQL, EMIT_CSV, EMIT_DISPLAY = 0, 1, 2
ORDER_COUNTS, DESCUENTO_UNITS, DESCUENTO_PERCENT = 1, 0, 1
OPT_LINE_ORDER, ORDER_GROUPED, ORDER_ANY, ORDER_AUTO = 1, 0, 1, 2
PATH_NONE, PATH_ATOMIC, PATH_MOVED, PATH_MOVED_STOCK, PATH_GROUPED = -1, 0, 1, 2, 3
K_LINES, K_FIXUP, K_CUSTOMERS, K_CUSTMERGE, N_KERNELS = 0, 1, 2, 3, 4


class SynthCfg(C.Structure):
    _fields_ = [("n_lines_total", C.c_int64), ("n_products", C.c_int32), ("n_customers", C.c_int32),
                ("order_id_base", C.c_int32), ("zipf", C.c_int32), ("qty_mode", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32)]


class SynthShard(C.Structure):
    _fields_ = [("n_orders_total", C.c_int64), ("order_lo", C.c_int64), ("order_hi", C.c_int64),
                ("line_lo", C.c_int64), ("line_hi", C.c_int64)]


class Lifecycle(C.Structure):
    _fields_ = [("line_price_cents", C.c_void_p), ("order_state", C.c_void_p), ("order_descuento", C.c_void_p),
                ("descuento_mode", C.c_int32)]


_lib = None


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.

    The PyTorch wheel bundles its own libamdhip64.so (soname libamdhip64.so.7, the soname libacmpc.so
    needs).  If libacmpc.so were loaded first it would pull in /opt/rocm's copy, a later `import torch`
    would load the bundled one as well, and whichever runtime initialises second finds no device.  So
    when PyTorch is installed, its runtime is mapped first and libacmpc.so binds to it; without PyTorch
    the system runtime is used.
    """
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libacmpc.so once and declare every prototype.  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python build_native.py`). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_uint32)
    lib.acmpc_init.argtypes = [C.POINTER(vp), C.c_int]
    lib.acmpc_init.restype = C.c_int
    lib.acmpc_destroy.argtypes = [vp]
    lib.acmpc_destroy.restype = None
    lib.acmpc_last_error.argtypes = [vp]
    lib.acmpc_last_error.restype = C.c_char_p
    lib.acmpc_version.argtypes = []
    lib.acmpc_version.restype = C.c_char_p
    lib.acmpc_set_products.argtypes = [vp, vp, i32]
    lib.acmpc_set_products.restype = C.c_int
    lib.acmpc_aggregate.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp]
    lib.acmpc_aggregate.restype = C.c_int
    lib.acmpc_aggregate_device.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp]
    lib.acmpc_aggregate_device.restype = C.c_int
    lib.acmpc_aggregate_device_noalloc.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp]
    lib.acmpc_aggregate_device_noalloc.restype = C.c_int
    lib.acmpc_comm_id.argtypes = [vp]
    lib.acmpc_comm_id.restype = C.c_int
    lib.acmpc_comm_init_rank.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    lib.acmpc_comm_init_rank.restype = C.c_int
    lib.acmpc_comm_init_all.argtypes = [vp, vp, C.c_int]
    lib.acmpc_comm_init_all.restype = C.c_int
    lib.acmpc_comm_destroy.argtypes = [vp]
    lib.acmpc_comm_destroy.restype = None
    lib.acmpc_merge_device.argtypes = [vp, vp, vp, i64, vp]
    lib.acmpc_merge_device.restype = C.c_int
    lib.acmpc_merge_group_device.argtypes = [vp, vp, C.c_int, vp, i64, vp]
    lib.acmpc_merge_group_device.restype = C.c_int
    lib.acmpc_aggregate_sharded.argtypes = [vp, vp, C.c_int, vp, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp]
    lib.acmpc_aggregate_sharded.restype = C.c_int
    lib.acmpc_shard_plan.argtypes = [vp, i64, i64, i32, C.c_int, C.c_int, vp, vp]
    lib.acmpc_shard_plan.restype = C.c_int
    lib.acmpc_shard_partition.argtypes = [vp, vp, vp, i64, i64, i32, C.c_int, vp, vp, vp, vp]
    lib.acmpc_shard_partition.restype = C.c_int
    lib.acmpc_last_line_order_path.argtypes = [vp, vp, C.POINTER(C.c_int)]
    lib.acmpc_last_line_order_path.restype = C.c_int
    lib.acmpc_status.argtypes = [vp, vp, u32p]
    lib.acmpc_status.restype = C.c_int
    lib.acmpc_set_option.argtypes = [vp, C.c_int, i64]
    lib.acmpc_set_option.restype = C.c_int
    lib.acmpc_set_profiling.argtypes = [vp, C.c_int]
    lib.acmpc_set_profiling.restype = C.c_int
    lib.acmpc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.acmpc_last_kernel_ms.restype = C.c_int
    lib.acmpc_lines_geometry.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.acmpc_lines_geometry.restype = C.c_int
    lib.acmpc_synth_plan.argtypes = [C.POINTER(SynthCfg), C.POINTER(SynthShard)]
    lib.acmpc_synth_plan.restype = C.c_int
    lib.acmpc_synth_prices.argtypes = [C.POINTER(SynthCfg), vp]
    lib.acmpc_synth_prices.restype = C.c_int
    lib.acmpc_synth_generate_device.argtypes = [vp, C.POINTER(SynthCfg), vp, vp, vp, vp, vp]
    lib.acmpc_synth_generate_device.restype = C.c_int
    lib.acmpc_host_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    lib.acmpc_host_alloc.restype = C.c_int
    lib.acmpc_host_free.argtypes = [vp, vp]
    lib.acmpc_host_free.restype = C.c_int
    lib.acmpc_set_product_codes.argtypes = [vp, vp, i32]
    lib.acmpc_set_product_codes.restype = C.c_int
    lib.acmpc_ingest_detalle_device.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, vp]
    lib.acmpc_ingest_detalle_device.restype = C.c_int
    lib.acmpc_ingest_detalle.argtypes = [vp, vp, i64, vp, vp, vp, i64, C.POINTER(i64), C.POINTER(i64)]
    lib.acmpc_ingest_detalle.restype = C.c_int
    lp = C.POINTER(Lifecycle)
    lib.acmpc_aggregate_lifecycle.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, i32, lp, vp, vp, vp]
    lib.acmpc_aggregate_lifecycle.restype = C.c_int
    lib.acmpc_aggregate_lifecycle_device.argtypes = [vp, vp, vp, vp, i64, vp, i64, i32, i32, lp, vp, vp, vp, vp]
    lib.acmpc_aggregate_lifecycle_device.restype = C.c_int
    lib.acmpc_emit_totals_device.argtypes = [vp, vp, i64, i32, C.c_int, vp, i64, vp, vp]
    lib.acmpc_emit_totals_device.restype = C.c_int
    lib.acmpc_emit_stock_device.argtypes = [vp, vp, i32, C.c_int, vp, i64, vp, vp]
    lib.acmpc_emit_stock_device.restype = C.c_int
    lib.acmpc_emit_totals.argtypes = [vp, vp, i64, i32, C.c_int, vp, i64, C.POINTER(i64)]
    lib.acmpc_emit_totals.restype = C.c_int
    lib.acmpc_emit_stock.argtypes = [vp, vp, i32, C.c_int, vp, i64, C.POINTER(i64)]
    lib.acmpc_emit_stock.restype = C.c_int
    _lib = lib
    return lib
