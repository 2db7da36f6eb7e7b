"""The C-ABI library loads without a GPU and exports every symbol include/acmpc.h declares.
No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np

from ac_mpc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "acmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(acmpc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name


def test_version_and_error_strings():
    lib = _lib.load()
    assert lib.acmpc_version().decode().startswith("acmpc ")
    assert isinstance(lib.acmpc_last_error(None), bytes)


def test_init_rejects_cpu_backend_request():
    """A negative device is not a request for a CPU backend: the product has none and says so."""
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.acmpc_init(C.byref(ctx), -1)
    assert rc == _lib.E_INVALID and not ctx.value
    assert b"no CPU backend" in lib.acmpc_last_error(None)


def test_null_context_calls_fail_cleanly():
    lib = _lib.load()
    assert lib.acmpc_set_products(None, None, 0) == _lib.E_INVALID
    assert lib.acmpc_set_profiling(None, 1) == _lib.E_INVALID
    flags = C.c_uint32(0)
    assert lib.acmpc_status(None, None, C.byref(flags)) == _lib.E_INVALID
    out = C.c_void_p(1)
    assert lib.acmpc_host_alloc(None, 64, C.byref(out)) == _lib.E_INVALID
    assert lib.acmpc_host_free(None, None) == _lib.E_INVALID
    lib.acmpc_destroy(None)      # no-op


def test_host_side_synth_plan_and_prices_match_oracle(oracle):
    from ac_mpc_amd import sintetico
    for n in (0, 1, 1000, 65536, 1_000_003):
        sh = sintetico.plan(n)
        assert sh.n_orders_total == oracle.lib.oracle_synth_n_orders(n)
        assert (sh.order_lo, sh.order_hi, sh.line_lo, sh.line_hi) == (0, sh.n_orders_total, 0, n)
    np.testing.assert_array_equal(sintetico.precios(5000), oracle.synth_prices(5000))
    # shards tile the data set and cut on order boundaries
    n = 250_000
    m = sintetico.plan(n).n_orders_total
    prev_line, prev_order = 0, 0
    for g in range(8):
        sh = sintetico.plan(n, g, 8)
        assert sh.order_lo == prev_order and sh.line_lo == prev_line
        line_lo = C.c_int64(0)
        cnt = oracle.lib.oracle_synth_count_lines(n, sh.order_lo, sh.order_hi, C.byref(line_lo))
        assert (line_lo.value, cnt) == (sh.line_lo, sh.line_hi - sh.line_lo)
        prev_line, prev_order = sh.line_hi, sh.order_hi
    assert prev_line == n and prev_order == m
