"""MI355X-native batch backend for the checkout aggregation of a store.sql-schema shop.

Importing the package loads libacmpc.so (hand-written gfx950 HIP kernels behind a C ABI).  There
is no CPU path: a missing library is an import error.
"""
from . import _lib

_lib.load()          # fail loudly, at import, if the HIP extension has not been built

from .lote import ErrorLote, MotorLote, consultasLote, formato_subtotal   # noqa: E402
from . import emision, ingesta, reparto, sintetico                                   # noqa: E402

__all__ = ["ErrorLote", "MotorLote", "consultasLote", "formato_subtotal", "emision", "ingesta", "reparto", "sintetico"]
