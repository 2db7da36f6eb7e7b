"""Shared helpers of the lifecycle tests: the hand-derived fixture and seeded random lifecycle columns."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fixture():
    return json.load(open(os.path.join(GOLDEN, "lifecycle_cases.json"), encoding="utf-8"))


def columns(d, seed, which, wide=False):
    """Lifecycle columns for the synthetic data set `d` (a dict from the synthetic generator).
    `which` is a 3-letter mask over (p)rice, (s)tate, (d)iscount, e.g. "p-d"."""
    rng = np.random.default_rng(seed)
    n, m = d["order_id"].size, d["order_customer"].size
    out = dict(line_price=None, order_state=None, order_descuento=None)
    if which[0] == "p":
        hi = 2**62 if wide else 500_000
        out["line_price"] = rng.integers(-hi if wide else 1, hi, size=n, dtype=np.int64)
    if which[1] == "s":
        out["order_state"] = (rng.integers(0, 256, size=m) & np.where(rng.integers(0, 5, size=m) == 0, 0xFE, 0xFF)).astype(np.uint8)
    if which[2] == "d":
        out["order_descuento"] = rng.integers(-2**31 if wide else 0, 2**31 if wide else 101, size=m, dtype=np.int64).astype(np.int32)
    return out


MASKS = ["p--", "-s-", "--d", "ps-", "p-d", "-sd", "psd"]
