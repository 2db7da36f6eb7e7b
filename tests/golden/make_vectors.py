"""Generate the committed golden vectors (inputs + expected outputs) under tests/golden/.

    python tests/golden/make_vectors.py

Inputs come from a pure-Python restatement of the synthetic recipe of include/acmpc.h (a third
implementation, next to the device generator and the C oracle's).  Expected outputs come from
oracle/checkout_oracle.py::aggregate_py, the row-at-a-time integer-cent restatement of
process/carrito.php:12-18 and process/confirmcompra.php:24,35-37.  The reference itself cannot be
run (no PHP/MySQL in the image) and holds no fixtures, so these vectors pin the oracle and the
engine to each other and to the hand-derived cases of checkout_cases.json, not to a reference run.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.checkout_oracle import aggregate_py  # noqa: E402

MASK = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
SEED_LEN, SEED_PROD, SEED_QTY, SEED_PRICE, SEED_CUST = 0xA11CE, 0xB0B, 0x51DE, 0xCAFE, 0xD00D
BASE = 5


def h64(seed, i):
    z = ((seed ^ ((i * GOLDEN) & MASK)) + GOLDEN) & MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def synth(n_lines, n_products, n_customers, zipf, qty_mode):
    cdf, w = [], 0
    if zipf:
        for k in range(n_products):
            w += (1 << 32) // (k + 1)
            cdf.append(w)
    oid, pidx, qty, cust = [], [], [], []
    line, o = 0, 0
    while line < n_lines:
        ln = 1 + h64(SEED_LEN, o) % 7
        ln = min(ln, n_lines - line)
        cust.append(h64(SEED_CUST, o) % n_customers)
        for i in range(ln):
            l = line + i
            hp = h64(SEED_PROD, l)
            if zipf:
                u = hp % w
                a, b = 0, n_products
                while a < b:
                    m = a + (b - a) // 2
                    if cdf[m] > u:
                        b = m
                    else:
                        a = m + 1
                p = a
            else:
                p = hp % n_products
            oid.append(BASE + o)
            pidx.append(p)
            qty.append(1 + h64(SEED_QTY, l) % 5 if qty_mode else 1)
        line += ln
        o += 1
    price = [(25000, 45000, 150000)[p] if p < 3 else 1 + h64(SEED_PRICE, p) % 500000 for p in range(n_products)]
    return (np.array(oid, np.int32), np.array(pidx, np.int32), np.array(qty, np.int32),
            np.array(cust, np.int32), np.array(price, np.int64))


def emit(name, n_lines, n_products, n_customers, zipf, qty_mode, mutate=None):
    oid, pidx, qty, cust, price = synth(n_lines, n_products, n_customers, zipf, qty_mode)
    if mutate:
        mutate(oid, pidx, qty, cust, price)
    tot, pq, ct = aggregate_py(oid, pidx, qty, cust, BASE, n_customers, price)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), order_id=oid, product_idx=pidx, qty=qty,
                        order_customer=cust, price_cents=price, order_id_base=np.int32(BASE),
                        n_customers=np.int32(n_customers), zipf=np.int32(zipf), qty_mode=np.int32(qty_mode),
                        order_total_cents=np.array(tot, np.int64), product_qty=np.array(pq, np.int64),
                        customer_total_cents=np.array(ct, np.int64))
    print(name, "lines", len(oid), "orders", len(cust), "sum", int(np.array(tot, np.int64).sum()))


def rough(oid, pidx, qty, cust, price):
    """Unknown products, negative and zero quantities, large prices: the engine's rare branches."""
    pidx[::17] = -1
    pidx[5::101] = len(price) + 3
    qty[::13] = -2
    qty[3::29] = 0
    qty[7::31] = 2_000_000_000
    price[10] = 9_000_000_000_000_000     # needs the 64-bit price table and wraps int64 sums
    price[11] = -123_456_789_012
    price[12] = 3_000_000_000              # just above int32


def main():
    emit("synth_1k_uniform", 1000, 100, 50, 0, 0)
    emit("synth_1k_zipf", 1000, 100, 50, 1, 0)
    emit("synth_64k_uniform", 65536, 10000, 100000, 0, 0)
    emit("synth_64k_zipf_qty", 65536 + 77, 10000, 100000, 1, 1)
    emit("synth_5k_rough", 5003, 300, 64, 0, 1, rough)


if __name__ == "__main__":
    main()
