"""Shared helpers of the emit tests: an independent (Python string formatting) restatement of the row formats
of include/acmpc.h and value sets that exercise every digit count and both signs."""
import numpy as np

I64_MIN, I64_MAX = -2**63, 2**63 - 1


def amount(c: int) -> str:
    m = abs(int(c))
    return f"{'-' if c < 0 else ''}{m // 100}.{m % 100:02d}"


def grouped(c: int) -> str:
    """number_format($suma, 2) of process/carrito.php:19 from integer cents (Python's own ',' grouping)."""
    m = abs(int(c))
    return f"{'-' if c < 0 else ''}{m // 100:,}.{m % 100:02d}"


def totals_text(totals, base: int, style: int) -> bytes:
    if style == 2:
        return "".join(f"<tr><td>Subtotal</td><td>${grouped(t)}</td></tr>\n" for t in totals).encode()
    if style == 0:
        return "".join(f"update venta set TotalPagar='{amount(t)}' where NumPedido='{base + i}';\n" for i, t in enumerate(totals)).encode()
    return "".join(f"{base + i},{amount(t)}\n" for i, t in enumerate(totals)).encode()


def stock_text(deltas, codes, style: int) -> bytes:
    out = []
    for d, c in zip(deltas, codes):
        d = int(d)
        if d == 0:
            continue
        if style == 0:
            out.append(f"update producto set Stock=Stock{'+' if d < 0 else '-'}{abs(d)} where CodigoProd='{c}';\n")
        else:
            out.append(f"{c},{d}\n")
    return "".join(out).encode("latin-1")


def edge_values() -> np.ndarray:
    """Zero, both signs, every power of ten and its neighbours, the int64 limits."""
    v = [0, 1, -1, 5, 9, 10, 99, 100, 101, -99, -100, -101, 220000, I64_MAX, I64_MIN, I64_MIN + 1, 2**32 - 1, 2**32, 2**32 + 1]
    for k in range(1, 19):
        for d in (-1, 0, 1):
            v += [10**k + d, -(10**k + d)]
    return np.array(v, dtype=np.int64)


def random_values(n: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    mag = rng.integers(0, 63, size=n)                       # spread over every magnitude
    v = rng.integers(0, 2**62, size=n, dtype=np.int64) >> (62 - mag)
    sign = rng.integers(0, 4, size=n) == 0
    return np.where(sign, -v, v).astype(np.int64)
