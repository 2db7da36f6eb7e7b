"""checkout_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A pure-Python restatement of the reference checkout, one request at a time,
over in-memory copies of the `producto`, `cliente`, `venta` and `detalle`
tables.  Only tests, the smoke check and the benchmark's cpu_baseline leg may
import this module.

PARITY UNPINNED BY THE REFERENCE: the reference has no tests or fixtures and
neither PHP nor MySQL/MariaDB is present in the build image, so nothing here
was compared with a run of the reference.  The arithmetic that lives in PHP
7.3/7.4 and MariaDB 10.4 (store.sql:7-8) is restated from their documented
behaviour:

* ``$suma = 0; $suma += $fila['Precio']``  (process/carrito.php:5,16):
  ``Precio`` arrives from mysqli as a numeric string with a decimal point
  ("250.00"), so the addition is an IEEE-754 double addition, in cart order.
* ``number_format($suma, 2)``  (process/carrito.php:19): round half away from
  zero to 2 places after PHP's 15-significant-digit pre-rounding, comma as the
  thousands separator.
* ``'".$_SESSION['sumaTotal']."'``  (process/confirmcompra.php:24): PHP 7
  float -> string uses ``precision=14``, i.e. ``%.14G``.
* ``venta.TotalPagar DECIMAL(30,2)``  (store.sql:163): MariaDB parses the
  string exactly and rounds half away from zero to 2 places.
* ``venta.NumPedido`` is AUTO_INCREMENT (store.sql:231-232); ``detalle`` rows
  carry ``CantidadProductos = '1'`` (process/confirmcompra.php:36) and must
  satisfy the foreign key on ``producto.CodigoProd`` (store.sql:242).

The module also provides the integer-cent path (`cents_total`) and asserts in
`Checkout.confirmcompra` that both paths give the same TotalPagar whenever the
preconditions of the equivalence argument hold (see `equivalence_bound_ok`).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from decimal import Decimal, ROUND_HALF_UP
from typing import Dict, List, Optional, Tuple

CENT = Decimal("0.01")


def php_float_to_string(x: float) -> str:
    """PHP 7 ``(string)$float`` with ini ``precision=14`` (``%.14G``): fixed notation while the value
    has at most 14 digits before the point, exponent notation from 1e14 on (``1.0E+14``)."""
    if x == int(x) and abs(x) < 1e14:
        return str(int(x))
    s = "%.14G" % x
    if "E" in s:
        mant, exp = s.split("E")
        if "." not in mant:
            mant += ".0"
        s = mant + "E" + ("+" if int(exp) >= 0 else "-") + str(abs(int(exp)))
    return s


def mariadb_decimal_30_2(text: str) -> Decimal:
    """String -> DECIMAL(30,2): exact parse, round half away from zero."""
    return Decimal(text).quantize(CENT, rounding=ROUND_HALF_UP)


def php_number_format_2(x: float) -> str:
    """``number_format($x, 2)``: PHP rounds with a 15-digit pre-rounding step,
    then groups thousands with ','. (process/carrito.php:19)"""
    d = Decimal("%.15g" % x).quantize(CENT, rounding=ROUND_HALF_UP)
    sign = "-" if d < 0 else ""
    digits = f"{abs(d):.2f}"
    whole, frac = digits.split(".")
    groups = []
    while len(whole) > 3:
        groups.insert(0, whole[-3:])
        whole = whole[:-3]
    groups.insert(0, whole)
    return sign + ",".join(groups) + "." + frac


def to_cents(price: Decimal) -> int:
    """DECIMAL(30,2) value -> integer cents (exact)."""
    scaled = price * 100
    assert scaled == scaled.to_integral_value(), "price has more than 2 decimals"
    return int(scaled)


def equivalence_bound_ok(prices: List[Decimal]) -> bool:
    """True when the double path is provably equal to the integer-cent path:
    total below 10^12 currency units (14 significant digits still reach the
    cent) and the worst-case accumulated rounding error stays far below half a
    cent.  With L lines of price <= Pmax the error is <= L * ulp(L*Pmax) / 2."""
    if not prices:
        return True
    n = len(prices)
    pmax = max(abs(p) for p in prices)
    total_bound = float(pmax) * n
    if total_bound >= 1e12:
        return False
    import math
    ulp = math.ulp(total_bound) if total_bound > 0 else 0.0
    return n * ulp / 2 < 0.0049


class Die(Exception):
    """The reference ends the request with die("...") (library/consulSQL.php:12,19,31,38,45)."""


@dataclass
class Store:
    """In-memory copy of the four tables the checkout touches."""
    producto: Dict[str, Decimal] = field(default_factory=dict)      # CodigoProd -> Precio
    producto_nombre: Dict[str, str] = field(default_factory=dict)
    cliente: Dict[Tuple[str, str], int] = field(default_factory=dict)  # (Usuario, Clave md5) -> id_Cliente
    venta: List[dict] = field(default_factory=list)
    detalle: List[Tuple[int, str, int]] = field(default_factory=list)  # (NumPedido, CodigoProd, CantidadProductos)
    venta_auto_increment: int = 5                                    # store.sql:232

    @classmethod
    def from_seed(cls, seed: dict) -> "Store":
        s = cls()
        for row in seed["producto"]:
            s.producto[row["CodigoProd"]] = Decimal(row["Precio"])
            s.producto_nombre[row["CodigoProd"]] = row["NombreProd"]
        for row in seed["cliente"]:
            s.cliente[(row["Usuario"], row["Clave"])] = int(row["id_Cliente"])
        s.venta_auto_increment = int(seed["venta_auto_increment"])
        return s


@dataclass
class Session:
    """$_SESSION keys used by the cart (inc/navbar.php:4-6, process/carrito.php:7-8,21)."""
    producto: List[str] = field(default_factory=list)
    contador: int = 0
    sumaTotal: object = None     # unset until carrito.php has run once


class Checkout:
    def __init__(self, store: Store):
        self.store = store

    # process/carrito.php:5-21
    def carrito(self, session: Session, precio: Optional[str] = None) -> str:
        suma = 0                                   # :5  int 0
        if precio is not None:                     # :6-9 ($_GET['precio'] is really a CodigoProd)
            session.producto.append(precio)
            session.contador += 1
        for i in range(session.contador):          # :12
            code = session.producto[i]
            if code in self.store.producto:        # :13-14 unknown code -> zero rows
                precio_str = f"{self.store.producto[code]:.2f}"   # mysqli delivers DECIMAL as a string
                suma = suma + float(precio_str)    # :16 numeric string with '.' -> double add
        session.sumaTotal = suma                   # :21
        return "Subtotal $" + php_number_format_2(float(suma))   # :19

    # process/confirmcompra.php:14-49
    def confirmcompra(self, session: Session, usuario: str, clave_md5: str, fecha: str = "01-01-2022") -> str:
        nit = self.store.cliente.get((usuario, clave_md5))        # :14-16
        if nit is None:
            return "El nombre o contraseña invalidos"             # :49
        if not (session.sumaTotal is not None and session.sumaTotal > 0):   # :18
            return "No has seleccionado ningún producto, revisa el carrito de compras"  # :45
        # :24 — the double is interpolated into the SQL text, MariaDB parses it into DECIMAL(30,2)
        total_text = php_float_to_string(float(session.sumaTotal))
        total_pagar = mariadb_decimal_30_2(total_text)
        num_pedido = self.store.venta_auto_increment
        self.store.venta_auto_increment += 1
        self.store.venta.append({"NumPedido": num_pedido, "Fecha": fecha, "NIT": str(nit), "Descuento": 0,
                                 "TotalPagar": total_pagar, "Estado": "Pendiente"})
        # :27-32 — newest order of this NIT
        mine = [v["NumPedido"] for v in self.store.venta if v["NIT"] == str(nit)]
        num_pedido = max(mine)
        # integer-cent path must agree whenever the equivalence bound holds
        known = [self.store.producto[c] for c in session.producto[:session.contador] if c in self.store.producto]
        if equivalence_bound_ok(known):
            assert to_cents(total_pagar) == cents_total(self.store, session.producto[:session.contador]), \
                "double path and integer-cent path disagree"
        # :35-37 — one detalle row per cart entry, qty 1; FK on CodigoProd (store.sql:242)
        for i in range(session.contador):
            code = session.producto[i]
            if code not in self.store.producto:
                raise Die("Problemas al insertar")                # consulSQL.php:19 (mid-loop: partial order stays)
            self.store.detalle.append((num_pedido, code, 1))
        session.producto = []                                     # :40-41 (sumaTotal is NOT unset)
        session.contador = 0
        return "El pedido se ha realizado con éxito"              # :43


def cents_total(store: Store, cart: List[str]) -> int:
    """Integer-cent restatement of carrito.php:12-18 for one cart."""
    return sum(to_cents(store.producto[c]) for c in cart if c in store.producto)


def md5(text: str) -> str:
    return hashlib.md5(text.encode()).hexdigest()


# ---- columnar view: what the batch engine is given ---------------------------------------

def columnar(store: Store, n_customers_dict: Optional[Dict[str, int]] = None):
    """Dictionary-encode the store's venta/detalle/producto rows into the engine's columns.

    Returns a dict with numpy arrays: order_id, product_idx, qty, order_customer,
    price_cents, plus the dictionaries used (codes, nits) and order_id_base.
    """
    import numpy as np
    codes = sorted(store.producto.keys())
    code_idx = {c: i for i, c in enumerate(codes)}
    price_cents = np.array([to_cents(store.producto[c]) for c in codes], dtype=np.int64)
    ventas = sorted(store.venta, key=lambda v: v["NumPedido"])
    if ventas:
        base = ventas[0]["NumPedido"]
        m = ventas[-1]["NumPedido"] - base + 1
    else:
        base, m = 5, 0
    nits = sorted({v["NIT"] for v in ventas})
    nit_idx = n_customers_dict or {n: i for i, n in enumerate(nits)}
    order_customer = np.zeros(m, dtype=np.int32)
    for v in ventas:
        order_customer[v["NumPedido"] - base] = nit_idx[v["NIT"]]
    det = sorted(store.detalle, key=lambda r: r[0])      # stable: keeps insertion order inside an order
    order_id = np.array([r[0] for r in det], dtype=np.int32)
    product_idx = np.array([code_idx.get(r[1], -1) for r in det], dtype=np.int32)
    qty = np.array([r[2] for r in det], dtype=np.int32)
    return dict(order_id=order_id, product_idx=product_idx, qty=qty, order_customer=order_customer,
                price_cents=price_cents, order_id_base=base, n_orders=m, n_customers=max(1, len(nit_idx)),
                codes=codes, nits=list(nit_idx.keys()))


# ---- array-level oracle (pure Python loops; small inputs only) ------------------------------

def aggregate_py(order_id, product_idx, qty, order_customer, order_id_base, n_customers, price_cents,
                 line_price=None, order_state=None, order_descuento=None, descuento_mode=0):
    """Row-at-a-time integer-cent aggregate with Python ints reduced mod 2^64.

    The optional columns are the order lifecycle (definitions in include/acmpc.h): the price a line was
    sold at (process/updateProduct.php:15 changes Precio later), whether the order still counts
    (process/updatePedido.php:11, process/delPedido.php:12-13) and venta.Descuento (store.sql:162)."""
    def wrap(v):
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    m = len(order_customer)
    p_n = len(price_cents)
    tot = [0] * m
    pq = [0] * p_n
    ct = [0] * n_customers
    counts = [True] * m if order_state is None else [bool(int(s) & 1) for s in order_state]
    for l in range(len(order_id)):
        o = int(order_id[l]) - order_id_base
        if o < 0 or o >= m:
            continue
        p = int(product_idx[l])
        if p < 0 or p >= p_n:
            continue
        q = int(qty[l])
        price = int(price_cents[p]) if line_price is None else int(line_price[l])
        tot[o] += price * q
        if counts[o]:
            pq[p] += q
    for o in range(m):
        if order_descuento is not None:
            gross = wrap(tot[o])
            d = int(order_descuento[o])
            if descuento_mode == 0:
                disc = 100 * d
            else:
                x = gross * d
                disc = (abs(x) + 50) // 100 * (1 if x >= 0 else -1)     # half away from zero
            tot[o] = gross - disc
        c = int(order_customer[o])
        if 0 <= c < n_customers and counts[o]:
            ct[c] += wrap(tot[o])
    return [wrap(v) for v in tot], [wrap(v) for v in pq], [wrap(v) for v in ct]
