"""Shared helpers of the ingest tests: an independent (regex) reference parser and text generators."""
import re

import numpy as np

_INT = r"(?:(?P<q{n}>['\"])(?P<i{n}>-?\d{{1,10}})(?P=q{n})|(?P<j{n}>-?\d{{1,10}}))"
_CODE = r"(?:(?P<cq>['\"])(?P<c1>[^\n]*?)(?P=cq)(?!(?P=cq))|(?P<c2>(?!['\"])[^,\t \r\n)]*))"
_ROW = re.compile(r"\(? *" + _INT.format(n=1) + r" *[,\t] *" + _CODE + r" *[,\t] *" + _INT.format(n=2) +
                  r" *\)?[,;]?\r?$")


# "insert" + anything without a VALUES( in it + VALUES + spaces + "(": the first such parenthesis
_STMT = re.compile(rb"(?i:insert)(?:(?!(?i:values) *\()[^\n])*?(?i:values) *\(")


def code_of(p: int) -> str:
    return str(p + 1) if p < 3 else f"SKU-{p:05d}"


def reference_parse(text: bytes, codes):
    """Row-at-a-time regex restatement of the grammar in include/acmpc.h (independent of the C oracle)."""
    idx = {c.encode("latin-1") if isinstance(c, str) else c: i for i, c in enumerate(codes)}
    rows, bad = [], 0
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines = lines[:-1]
    for ln in lines:
        if not ln:
            continue
        c0 = ln[:1]
        data = c0.isdigit() or c0 in (b"(", b"'", b'"') or (c0 == b"-" and ln[1:2].isdigit())
        if not data:
            # the one-line statement of library/consulSQL.php:30: the row starts at the '(' after VALUES
            m = _STMT.match(ln) if ln[:6].lower() == b"insert" else None
            if m is None or m.end() - 1 > 100:
                continue
            ln = ln[m.end() - 1:]
        s = ln.decode("latin-1")
        m = _ROW.match(s) if len(ln) <= 120 else None
        ok = m is not None
        if ok:
            a = int(m.group("i1") or m.group("j1"))
            q = int(m.group("i2") or m.group("j2"))
            code = m.group("c1") if m.group("cq") else m.group("c2")
            if m.group("cq") and m.group("cq") in code:
                ok = False                     # a closing quote followed by more text: not this grammar
            if not (-2**31 <= a < 2**31 and -2**31 <= q < 2**31) or len(code.encode("latin-1")) > 30:
                ok = False
        if ok:
            rows.append((a, idx.get(code.encode("latin-1"), -1), q))
        else:
            rows.append((0, -1, 0))
            bad += 1
    arr = np.array(rows, dtype=np.int64).reshape(-1, 3)
    return arr[:, 0].astype(np.int32), arr[:, 1].astype(np.int32), arr[:, 2].astype(np.int32), bad


def render(oid, pidx, qty, style: str, rng=None) -> bytes:
    """Columns -> text in one of the accepted shapes."""
    out = []
    if style == "dump":
        for s in range(0, len(oid), 1000):
            out.append("INSERT INTO `detalle` (`NumPedido`, `CodigoProd`, `CantidadProductos`) VALUES")
            chunk = [f"({a}, '{code_of(p) if p >= 0 else 'NOPE'}', {q})" for a, p, q in zip(oid[s:s + 1000], pidx[s:s + 1000], qty[s:s + 1000])]
            out.append(",\n".join(chunk) + ";")
        return ("\n".join(out) + "\n").encode("latin-1")
    if style == "csv":
        return ("".join(f"{a},{code_of(p) if p >= 0 else 'NOPE'},{q}\n" for a, p, q in zip(oid, pidx, qty))).encode("latin-1")
    if style == "tsv_crlf":
        return ("".join(f"{a}\t{code_of(p) if p >= 0 else 'NOPE'}\t{q}\r\n" for a, p, q in zip(oid, pidx, qty))).encode("latin-1")
    if style == "quoted":
        body = "".join(f"('{a}', \"{code_of(p) if p >= 0 else 'NOPE'}\", '{q}')\n" for a, p, q in zip(oid, pidx, qty))
        return ("insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES\n" + body).rstrip("\n").encode("latin-1")
    if style == "statements":           # one statement per line, exactly as library/consulSQL.php:30 builds it
        return ("".join(f"insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES('{a}', '{code_of(p) if p >= 0 else 'NOPE'}', '{q}')\n"
                        for a, p, q in zip(oid, pidx, qty))).encode("latin-1")
    raise ValueError(style)


def statement_prefix(rng) -> str:
    """A random front part of a statement line: mostly the reference's own (library/consulSQL.php:30), sometimes
    other letter cases, spacing, paddings that push the '(' past the 100-byte reach, or no VALUES( at all."""
    r = rng.random()
    if r < 0.5:
        return "insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES"
    if r < 0.65:
        return "INSERT INTO `detalle` (`NumPedido`, `CodigoProd`, `CantidadProductos`) VALUES" + " " * int(rng.integers(0, 30))
    if r < 0.75:
        return "Insert into detalle" + " " * int(rng.integers(1, 90)) + "values "
    if r < 0.85:
        return "insert into detalle (a values (b) c) VALUES "      # the first VALUES( wins
    if r < 0.92:
        return "insertvalues"
    return "insert into detalle (NumPedido) value"                  # not a statement row: skipped


EDGE_TEXT = (
    b"-- comment line\n"
    b"INSERT INTO `detalle` (`NumPedido`, `CodigoProd`, `CantidadProductos`) VALUES\n"
    b"(5, '1', 1),\n"
    b"(5,'2',1),\n"
    b"( 5 , '3' , 1 ) ,\n"                      # malformed: space before the trailing comma
    b"(6, 'SKU-00003', 2);\n"
    b"\n"
    b"7,1,3\n"
    b"7\t2\t-4\r\n"
    b"'8', '3', '1'\n"
    b"(9, 'unknown code', 1),\n"
    b"(10, '', 1),\n"                            # empty code: unknown
    b"(11, 'abcdefghijklmnopqrstuvwxyz0123', 1),\n"      # 30 bytes: fine, unknown
    b"(12, 'abcdefghijklmnopqrstuvwxyz01234', 1),\n"     # 31 bytes: malformed
    b"(13, 'it''s', 1),\n"                       # doubled quote: malformed
    b"(14, '1', 99999999999),\n"                 # 11 digits: malformed
    b"(15, '1', 2147483648),\n"                  # does not fit int32: malformed
    b"(-2147483648, '1', 2147483647),\n"
    b"(16, '1' 1),\n"                            # missing separator: malformed
    b"(17, '1', 1) trailing\n"                   # malformed
    b"x,1,1\n"                                   # not a data row: skipped
    b"-- 18,1,1\n"                               # comment: skipped
    b"-5,2,1\n"
    b"insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES('20', '1', '1')\n"   # library/consulSQL.php:30
    b"INSERT INTO `detalle` (`NumPedido`, `CodigoProd`, `CantidadProductos`) VALUES (21, '2', 3);\r\n"
    b"insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES('22', '1' '1')\n"    # malformed row in a statement
    b"insert into detalle (NumPedido) select 23\n"                                             # no VALUES(: skipped
    b"Insert Into detalle Values   ('24',\"3\",'2')\n"
    b"(19, \"2\", 7)"                            # last row without newline
)
EDGE_CODES = ["1", "2", "3", "SKU-00003"]
EDGE_EXPECT = [(5, 0, 1), (5, 1, 1), (0, -1, 0), (6, 3, 2), (7, 0, 3), (7, 1, -4), (8, 2, 1), (9, -1, 1), (10, -1, 1),
               (11, -1, 1), (0, -1, 0), (0, -1, 0), (0, -1, 0), (0, -1, 0), (-2147483648, 0, 2147483647), (0, -1, 0),
               (0, -1, 0), (-5, 1, 1), (20, 0, 1), (21, 1, 3), (0, -1, 0), (24, 2, 2), (19, 1, 7)]
EDGE_BAD = 8
