"""Write tests/golden/detalle_dump.sql: the rows of synth_1k_zipf.npz as a phpMyAdmin-style dump of `detalle`.

    python tests/golden/make_dump.py

The layout copies the reference's own dump, store.sql:88-89,125-128 (header line, one row per line, ",\\n"
between rows, ";" after the last), split into several INSERT statements as phpMyAdmin does for long tables.
The reference ships no `detalle` rows, so the rows are the synthetic ones; product codes are the seed codes
'1','2','3' (store.sql:126-128) followed by 'SKU-00003', ...  A second file, detalle_insert_stmt.sql, holds
the first 40 rows as the literal statements process/confirmcompra.php:36 sends through
library/consulSQL.php:30, one statement per line (the form a query log holds them in): every value quoted,
CantidadProductos always '1', no terminator.
"""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def code_of(p: int) -> str:
    return str(p + 1) if p < 3 else f"SKU-{p:05d}"


def main():
    v = np.load(os.path.join(HERE, "synth_1k_zipf.npz"))
    oid, pidx, qty = v["order_id"], v["product_idx"], v["qty"]
    out = ["-- phpMyAdmin SQL Dump", "--", "-- Volcado de datos para la tabla `detalle`", "--", ""]
    for s in range(0, oid.size, 300):
        out.append("INSERT INTO `detalle` (`NumPedido`, `CodigoProd`, `CantidadProductos`) VALUES")
        rows = [f"({int(a)}, '{code_of(int(p))}', {int(q)})" for a, p, q in zip(oid[s:s + 300], pidx[s:s + 300], qty[s:s + 300])]
        out.append(",\n".join(rows) + ";")
        out.append("")
    open(os.path.join(HERE, "detalle_dump.sql"), "w", newline="\n").write("\n".join(out) + "\n")
    stmt = [f"insert into detalle (NumPedido, CodigoProd, CantidadProductos) VALUES('{int(a)}', '{code_of(int(p))}', '1')"
            for a, p in zip(oid[:40], pidx[:40])]
    open(os.path.join(HERE, "detalle_insert_stmt.sql"), "w", newline="\n").write("\n".join(stmt) + "\n")
    print("rows", oid.size)


if __name__ == "__main__":
    main()
