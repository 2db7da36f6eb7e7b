"""Extract the seed rows the checkout depends on from the reference's store.sql into store_seed.json.

Run in the build container (the reference is mounted at /root/reference there):
    python tests/golden/make_seed_fixture.py
The output is data only: the `producto` and `cliente` INSERT rows (store.sql:88-89,125-128) and the
AUTO_INCREMENT start of `venta` (store.sql:231-232).
"""
import json
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/store.sql"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "store_seed.json")


def rows_of(sql: str, table: str):
    m = re.search(r"INSERT INTO `%s` \(([^)]*)\) VALUES\s*(.*?);\s*\n" % table, sql, re.S)
    cols = [c.strip(" `") for c in m.group(1).split(",")]
    out = []
    for tup in re.findall(r"\((.*?)\)(?:,|$)", m.group(2), re.S):
        vals = [v.strip().strip("'") for v in re.findall(r"'(?:[^']*)'|[^,]+", tup)]
        out.append(dict(zip(cols, vals)))
    return out


def main():
    sql = open(SRC, encoding="latin-1").read()
    seed = {
        "source": "store.sql:88-89,125-128,231-232",
        "producto": [{k: r[k] for k in ("CodigoProd", "NombreProd", "Precio", "Stock")} for r in rows_of(sql, "producto")],
        "cliente": [{k: r[k] for k in ("id_Cliente", "Usuario", "Clave")} for r in rows_of(sql, "cliente")],
        "venta_auto_increment": int(re.search(r"MODIFY `NumPedido`.*AUTO_INCREMENT=(\d+)", sql).group(1)),
    }
    with open(OUT, "w") as f:
        json.dump(seed, f, indent=1)
    print(json.dumps(seed, indent=1))


if __name__ == "__main__":
    main()
