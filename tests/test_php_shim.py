"""The PHP FFI shim cannot run here (no PHP interpreter): at least keep its C declarations in step with
include/acmpc.h — every function it declares must exist in the header with the same parameter types."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _prototypes(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"\b(?:int|void|const char\*)\s+(acmpc_\w+)\s*\(([^)]*)\)\s*;", text):
        params = []
        for p in m.group(2).split(","):
            p = " ".join(p.split())
            if p in ("", "void"):
                continue
            # drop the parameter name, keep the type
            params.append(re.sub(r"\s*\b\w+$", "", p).replace(" *", "*").replace("* ", "*"))
        out[m.group(1)] = params
    return out


def test_php_cdef_matches_the_header():
    header = _prototypes(open(os.path.join(ROOT, "include", "acmpc.h")).read())
    php = open(os.path.join(ROOT, "ac_mpc_amd", "php", "consulSQL_lote.php")).read()
    cdef = php[php.index('const CDEF = "') + len('const CDEF = "'):]
    cdef = cdef[:cdef.index('";')]
    shim = _prototypes(cdef)
    assert len(shim) >= 9
    for name, params in shim.items():
        assert name in header, name
        assert params == header[name], (name, params, header[name])
    # every function the class calls through FFI is declared in its CDEF
    for called in set(re.findall(r"->(acmpc_\w+)\(", php)):
        assert called in shim, called
