"""Developer tool: does the library initialise whatever the order in which it and torch first touch the
GPU?  usage: probe_load.py mine | torch_first | mine_first | load_then_torch_then_init"""
import sys, ctypes as C, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    raise SystemExit(__doc__)
mode = sys.argv[1]
def mine():
    from ac_mpc_amd import _lib
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.acmpc_init(C.byref(ctx), 0)
    print(mode, "acmpc_init rc", rc, lib.acmpc_last_error(None).decode())
    if rc == 0: lib.acmpc_destroy(ctx)
def tor():
    import torch
    print(mode, "torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    if torch.cuda.is_available():
        x = torch.ones(4, device="cuda"); print(float(x.sum()))
if mode == "mine": mine()
elif mode == "torch_first": tor(); mine()
elif mode == "mine_first": mine(); tor()
elif mode == "load_then_torch_then_init":
    from ac_mpc_amd import _lib; _lib.load(); tor(); mine()
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print(mode, libs)
