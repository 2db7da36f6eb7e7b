"""Build the native pieces in-tree: libacmpc.so (HIP, gfx950), the CPU oracle and the developer tool.

Run as ``python build_native.py`` or through ``__graft_entry__.build()``.  It deliberately lives outside
the ac_mpc_amd package: importing the package loads libacmpc.so, which is what this script creates.  hipcc cross-compiles
for gfx950 without a GPU, so this works on a CPU-only machine.
"""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "ac_mpc_amd")
LIB = os.path.join(PKG, "lib", "libacmpc.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")
TOOL = os.path.join(ROOT, "tools", "acmpc_check")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden"]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=cwd)


def build_library(force: bool = False) -> str:
    csrc = os.path.join(PKG, "csrc")
    srcs = [os.path.join(csrc, "acmpc.hip")]
    srcs += sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hpp"))
    srcs.append(os.path.join(ROOT, "include", "acmpc.h"))
    if force or not _newer(LIB, srcs):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        _run([HIPCC, *HIP_FLAGS, "-shared", "-o", LIB, srcs[0]])
    return LIB


def build_oracle(force: bool = False) -> str:
    if force or not _newer(ORACLE_LIB, [os.path.join(ORACLE_DIR, "aggregate_oracle.c"),
                                        os.path.join(ORACLE_DIR, "ingest_oracle.c"),
                                        os.path.join(ORACLE_DIR, "emit_oracle.c"),
                                        os.path.join(ORACLE_DIR, "Makefile")]):
        _run(["make", "-C", ORACLE_DIR, "-B", "liboracle.so"])
    return ORACLE_LIB


def build_tool(force: bool = False) -> str:
    src = os.path.join(ROOT, "tools", "acmpc_check.cpp")
    if force or not _newer(TOOL, [src, LIB, ORACLE_LIB]):
        _run([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", TOOL, src,
              "-L" + os.path.dirname(LIB), "-lacmpc", "-L" + ORACLE_DIR, "-loracle",
              "-Wl,-rpath,$ORIGIN/../ac_mpc_amd/lib", "-Wl,-rpath,$ORIGIN/../oracle"])
    return TOOL


def build_all(force: bool = False) -> None:
    build_library(force)
    build_oracle(force)
    build_tool(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
