"""Builds libimgproc_hip.so for gfx950 with hipcc (cross-compiles without a GPU), in-tree."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libimgproc_hip.so")
SOURCES = ["api.hip", "kernels.hip", "fused.hip", "morphbits.hip", "gauss.hip"]
HEADERS = ["kernels.h", "fused.h", "morphbits.h", os.path.join("..", "..", "include", "imgproc.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or _stale():
        os.makedirs(LIBDIR, exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_node_sim(force)
    return LIB


NODE_SIM = os.path.join(LIBDIR, "node_sim")
HOST = os.path.join(HERE, "host")


def build_node_sim(force: bool = False) -> str:
    """The ROS-free node simulator (host/node_sim.cpp): plain g++ against the C ABI, no HIP headers needed."""
    srcs = [os.path.join(HOST, "node_sim.cpp"), os.path.join(HOST, "node_shim.hpp"), LIB]
    if force or not os.path.exists(NODE_SIM) or any(os.path.getmtime(f) > os.path.getmtime(NODE_SIM) for f in srcs):
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-Wextra", "-o", NODE_SIM, srcs[0], "-L", LIBDIR,
                               "-limgproc_hip", "-Wl,-rpath,$ORIGIN"])
    return NODE_SIM


if __name__ == "__main__":
    print(build(force=True, verbose=True))
