"""Builds libimgproc_hip.so for gfx950 with hipcc (cross-compiles without a GPU), in-tree."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libimgproc_hip.so")
SOURCES = ["api.hip", "kernels.hip", "fused.hip", "fused_inst_plain_small.hip", "fused_inst_plain_large.hip",
           "fused_inst_bayer_small.hip", "fused_inst_bayer_large.hip", "morphbits.hip", "gauss.hip", "gauss_mfma.hip", "gauss_fused.hip"]
HEADERS = ["kernels.h", "fused.h", "fused_kernel.hpp", "hsv_px.hpp", "morphbits.h", os.path.join("..", "..", "include", "imgproc.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-pthread"]
# per-source extras.  gauss_mfma.hip: MFMA results in VGPRs (the allocator's default put the accumulators into AGPRs and paid
# 48 v_accvgpr_read per 16-row block, a quarter of the kernel's vector instructions, to get them back out)
EXTRA_FLAGS = {"gauss_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "gauss_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


OBJDIR = os.path.join(LIBDIR, "obj")


def _newer(path: str, deps) -> bool:
    """True when `path` is missing or older than any of `deps`."""
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """One hipcc -c per source (in parallel, only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append([hipcc] + [f for f in FLAGS if f != "-shared"] + EXTRA_FLAGS.get(src, []) +
                        ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), int(os.environ.get("MAX_JOBS", "6")))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _newer(LIB, objs):
        run([hipcc] + FLAGS + ["-o", LIB] + objs)
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
