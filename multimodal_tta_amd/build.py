"""Build libmmtta.so (HIP, gfx950 only) in-tree with hipcc.

``python -m multimodal_tta_amd.build`` or ``build_library()``.  Cross-compiles without a GPU.
The .so lands next to the sources (git-ignored, but it travels with gpurun snapshots).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmmtta.so")
SOURCES = ["api.hip", "conv_igemm.hip", "conv_direct.hip", "conv_wgrad.hip", "pointwise.hip", "loss_optim_metric.hip", "preproc.hip",
           "surface.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; this package is gfx950-only and has no CPU fallback)")


def _digest() -> str:
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, name), "rb") as fh:
                h.update(name.encode())
                h.update(fh.read())
    with open(os.path.join(os.path.dirname(HERE), "include", "mmtta.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_library(force: bool = False, verbose: bool = False) -> str:
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs: List[str] = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and (verbose or p.returncode != 0):
            print(f"--- {src}\n{out}", file=sys.stderr)
        failed = failed or p.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
