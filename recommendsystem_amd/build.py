"""In-tree build of librs_hip.so (gfx950 only).

    python -m recommendsystem_amd.build [--force]

Compiles every csrc/*.hip with hipcc for gfx950 and links one shared library exporting the
C-ABI declared in include/rs_hip.h.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "librs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-I", CSRC]
# Per-file additions.  interact.hip: MFMA results are consumed by vector instructions right away, so the products are
# written to ordinary vector registers instead of accumulation registers that have to be copied out first.
EXTRA_FLAGS = {"interact.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _newer(src: str, dst: str, deps) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src, *deps])


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o, hdrs):
            jobs.append([HIPCC, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
