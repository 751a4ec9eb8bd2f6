"""Build libdvq_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so lives next to this file so it
travels with the repo snapshot to the GPU box.

    python -m dynamicvectorquantization_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdvq_hip.so")
SOURCES = ["vq.hip", "entropy.hip", "groupnorm.hip", "igemm.hip", "conv_halo.hip", "conv_halo2.hip", "misc.hip", "lossnet.hip", "router.hip", "permuter.hip", "transformer.hip", "attention.hip", "imgproc.hip", "decode.hip", "cmdlist.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
# per-file additions.  vq.hip: no SLP vectorisation -- it pairs the scalar fp32 bookkeeping between the MFMAs of the argmin main loop
# into v_pk_fma_f32 / v_pk_add_f32, which issue more slowly beside a busy matrix pipe than the two instructions they replace
EXTRA_FLAGS = {"vq.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libdvq_hip.so cannot be built on this machine")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "dvq_common.h"), os.path.join(HERE, "..", "include", "dvq_hip.h")]
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return cmd[-1]

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for done in ex.map(run, jobs):
                if verbose:
                    print("[dvq build] compiled", os.path.basename(done), flush=True)
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print("[dvq build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
