"""Build libdvq_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so lives next to this file so it
travels with the repo snapshot to the GPU box.

    python -m dynamicvectorquantization_amd.build [--force] [--probes]

`--probes` builds a SECOND library, libdvq_hip_probes.so, with -DDVQ_PROBES: the timing experiments that remove pieces of a kernel
(DVQ_HALO_DBG, DVQ_WGRAD_DBG, DVQ_ATTN_DBG, DVQ_VQ_DBG -- wrong results by construction) and the slower persistent convolution
(conv_halo2.hip, DVQ_HALO2=1).  `_lib.load()` takes it only when DVQ_USE_PROBES_LIB=1 (tools/debug/); the product library contains
none of that code and does not read those variables.
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
LIB_PROBES = os.path.join(HERE, "libdvq_hip_probes.so")
PROBE_SOURCES = ["conv_halo2.hip"]          # compiled and linked into the probe library only
SOURCES = ["vq.hip", "entropy.hip", "groupnorm.hip", "igemm.hip", "conv_halo.hip", "misc.hip", "lossnet.hip", "router.hip", "permuter.hip", "transformer.hip", "attention.hip", "attention2.hip", "imgproc.hip", "decode.hip", "cmdlist.hip"]
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


def build(force: bool = False, verbose: bool = True, probes: bool = False) -> str:
    obj_dir = OBJ + "_probes" if probes else OBJ
    lib = LIB_PROBES if probes else LIB
    # DVQ_BUILD_EXTRA_FLAGS: extra compiler flags for A/B builds on the GPU box (e.g. -DDVQ_STREAM_NT=0); use with --force
    flags = FLAGS + (["-DDVQ_PROBES"] if probes else []) + os.environ.get("DVQ_BUILD_EXTRA_FLAGS", "").split()
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "dvq_common.h"), os.path.join(HERE, "..", "include", "dvq_hip.h")]
    jobs = []
    objs = []
    for src in SOURCES + (PROBE_SOURCES if probes else []):
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + flags + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])

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
    if jobs or force or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        if verbose:
            print("[dvq build] linked", lib, flush=True)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, probes="--probes" in sys.argv)
