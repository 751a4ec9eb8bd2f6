"""Lint: does any s_barrier of a kernel that uses LDS-DMA (buffer_load / global_load ... lds) sit behind DMA pieces the wave has not
waited for?

Why: __syncthreads() is a workgroup-scope fence + s_barrier, and on gfx950 that fence waits for LDS traffic (lgkmcnt) only.  A wave
that passes the barrier with its own DMA pieces still in flight publishes nothing: the other waves' ds_reads behind the barrier can
overtake the pieces.  Whether the compiler ALSO emits a vmcnt wait there depends on its alias guess for the LDS accesses that follow;
the fp32 D = 64 VQ kernel had one stage body of its loop without any, and returned run-to-run different indices once a second process
shared the GPU (tools/debug/race_hunt.py).  Every barrier that publishes DMA'd data therefore carries an explicit s_waitcnt vmcnt;
this script checks the generated code.

Method: device assembly of each source (same flags as the build), per kernel a linear walk that keeps the queue of outstanding
vector-memory operations (vmcnt counts loads, stores and atomics in issue order on gfx9-family parts); `s_waitcnt vmcnt(n)` keeps the
last n.  Loops: the state at a backward branch is merged into its target label and the walk repeated until nothing changes.  A barrier
reached with a DMA operation still in the queue is reported.  Barriers that are MEANT to leave DMA in flight (LDS-only barriers, e.g.
the halo convolution's bias publication) are listed in ALLOW with the reason.

    python tools/lint_dma_barriers.py [file.hip ...]        exit code 1 if an unexpected barrier is found"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from dynamicvectorquantization_amd import build as B      # noqa: E402

# (source, kernel-name substring, number of such barriers, reason): barriers that may be reached with DMA in flight -- reviewed by hand
ALLOW = [
    ("conv_halo.hip", "conv3x3_halo_kernel", 1,
     "the bias-publication barrier ahead of the main loop is LDS-only on purpose (inline s_waitcnt lgkmcnt(0) + s_barrier): the first halo / "
     "weight DMA stays in flight across it and is waited for by the chunk barrier (dvq_dma_barrier) that follows"),
    ("igemm.hip", "gemm_nt_8phase_kernel", 10,
     "by design: the 8-phase main loop never drains the DMA queue.  A half-tile is published by the counted s_waitcnt vmcnt(6) of phase 3 "
     "(all but the three youngest half-tiles have landed) followed by TWO barriers before its first ds_read; the other barriers of a K tile "
     "separate read and MFMA sections only.  The queue is drained (vmcnt(0)) ahead of the epilogue's barrier."),
    ("igemm.hip", "gemm_tn_8phase_kernel", 10,
     "by design, as gemm_nt_8phase_kernel: counted s_waitcnt vmcnt(6) once per K tile, two barriers between that wait and the first transpose "
     "read of the half-tiles it retires, vmcnt(0) ahead of the barrier that ends the loop"),
    ("conv_halo.hip", "conv3x3_halo_wgrad_kernel", 1,
     "false positive of the path merge: the barrier behind the fused GroupNorm pass runs only when gn_ss != nullptr, the prefetch issue "
     "ahead of it only when gn_ss == nullptr (in the GroupNorm mode the next tile is issued BEHIND that barrier)"),
]

VM = re.compile(r"^\s*(buffer_|global_|flat_|scratch_)(load|store|atomic)")
WAIT = re.compile(r"^\s*s_waitcnt\b(.*)")
VMCNT = re.compile(r"vmcnt\((\d+)\)")
LABEL = re.compile(r"^(\.LBB[0-9_]+):")
BRANCH = re.compile(r"^\s*s_c?branch\w*\s+(\.LBB[0-9_]+)")
FUNC = re.compile(r"^(_Z\w+):")


def asm_of(src):
    out = os.path.join(tempfile.gettempdir(), "dvq_lint_" + src.replace(".hip", ".s"))
    s = os.path.join(B.CSRC, src)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s), os.path.getmtime(os.path.join(B.CSRC, "dvq_common.h"))):
        cmd = [B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["--cuda-device-only", "-S", s, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
    return open(out).read().splitlines()


def merge(a, b):
    """conservative join of two queues (lists of 'd' / 'o'): position-wise from the newest end, DMA wins"""
    if a is None:
        return list(b)
    n = max(len(a), len(b))
    pa, pb = ["-"] * (n - len(a)) + a, ["-"] * (n - len(b)) + b
    return [("d" if "d" in (x, y) else "o") for x, y in zip(pa, pb)]


def walk(lines):
    """-> list of (line number, pending DMA count, queue length) for barriers reached with DMA outstanding"""
    seeds = {}
    report = {}
    for _ in range(6):
        q = []
        changed = False
        labels_seen = {}
        for n, l in enumerate(lines):
            m = LABEL.match(l)
            if m:
                labels_seen[m.group(1)] = n
                if m.group(1) in seeds:
                    q = merge(q, seeds[m.group(1)])
                continue
            if VM.match(l):
                q.append("d" if re.search(r"\blds\b", l) else "o")
                q = q[-64:]
                continue
            m = WAIT.match(l)
            if m:
                v = VMCNT.search(m.group(1))
                if v:
                    k = int(v.group(1))
                    q = q[len(q) - k:] if k < len(q) else q
                    if k == 0:
                        q = []
                continue
            if re.match(r"^\s*s_barrier\b", l):
                nd = q.count("d")
                if nd:
                    report[n] = (nd, len(q))
                elif n in report:
                    pass
                continue
            m = BRANCH.match(l)
            if m:
                t = m.group(1)
                new = merge(seeds.get(t), q)
                if new != seeds.get(t):
                    seeds[t] = new
                    changed = True
                if re.match(r"^\s*s_branch\b", l):
                    q = []          # unconditional: the fall-through is reached from elsewhere (its label merges the seeds)
        if not changed:
            break
    return sorted((n, nd, ql) for n, (nd, ql) in report.items())


def main():
    sources = sys.argv[1:] or [s for s in B.SOURCES if "load_lds" in open(os.path.join(B.CSRC, s)).read()]
    bad = 0
    import concurrent.futures as cf
    with cf.ThreadPoolExecutor(max_workers=len(sources)) as ex:          # the device assemblies are independent hipcc runs
        asm = dict(zip(sources, ex.map(asm_of, sources)))
    for src in sources:
        lines = asm[src]
        starts = [(i, FUNC.match(l).group(1)) for i, l in enumerate(lines) if FUNC.match(l)]
        for j, (i0, name) in enumerate(starts):
            i1 = starts[j + 1][0] if j + 1 < len(starts) else len(lines)
            body = lines[i0:i1]
            end = next((k for k, l in enumerate(body) if "s_endpgm" in l), len(body))
            body = body[:end]
            if not any(re.search(r"\blds\b", l) and VM.match(l) for l in body):
                continue
            nb = sum(1 for l in body if re.match(r"^\s*s_barrier\b", l))
            rep = walk(body)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:150]
            allowed = next((why for s, sub, cnt, why in ALLOW if s == src and sub in dem and len(rep) <= cnt), None)
            tag = "ok" if not rep else ("allowed: " + allowed if allowed else "CHECK")
            print(f"{src}: {dem}: {nb} barriers, {len(rep)} reached with DMA outstanding [{tag}]")
            for n, nd, ql in rep:
                print(f"      asm line +{n}: {nd} DMA of {ql} outstanding vm ops")
            if rep and not allowed:
                bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
