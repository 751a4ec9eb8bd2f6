#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/gpu_pmc_bench.sh into per-kernel HBM bytes per launch.

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Corrections of MI355X_MICROARCH.md ("HBM [CDNA4]"): on gfx950 FETCH_SIZE tallies
64 B per 128-B request of wide coalesced streaming reads -> x2.  WRITE_SIZE is "uncalibrated" there; it is calibrated here on
gn_apply_kernel, which writes exactly as many bytes as it reads (same shape, same dtype): the factor is printed and applied.
    python tools/pmc_summarise.py gpurun_out/pmcb_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmcb_WRITE_SIZE/p_counter_collection.csv out.json
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys


def csrc_sha16():
    """hash of the kernel sources this profile was taken with (bench.py compares it with the sources IT runs and says so in
    roofline.traffic_source: the traffic figure is a committed profile, not an in-run measurement)"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynamicvectorquantization_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        agg[name].append(float(r["Counter_Value"]) * 1024.0)
    return agg


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    # (round 6: the kernel has a nontemporal variant for operands larger than the Infinity Cache; calibrate on the plain one when
    #  the run has it -- cache-resident operands -- else on the streaming one)
    cal = next((c for c in ("gn_apply_kernel<unsigned short, 1, false>", "gn_apply_kernel<unsigned short, 1, true>",
                            "gn_apply_kernel<unsigned short, 1>") if c in fetch and c in write), "gn_apply_kernel<unsigned short, 1>")
    wf = 1.0
    if cal in fetch and cal in write:
        wf = (2.0 * sum(fetch[cal])) / sum(write[cal])
    out = {"_method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0`; "
                      "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE x%.4f (calibrated on %s)" % (wf, cal),
           "csrc_sha16": csrc_sha16(), "commit": os.environ.get("HEAD_SHA", ""), "kernels": {}}
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        n = len(fetch[k])
        fb = 2.0 * sum(fetch[k]) / n
        wb = wf * sum(write.get(k, [0.0])) / max(1, len(write.get(k, [0.0])))
        out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb),
                             "hbm_bytes_per_launch": round(fb + wb)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k in list(out["kernels"])[:8]:
        print(k, out["kernels"][k])
    print("write calibration factor", wf)


if __name__ == "__main__":
    main()
