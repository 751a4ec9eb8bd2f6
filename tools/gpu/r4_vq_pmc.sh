#!/bin/bash
# VQ main kernel: effective clock (GRBM_GUI_ACTIVE / duration) and matrix-pipe busy fraction from PMC counters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R="$PWD"
for v in ${VARIANTS:-0 29}; do
  rm -rf gpurun_out/pmc_vq
  DVQ_VQ_DBG=$v timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_vq" -o p -- python "$R/bench.py" --vq-only > gpurun_out/pmc_vq.log 2>&1
  cc=$(find gpurun_out/pmc_vq -name "*counter_collection.csv" | head -1); kt=$(find gpurun_out/pmc_vq -name "*kernel_trace.csv" | head -1)
  python - "$cc" "$kt" "$v" <<'P'
import csv,sys,collections
cc,kt,v=sys.argv[1:4]
dur={}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]]=(r["Kernel_Name"],int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    n=r["Kernel_Name"]
    if "rb2" not in n and "pipe_kernel" not in n: continue
    key="rb2" if "rb2" in n else "pipe8_f32"
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["_dur"].append(dur.get(r["Dispatch_Id"],("",0))[1])
for k,c in agg.items():
    m={n:sum(x)/len(x) for n,x in c.items()}
    d=m["_dur"]
    print(f"DBG={v} {k}: dur {d/1e3:.1f} us  GUI_ACTIVE {m.get('GRBM_GUI_ACTIVE',0):.0f} -> {m.get('GRBM_GUI_ACTIVE',0)/max(d,1):.2f} GHz  MFMA_BUSY {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g}  BUSY_CYCLES {m.get('SQ_BUSY_CYCLES',0):.3g}  WAVE_CYCLES {m.get('SQ_WAVE_CYCLES',0):.3g}  MOPS {m.get('SQ_INSTS_VALU_MFMA_MOPS_BF16',0):.3g}  WAIT_INST {m.get('SQ_WAIT_INST_ANY',0):.3g} ACTIVE_INST {m.get('SQ_ACTIVE_INST_ANY',0):.3g} VALU {m.get('SQ_INSTS_VALU',0):.3g}")
P
done
