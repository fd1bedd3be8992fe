#!/bin/bash
# Collect SQ / TCC counters for the streaming kernels in separate rocprofv3 passes
# (one --pmc set per run; no trace options combined with --pmc).
# usage: tools/pmc.sh <outdir> [elems] [kinds]
set -u
OUT=${1:-gpurun_out/pmc}
ELEMS=${2:-268435456}
KINDS=${3:-gauss}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $SET --output-format csv -d "$OUT/p$i" -- \
      python3 tools/kprobe.py --elems "$ELEMS" --reps 1 --kinds "$KINDS" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; exit 1; }
done <<'EOF'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_MISC
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE
EOF
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hf_k_" not in k:
            continue
        name = k[k.index("hf_k_"):].split("(")[0].split("<")[0]
        agg[name][r["Counter_Name"]] = agg[name].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        agg[name]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open(out + "/summary.txt", "w") as fo:
    for k, v in agg.items():
        line = k + " " + " ".join(f"{a}={b:.0f}" for a, b in sorted(v.items()))
        print(line)
        fo.write(line + "\n")
PY
