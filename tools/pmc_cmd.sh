#!/bin/bash
# SQ / TCC counters and kernel durations for one probe command, each counter set in a rocprofv3
# run of its own (no trace option is ever combined with --pmc), plus one --kernel-trace --stats run.
# usage: tools/pmc_cmd.sh <outdir> <python script> [args...]      (run on the GPU box)
# writes <outdir>/summary.txt: one line per hf_k_* kernel (template arguments kept) with the
# summed counters over its dispatches, the dispatch count and the average duration.
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 "$@" > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -n 5 "$OUT/trace.log"; exit 1; }
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $SET --output-format csv -d "$OUT/p$i" -- \
      python3 "$@" > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -n 5 "$OUT/p$i.log"; exit 1; }
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_MISC
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def short(k):
    k = k[k.index("hf_k_"):]
    return k.split("(")[0].replace("(anonymous namespace)::", "").replace(" ", "")
agg = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hf_k_" not in k:
            continue
        name = short(k)
        c = r["Counter_Name"]
        agg[name][c] = agg[name].get(c, 0.0) + float(r["Counter_Value"])
        seen[(name, c)] += 1
    for (name, c), n in seen.items():
        agg[name]["_dispatches"] = n
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hf_k_" in r["Name"]:
            agg[short(r["Name"])].update(_calls=float(r["Calls"]), _avg_ns=float(r["AverageNs"]),
                                         _min_ns=float(r["MinNs"]))
with open(out + "/summary.txt", "w") as fo:
    fo.write("# " + " ".join(sys.argv[1:]) + "\n")
    for k, v in sorted(agg.items()):
        line = k + " " + " ".join(f"{a}={b:.0f}" for a, b in sorted(v.items()))
        print(line)
        fo.write(line + "\n")
PY
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
