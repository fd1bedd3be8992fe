#!/bin/bash
# per-dispatch counters of the page encoder over freshly allocated payload buffers (tools/kprobe_place_pmc.py)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/place_pmc
mkdir -p $OUT
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python3 tools/kprobe_place_pmc.py $((i+20)) > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -n 5 $OUT/p$i.log; exit 1; }
  grep round $OUT/p$i.log | head -3
done <<'SETS'
GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TCC_WRITE_REQ_LATENCY_sum
GRBM_GUI_ACTIVE TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TCC_READ_REQ_LATENCY_sum
SETS
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "hf_k_encode_pages" not in r["Kernel_Name"]:
            continue
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print(f)
    for k, (d, c) in enumerate(rows.items()):
        if k % 3 == 2:
            print(f"  round {k // 3:2d} " + " ".join(f"{a}={b:.4g}" for a, b in sorted(c.items())))
PY
find $OUT -name "*counter_collection.csv" -size +3M -delete
