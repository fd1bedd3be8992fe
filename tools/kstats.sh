#!/bin/bash
# per-kernel durations of one probe command under rocprofv3 (run on the GPU box through gpurun)
# usage: tools/kstats.sh <tag> <python script and args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -- python3 "$@" > gpurun_out/$TAG.log 2>&1 || echo "(the probe exited with an error; kernel stats below cover what ran)"
python3 - "$TAG" <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if "hf_k_" in r["Name"]:
            n = r["Name"]; n = n[n.index("hf_k_"):].split("(")[0]
            print("%-45s calls=%4s avg_us=%9.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
