#!/bin/bash
# rocprofv3 evidence for the bench command (run on the GPU box through gpurun):
#   pass 1: --kernel-trace --stats          -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE                -> HBM read traffic   (own run, no trace options)
#   pass 3: --pmc WRITE_SIZE                -> HBM write traffic  (own run)
# usage: tools/profile_bench.sh <tag> [bench args...]
# writes gpurun_out/<tag>/..., and a summary gpurun_out/<tag>/summary.json
set -u
TAG=${1:-prof}
shift || true
ARGS=${*:---steps 3 --warmup 1 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python3 bench.py $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- \
    python3 bench.py $ARGS > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- \
    python3 bench.py $ARGS > "$OUT/write.log" 2>&1 || { echo "write pass failed"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
def short(k):
    return k[k.index("hf_k_"):].split("(")[0].split("<")[0]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hf_k_" in r["Name"]:
            res[short(r["Name"])].update(calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]),
                                         min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]))
for name, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "hf_k_" in r["Kernel_Name"] and r["Counter_Name"] == key:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][key + "_KiB_per_launch"] = sum(v) / len(v)
# the workload the counters belong to (bench.py matches on it before quoting them)
import re
m = re.search(r'"elements_per_gpu": (\d+)', open(out + "/trace.log").read())
if m:
    res["elements_per_gpu"] = int(m.group(1))
# ... and the build they were taken with (bench.py quotes them for this build only)
m = re.search(r'"library_sha256_16": "([0-9a-f]+)"', open(out + "/trace.log").read())
if m:
    res["library_sha256_16"] = m.group(1)
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
# keep the merged output small: the raw traces are large
find "$OUT" -name "*kernel_trace.csv" -delete
