#!/bin/bash
# Copy what tools/profile_round.sh <tag> left under gpurun_out/ into profiles/ (run here, after
# gpurun has merged the files back).  usage: tools/collect_profiles.sh <tag>
set -eu
TAG=$1
cd "$(dirname "$0")/.."
cp gpurun_out/${TAG}_bench/summary.json profiles/${TAG}_bench_summary.json
cp "$(ls -t $(find gpurun_out/${TAG}_bench/trace -name '*kernel_stats.csv') | head -n 1)" profiles/${TAG}_bench_kernel_stats.csv
for k in gauss fp8 uniform geom blocks pages_bf16 pages_fp8; do
  cp gpurun_out/${TAG}_pmc_$k/summary.txt profiles/${TAG}_pmc_${k}_counters.txt
  cp "$(ls -t $(find gpurun_out/${TAG}_pmc_$k/trace -name '*kernel_stats.csv') | head -n 1)" profiles/${TAG}_pmc_${k}_kernel_stats.csv
done
for c in 1 2 3 5 4_shard; do cp gpurun_out/${TAG}_line_config$c.json profiles/${TAG}_bench_line_config$c.json; done
for f in kprobe kprobe_blocks kprobe_codebook kprobe_fields kprobe_slotted kprobe_pages_bf16 kprobe_pages_fp8 sort_cycles memset_graph_repro shard_driver_n1; do
  grep -v "amdgpu.ids" gpurun_out/${TAG}_$f.txt > profiles/${TAG}_$f.txt
done
ls -la profiles/${TAG}_*
