#!/bin/bash
# Everything profiles/ holds for one build, in one go (run on the GPU box through gpurun):
#   the headline bench under rocprofv3 (kernel stats, FETCH_SIZE, WRITE_SIZE: tools/profile_bench.sh),
#   SQ / TCC counter passes for the stream kernels on four distributions and for the per-page path
#   (tools/pmc_cmd.sh), and the bench lines of every configuration.
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -u
TAG=${1:-rXX}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
bash tools/profile_bench.sh ${TAG}_bench > gpurun_out/${TAG}_bench.log 2>&1 || { echo "profile_bench failed"; tail -n 5 gpurun_out/${TAG}_bench.log; exit 1; }
for k in gauss fp8 uniform geom; do
  bash tools/pmc_cmd.sh gpurun_out/${TAG}_pmc_$k tools/kprobe.py --elems 1073741824 --reps 1 --kinds $k > gpurun_out/${TAG}_pmc_$k.log 2>&1 || { echo "pmc $k failed"; exit 1; }
done
bash tools/pmc_cmd.sh gpurun_out/${TAG}_pmc_blocks tools/kprobe_blocks.py --pages 65536 --reps 1 > gpurun_out/${TAG}_pmc_blocks.log 2>&1 || { echo "pmc blocks failed"; exit 1; }
# the paged streams (pages with records, pages without) beside the dense and slotted ones
bash tools/pmc_cmd.sh gpurun_out/${TAG}_pmc_pages_bf16 tools/kprobe_pages.py --kind bf16 --pages 131072 --no-read --reps 2 > gpurun_out/${TAG}_pmc_pages_bf16.log 2>&1 || { echo "pmc pages bf16 failed"; exit 1; }
bash tools/pmc_cmd.sh gpurun_out/${TAG}_pmc_pages_fp8 tools/kprobe_pages.py --kind fp8 --pages 65536 --no-read --reps 2 > gpurun_out/${TAG}_pmc_pages_fp8.log 2>&1 || { echo "pmc pages fp8 failed"; exit 1; }
timeout -k 10 300 python3 bench.py > gpurun_out/${TAG}_line_config2.json 2> gpurun_out/${TAG}_line_config2.err || { echo "bench failed"; exit 1; }
for c in 1 3 5; do
  timeout -k 10 300 python3 bench.py --config $c > gpurun_out/${TAG}_line_config$c.json 2> gpurun_out/${TAG}_line_config$c.err || { echo "bench config $c failed"; exit 1; }
done
timeout -k 10 300 python3 bench.py --elems 8819228672 --no-cpu-baseline > gpurun_out/${TAG}_line_config4_shard.json 2> gpurun_out/${TAG}_line_config4_shard.err || { echo "config 4 shard failed"; exit 1; }
timeout -k 10 120 python3 tools/kprobe.py --elems 1073741824 --reps 5 > gpurun_out/${TAG}_kprobe.txt 2>&1
timeout -k 10 120 python3 tools/kprobe_blocks.py --pages 65536 --reps 5 > gpurun_out/${TAG}_kprobe_blocks.txt 2>&1
timeout -k 10 120 python3 tools/kprobe_codebook.py --blocks 65536 --reps 5 > gpurun_out/${TAG}_kprobe_codebook.txt 2>&1
timeout -k 10 120 python3 tools/kprobe_fields.py > gpurun_out/${TAG}_kprobe_fields.txt 2>&1
timeout -k 10 120 python3 tools/kprobe_slotted.py --reps 5 > gpurun_out/${TAG}_kprobe_slotted.txt 2>&1
timeout -k 10 120 python3 tools/kprobe_pages.py --kind bf16 --pages 131072 --no-read > gpurun_out/${TAG}_kprobe_pages_bf16.txt 2>&1
timeout -k 10 120 python3 tools/kprobe_pages.py --kind fp8 --pages 65536 > gpurun_out/${TAG}_kprobe_pages_fp8.txt 2>&1
timeout -k 10 60 ./tools/sort_cycles > gpurun_out/${TAG}_sort_cycles.txt 2>&1
timeout -k 10 60 ./tools/memset_graph_repro > gpurun_out/${TAG}_memset_graph_repro.txt 2>&1
timeout -k 10 120 ./huffman_amd/lib/hf_shard_driver --gpus 1 --elems 268435456 > gpurun_out/${TAG}_shard_driver_n1.txt 2>&1
echo "profile round $TAG done"
