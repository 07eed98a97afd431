#!/bin/bash
# Everything profiles/<tag>_* holds for a round, in one go on the GPU box (repository root): tools/collect_round_end.sh <tag>
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
bash tools/collect_profiles.sh "$TAG" > "$OUT/${TAG}_collect.log" 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench_config3_steps20_warmup5.log" 2>&1
timeout 600 python bench.py --workload config4 --steps 60 --warmup 20 > "$OUT/${TAG}_bench_config4_1gpu.log" 2>&1
{ python tools/small_bench.py; python tools/small_bench2.py; } > "$OUT/${TAG}_small_configs.log" 2>&1
bash tools/experiments/run_timeline.sh > /dev/null 2>&1; cp "$OUT/step_timeline.md" "$OUT/${TAG}_step_timeline_body.md"
bash tools/collect_pmc.sh > "$OUT/${TAG}_pmc.log" 2>&1
for f in bench_config3 bench_config3_steps20_warmup5 bench_config4_1gpu bench_config5; do tail -1 "$OUT/${TAG}_$f.log" | cut -c1-220; done
