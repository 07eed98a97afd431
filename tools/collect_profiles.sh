#!/bin/bash
# Round-end evidence for profiles/: rocprofv3 kernel-trace summaries of bench.py for both workloads and the plain bench lines.
# Usage (GPU box, repository root): tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-rXX}
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in config3 config5; do
	rm -rf "$OUT/prof_$wl"
	timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$wl" -o kt -- python "$REPO/bench.py" --workload $wl --steps 60 --warmup 120 --cpu-steps 0 > "$OUT/${TAG}_bench_under_rocprof_$wl.log" 2>&1
	db=$(find "$OUT/prof_$wl" -name "*.db" | head -1)
	{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --workload $wl --steps 60 --warmup 120 --cpu-steps 0 (from /tmp, TMPDIR=/tmp)"; echo;
	  echo "Summarised from the rocpd database with tools/rocpd_summary.py (all steps of the run: warm-up, timed, read-back and profiled steps)."; echo;
	  python "$REPO/tools/rocpd_summary.py" "$db"; } > "$OUT/${TAG}_kernel_stats_$wl.md"
	# the average kernel durations as bench.py reads them (roofline.frac_kernel_time): copy gpurun_out/kernel_time.json to profiles/ with the summary it belongs to
	if [ $wl = config3 ]; then python "$REPO/tools/rocpd_summary.py" "$db" --json "$OUT/kernel_time.json" --source "profiles/${TAG}_kernel_stats_config3.md (rocprofv3 --kernel-trace --stats -- python bench.py --workload config3 --steps 60 --warmup 120 --cpu-steps 0; tools/rocpd_summary.py --json)" > /dev/null; fi
	rm -rf "$OUT/prof_$wl"
done
cd "$REPO"
timeout 300 python bench.py > "$OUT/${TAG}_bench_config3.log" 2>&1
timeout 300 python bench.py --workload config5 > "$OUT/${TAG}_bench_config5.log" 2>&1
tail -1 "$OUT/${TAG}_bench_config3.log" | cut -c1-200
tail -1 "$OUT/${TAG}_bench_config5.log" | cut -c1-200
head -12 "$OUT/${TAG}_kernel_stats_config3.md"
