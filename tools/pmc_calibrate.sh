#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE per ACCESS PATTERN on this GPU (VERDICT r05 task 2b): runs tools/pmc_calibrate (known byte
# counts, buffer >> Infinity Cache) under counters-only passes (one --pmc set per pass, --kernel-trace only) and writes
# gpurun_out/pmc_calibration.md + gpurun_out/pmc_calibration.json (copy to profiles/; tools/pmc_summary.py reads the json).
# Usage (GPU box, repository root): bash tools/pmc_calibrate.sh [MiB]
REPO=$PWD
OUT=$REPO/gpurun_out/pmc_cal
MIB=${1:-2048}
rm -rf "$OUT"; mkdir -p "$OUT"
[ -x "$REPO/tools/pmc_calibrate" ] || hipcc --offload-arch=gfx950 -O3 "$REPO/tools/pmc_calibrate.hip" -o "$REPO/tools/pmc_calibrate" || exit 1
cd /tmp && export TMPDIR=/tmp
"$REPO/tools/pmc_calibrate" $MIB 3 > "$OUT/plain.txt" 2>&1
cat "$OUT/plain.txt"
# one counter set per pass (the TCC block holds four counters at a time; FETCH_SIZE alone takes three)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"; do
	i=$((i + 1))
	timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -o cal -- "$REPO/tools/pmc_calibrate" $MIB 2 > "$OUT/pass$i.log" 2>&1
	echo "pass $i ($set): rc=$?"
done
python "$REPO/tools/pmc_calibrate_summary.py" "$OUT" "$REPO/gpurun_out/pmc_calibration.md" "$REPO/gpurun_out/pmc_calibration.json"
