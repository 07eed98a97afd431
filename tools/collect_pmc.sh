#!/bin/bash
# HBM traffic per kernel launch from the PMC counters (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 passes, counters only (--kernel-trace, no other trace domains), eager launches (SGP_NO_GRAPH=1) so that every
# launch is attributed to its kernel.  Run on the GPU box from the repository root; writes gpurun_out/pmc/pmc_summary.md
# and gpurun_out/pmc/pmc_traffic.json (copy both to profiles/; bench.py reads profiles/pmc_traffic.json).  tools/pmc_summary.py applies the guide's unit (KiB) and gfx950 (x2 on FETCH_SIZE) corrections.
REPO=$PWD
OUT=$REPO/gpurun_out/pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
	SGP_NO_GRAPH=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o pmc -- \
		python "$REPO/bench.py" --steps 30 --warmup 20 --cpu-steps 0 --no-readback-leg "$@" > "$OUT/$c.log" 2>&1
	echo "$c pass: rc=$?"
done
f=$(find "$OUT/FETCH_SIZE" -name "*counter_collection.csv" | head -1)
w=$(find "$OUT/WRITE_SIZE" -name "*counter_collection.csv" | head -1)
python "$REPO/tools/pmc_summary.py" "$f" "$w" "$OUT/pmc_summary.md" "$OUT/pmc_traffic.json" ${PMC_BODIES:-100001} | head -60
# keep only the summary (the raw csv files are hundreds of MB)
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
