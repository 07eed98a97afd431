#!/bin/bash
# Wave / memory-pipe counters per kernel (why is a kernel slow: waiting, issuing memory instructions, stalled behind the L2?) -- counters-only rocprofv3 passes
# (--kernel-trace, eager launches), one small counter set per pass.  Usage (GPU box, repository root): bash tools/collect_sq.sh [bench args]
# -> gpurun_out/sq/sq_summary.md (kernels with >= 0.5 % of the summed SQ_BUSY_CYCLES).
REPO=$PWD
OUT=$REPO/gpurun_out/sq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
	i=$((i + 1))
	SGP_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pass$i" -o sq -- \
		python "$REPO/bench.py" --steps 12 --warmup 8 --cpu-steps 0 --no-readback-leg --profile-steps 1 "$@" > "$OUT/pass$i.log" 2>&1
	echo "pass $i ($set): rc=$?"
done
python "$REPO/tools/sq_summary.py" "$OUT" "$OUT/sq_summary.md"
find "$OUT" -name "*.csv" -delete
