#!/bin/bash
# Same-box A/B of two builds of libsgp.so (same ABI): alternating runs of bench.py, value + per-class kernel times.
# Usage (GPU box, repository root): bash tools/ab_bench.sh <other libsgp.so> [rounds] [bench args]      (the in-tree library is "new")
OTHER=$1; ROUNDS=${2:-2}; shift 2
for r in $(seq 1 $ROUNDS); do
	for which in base new; do
		if [ $which = base ]; then export SGP_LIB_PATH=$PWD/$OTHER; else unset SGP_LIB_PATH; fi
		python bench.py --steps 300 --warmup 60 --cpu-steps 0 --no-readback-leg "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['kernel_ms_per_step']
print('$which', round(j['value'], 1), 'steps/s |', ' '.join(f'{n}={v*1000:.0f}' for n, v in k.items() if v >= 0.012))"
	done
done
