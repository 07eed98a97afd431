#!/bin/bash
# Same-box A/B of builds of libsgp.so (same ABI): alternating runs of bench.py, value + per-class kernel times.
# Usage (GPU box, repository root): bash tools/ab_bench.sh <rounds> <lib or "tree"> [<lib> ...] [-- bench args]      ("tree" = the in-tree library)
ROUNDS=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq 1 $ROUNDS); do
	for lib in "${LIBS[@]}"; do
		if [ "$lib" = tree ]; then unset SGP_LIB_PATH; else export SGP_LIB_PATH=$PWD/$lib; fi
		python bench.py --steps 300 --warmup 60 --cpu-steps 0 --no-readback-leg "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['kernel_ms_per_step']
print('$(basename $lib .so)'.ljust(22), round(j['value'], 1), 'steps/s |', ' '.join(f'{n}={v*1000:.0f}' for n, v in k.items() if v >= 0.012))"
	done
done
