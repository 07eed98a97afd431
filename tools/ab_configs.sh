#!/bin/bash
# Same-box A/B of builds of libsgp.so over the bench workloads: config 3 and 5 through tools/ab_bench.sh, config 4 (1 M bodies) once per build.
# Usage (GPU box, repository root): bash tools/ab_configs.sh <rounds> <lib or "tree"> [<lib> ...]
ROUNDS=$1; shift
echo CONFIG3; bash tools/ab_bench.sh $ROUNDS "$@"
echo CONFIG5; bash tools/ab_bench.sh 1 "$@" -- --workload config5
echo CONFIG4
for lib in "$@"; do
	if [ "$lib" = tree ]; then unset SGP_LIB_PATH; else export SGP_LIB_PATH=$PWD/$lib; fi
	timeout 600 python bench.py --workload config4 --steps 60 --warmup 20 --cpu-steps 0 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); k = j['kernel_ms_per_step']
print('$(basename $lib .so)'.ljust(22), round(j['value'], 2), 'steps/s |', ' '.join(f'{n}={v*1000:.0f}' for n, v in k.items() if v >= 0.1))"
done
