#!/bin/bash
# usage: rerun_seeds.sh <steps> seed seed ...
steps=$1; shift
for s in "$@"; do timeout 300 python tools/fuzz_parity.py --seeds $s-$s --steps $steps 2>&1 | grep -E "MISMATCH" | cut -c1-400; done; echo done
