#!/bin/bash
# A/B on the GPU box: bench, apply a sed edit to the stage files (sgp_k_*.hip), rebuild, bench again (the tree on the box is a scratch copy).
b() { timeout 200 python bench.py --cpu-steps 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value'],1), k['solve_velocity'], k['narrowphase'])"; }
echo "baseline:"; b
for edit in "$@"; do
  sed -i "$edit" substrata_amd/csrc/sgp_k_*.hip
  python -c "from substrata_amd import build; build.build(force=True)" 2>&1 | grep -E "error" | head -3
  echo "after [$edit]:"; b
done
