#!/bin/bash
# Sweep of the tail-kernel threshold (colours with at most this many constraints share the single-workgroup tail launch).
for t in 256 768 1536 3072 6144; do
  echo -n "threshold $t: "
  SGP_TAIL_THRESHOLD=$t timeout 200 python bench.py --cpu-steps 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value'],1), 'steps/s  vel', k['solve_velocity'], 'pos', k['solve_position'], 'warm', k['warm_start'], 'launches', d['roofline_solver']['launches_per_step'])"
done
