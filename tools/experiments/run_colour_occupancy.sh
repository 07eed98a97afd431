# colour launches with a minimum number of waves per SIMD forced through __launch_bounds__
for w in 0 4; do
  if [ $w = 0 ]; then lb="__launch_bounds__(MODE != 0 ? SOLVE_VEL_TPB : SOLVE_TPB) k_solve_colour"; else lb="__launch_bounds__(MODE != 0 ? SOLVE_VEL_TPB : SOLVE_TPB, $w) k_solve_colour"; fi
  sed -i "s/__launch_bounds__(MODE != 0 ? SOLVE_VEL_TPB : SOLVE_TPB[, 0-9]*) k_solve_colour/$lb/" substrata_amd/csrc/sgp_k_*.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "min waves per SIMD: $w"
  for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-readback-leg 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; print('  config3', round(j['value'],1), 'vel', k['solve_velocity'], 'pos', k['solve_position'])"; done
done
sed -i "s/__launch_bounds__(MODE != 0 ? SOLVE_VEL_TPB : SOLVE_TPB[, 0-9]*) k_solve_colour/__launch_bounds__(MODE != 0 ? SOLVE_VEL_TPB : SOLVE_TPB) k_solve_colour/" substrata_amd/csrc/sgp_k_*.hip
