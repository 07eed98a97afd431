# config 5: hull narrow phase / vehicle cast kernels with a minimum number of waves per SIMD forced through __launch_bounds__
for w in 0 3 4; do
  for k in k_narrowphase_hull k_narrowphase_hull_manifold k_vehicle_cast; do
    if [ $w = 0 ]; then lb="__launch_bounds__(64) $k(DV d)"; else lb="__launch_bounds__(64, $w) $k(DV d)"; fi
    sed -i "s/__launch_bounds__(64[, 0-9]*) $k(DV d)/$lb/" substrata_amd/csrc/sgp_k_*.hip
  done
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "min waves per SIMD: $w"
  python bench.py --workload config5 --steps 60 --warmup 60 --no-cpu-baseline --no-readback-leg 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); k=j['kernel_ms_per_step']; print('  config5', round(j['value'],1), 'narrowphase', k['narrowphase'], 'vehicle', k.get('vehicle'))"
done
for k in k_narrowphase_hull k_narrowphase_hull_manifold k_vehicle_cast; do sed -i "s/__launch_bounds__(64[, 0-9]*) $k(DV d)/__launch_bounds__(64) $k(DV d)/" substrata_amd/csrc/sgp_k_*.hip; done
