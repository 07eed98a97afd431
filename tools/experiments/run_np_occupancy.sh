# k_narrowphase with a minimum number of waves per SIMD forced through __launch_bounds__: rebuilds per variant
for w in 4 5 6 8; do
  if [ $w = 0 ]; then lb="__launch_bounds__(TPB) k_narrowphase(DV d)"; else lb="__launch_bounds__(TPB, $w) k_narrowphase(DV d)"; fi
  sed -i "s/__launch_bounds__(TPB[, 0-9]*) k_narrowphase(DV d)/$lb/" substrata_amd/csrc/sgp_k_*.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "min waves per SIMD: $w"
  bash tools/experiments/run_timeline.sh | grep -E "k_narrowphase "
done
sed -i "s/__launch_bounds__(TPB[, 0-9]*) k_narrowphase(DV d)/__launch_bounds__(TPB) k_narrowphase(DV d)/" substrata_amd/csrc/sgp_k_*.hip
