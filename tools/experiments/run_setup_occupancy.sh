# k_setup with a minimum number of waves per SIMD forced through __launch_bounds__ (fewer registers, possibly spills): rebuilds per variant
for w in 0 3 4; do
  if [ $w = 0 ]; then lb="__launch_bounds__(TPB) k_setup(DV d)"; else lb="__launch_bounds__(TPB, $w) k_setup(DV d)"; fi
  sed -i "s/__launch_bounds__(TPB[, 0-9]*) k_setup(DV d)/$lb/" substrata_amd/csrc/sgp_k_*.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "min waves per SIMD: $w"
  bash tools/experiments/run_timeline.sh | grep -E "k_setup "
done
sed -i "s/__launch_bounds__(TPB[, 0-9]*) k_setup(DV d)/__launch_bounds__(TPB) k_setup(DV d)/" substrata_amd/csrc/sgp_k_*.hip
