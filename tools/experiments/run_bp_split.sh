# k_bp_pairs: threads that share one body's candidate rows
for sp in 4 2 8; do
  sed -i "s/#define BP_SPLIT [0-9]*/#define BP_SPLIT $sp/" substrata_amd/csrc/sgp_k_*.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "split $sp"; bash tools/experiments/run_timeline.sh | grep -E "k_bp_pairs"
done
sed -i "s/#define BP_SPLIT [0-9]*/#define BP_SPLIT 4/" substrata_amd/csrc/sgp_k_*.hip
