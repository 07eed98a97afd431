# k_colour_count with fewer, longer-looping workgroups (its per-colour global atomics serialise)
for b in 512 256 128 64; do
  sed -i "s/hipLaunchKernelGGL(k_colour_count, dim3(std::min(stride_grid(est), [0-9]*u))/hipLaunchKernelGGL(k_colour_count, dim3(std::min(stride_grid(est), ${b}u))/" substrata_amd/csrc/sgp_k_*.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "blocks $b"; bash tools/experiments/run_timeline.sh | grep -E "k_colour_count"
done
sed -i "s/hipLaunchKernelGGL(k_colour_count, dim3(std::min(stride_grid(est), [0-9]*u))/hipLaunchKernelGGL(k_colour_count, dim3(std::min(stride_grid(est), 512u))/" substrata_amd/csrc/sgp_k_*.hip
