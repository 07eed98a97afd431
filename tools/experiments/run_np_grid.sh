# k_narrowphase: cap on the number of workgroups (they loop over the pairs; one global atomic per workgroup and iteration)
for b in 4096 2048 1024; do
  sed -i "s/hipLaunchKernelGGL(k_narrowphase, dim3(std::min(stride_grid(est), [0-9]*u))/hipLaunchKernelGGL(k_narrowphase, dim3(std::min(stride_grid(est), ${b}u))/" substrata_amd/csrc/sgp_k_*.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "cap $b"; bash tools/experiments/run_timeline.sh | grep -E "k_narrowphase "
done
