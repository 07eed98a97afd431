# k_bp_pairs with different LDS capacities (records of the halo, staged pairs): rebuilds libsgp.so per variant on the GPU box
for caps in "640 1024" "896 1024" "1024 1024" "1536 2048"; do
  set -- $caps
  sed -i "s/#define BP_LDS_CAP [0-9]*/#define BP_LDS_CAP $1/; s/#define BP_PAIR_CAP [0-9]*/#define BP_PAIR_CAP $2/" substrata_amd/csrc/sgp_kernels.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "caps $caps"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-readback-leg 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('  config3', round(j['value'],1), j['kernel_ms_per_step']['bp_pairs'])"
  python bench.py --workload config5 --steps 60 --warmup 60 --no-cpu-baseline --no-readback-leg 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('  config5', round(j['value'],1), j['kernel_ms_per_step']['bp_pairs'])"
  python tools/small_bench.py 2>&1 | grep -A1 config2 | sed 's/.*bp_pairs.: (\([0-9.]*\).*/  config2 bp_pairs \1/' | head -2
done
