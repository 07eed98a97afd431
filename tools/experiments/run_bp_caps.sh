# k_bp_pairs: capacities of the small LDS instance (records of the halo, staged pairs); rebuilds libsgp.so per variant on the GPU box
for caps in "384 512" "512 768" "640 1024"; do
  set -- $caps
  sed -i "s/#define BP_LDS_CAP_SMALL [0-9]*/#define BP_LDS_CAP_SMALL $1/; s/#define BP_PAIR_CAP_SMALL [0-9]*/#define BP_PAIR_CAP_SMALL $2/" substrata_amd/csrc/sgp_k_*.hip
  python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "small instance: $caps"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-readback-leg 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('  config3', round(j['value'],1), j['kernel_ms_per_step']['bp_pairs'])"
  python bench.py --workload config5 --steps 60 --warmup 60 --no-cpu-baseline --no-readback-leg 2>&1 | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('  config5', round(j['value'],1), j['kernel_ms_per_step']['bp_pairs'])"
done
sed -i "s/#define BP_LDS_CAP_SMALL [0-9]*/#define BP_LDS_CAP_SMALL 640/; s/#define BP_PAIR_CAP_SMALL [0-9]*/#define BP_PAIR_CAP_SMALL 1024/" substrata_amd/csrc/sgp_k_*.hip
