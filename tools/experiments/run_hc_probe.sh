# k_solve_hc with parts removed (SGP_DEBUG_FLAGS 8: no colour phases at all; results are wrong, only the launch time is of interest)
R=$PWD; cd /tmp && export TMPDIR=/tmp
for f in 0 8; do
  export SGP_DEBUG_FLAGS=$f
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_hc -o kt -- python $R/bench.py --steps 20 --warmup 30 --no-cpu-baseline --no-readback-leg > $R/gpurun_out/hc_prof.log 2>&1
  db=$(find $R/gpurun_out/prof_hc -name "*.db" | head -1); echo "flags $f"; python $R/tools/rocpd_summary.py $db | grep "k_solve_hc"; rm -rf $R/gpurun_out/prof_hc
done
