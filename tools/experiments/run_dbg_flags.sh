# kernel timing with parts switched off by SGP_DEBUG_FLAGS (results are garbage, timing only): KERNEL = name prefix, FLAGS = list
cd /tmp && export TMPDIR=/tmp
for F in ${FLAGS:-0}; do
  export SGP_DEBUG_FLAGS=$F
  rm -rf /tmp/prof
  timeout 900 rocprofv3 --kernel-trace -d /tmp/prof -o trace -- python $GRAFT_REPO_ROOT/tools/experiments/blocks_probe.py > /tmp/b.log 2>&1
  echo "== flags $F"; tail -1 /tmp/b.log
  for K in ${KERNELS:-k_narrowphase}; do python $GRAFT_REPO_ROOT/tools/experiments/kernel_windows.py /tmp/prof/trace_results.db "$K" 10 | tail -2; done
done
