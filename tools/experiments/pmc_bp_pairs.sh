# SQ counters for the kernels of one step (k_bp_pairs, k_narrowphase, k_setup ...): where the wave cycles go
R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES"; do
  rm -rf $R/gpurun_out/pmc_sq
  SGP_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o pmc -- python $R/bench.py --steps 6 --warmup 6 --cpu-steps 0 --no-readback-leg --profile-steps 1 > $R/gpurun_out/pmc_sq.log 2>&1
  f=$(find $R/gpurun_out/pmc_sq -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if any(t in k for t in ("k_bp_pairs", "k_narrowphase", "k_setup", "k_warm_bodies", "k_solve_hc<1>", "k_solve_colour<1>")):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v[len(v)//2:]) / max(1, len(v[len(v)//2:]))) for c, v in d.items()})
PY
done
rm -rf $R/gpurun_out/pmc_sq
