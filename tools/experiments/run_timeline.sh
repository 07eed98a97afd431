# step timeline of the timed leg of config 3 (see profiles/*_step_timeline_config3.md)
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o kt -- python $R/bench.py --steps 50 --warmup 20 --no-cpu-baseline --no-readback-leg > $R/gpurun_out/tl_bench.log 2>&1
db=$(find $R/gpurun_out/prof_tl -name "*.db" | head -1)
python $R/tools/step_timeline.py $db --steps 40 --skip-last 60 > $R/gpurun_out/step_timeline.md
rm -rf $R/gpurun_out/prof_tl
cat $R/gpurun_out/step_timeline.md
