# kernel trace of the C2 step: per-kernel stats + the launch timeline of the last steps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_step -o t -- python $R/bench.py --steps 20 --warmup 5 --no-hstu --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_step.log 2>&1
db=$(find $R/gpurun_out/prof_step -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $db 60 > $R/gpurun_out/step_timeline.txt
python $R/tools/rocpd_stats.py $db > $R/gpurun_out/step_stats.txt
rm -rf $R/gpurun_out/prof_step
