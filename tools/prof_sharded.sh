cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in rows partial; do
  rocprofv3 --kernel-trace -d $R/gpurun_out/prof_sh_$m -o t -- python $R/bench.py --force-sharded --shard-mode $m --steps 10 --warmup 5 --no-hstu --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/sh_$m.log 2>&1
  db=$(find $R/gpurun_out/prof_sh_$m -name '*.db' | head -1)
  python $R/tools/rocpd_timeline.py $db 150 > $R/gpurun_out/sh_${m}_timeline.txt
  python $R/tools/rocpd_stats.py $db > $R/gpurun_out/sh_${m}_stats.txt
  rm -rf $R/gpurun_out/prof_sh_$m
  tail -1 $R/gpurun_out/sh_$m.log
done
