cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/bench_extended.py --only c2_cold_insert 2>/dev/null | cut -c1-200
rocprofv3 --kernel-trace -d /tmp/pc -o t -- python $R/tools/bench_extended.py --only c2_cold_insert > /tmp/pc.log 2>&1
db=$(find /tmp/pc -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $db | head -14 | cut -c1-150
