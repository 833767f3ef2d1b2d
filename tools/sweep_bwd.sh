R=$GRAFT_REPO_ROOT
for cfg in "4 1024" "3 1024" "2 1024" "4 2048" "3 2048" "2 2048" "2 4096" "4 512"; do set -- $cfg; echo -n "HOT=$1 WB=$2 "; MI355_HOT=$1 MI355_WAVE_BLOCKS=$2 python $R/tools/bench_bwd_c2.py 15 zipf 2>&1 | grep bwd_kernel; done
