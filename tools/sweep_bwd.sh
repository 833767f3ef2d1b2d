# C2 backward under different hot-row thresholds / chunk sizes (env knobs of csrc/hot.h)
R=$GRAFT_REPO_ROOT
for hc in "8 256" "8 128" "8 512" "8 1024" "16 256" "32 512" "4 256" "64 1024" "1000000 256"; do set -- $hc
  echo -n "HOT=$1 CHUNK=$2 "; MI355_HOT=$1 MI355_CHUNK=$2 python $R/tools/bench_bwd_c2.py 15 zipf 2>&1 | grep bwd_kernel
done
