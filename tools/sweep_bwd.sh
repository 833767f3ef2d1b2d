# times every library variant under recsys-examples_amd/lib/var on the C2 dedup
R=$GRAFT_REPO_ROOT
for f in $R/recsys-examples_amd/lib/var/*.so; do MI355_LIB=$f timeout 120 python $R/tools/bench_uniq_c2.py 15 2>&1 | grep -E "segmented|rror" | tail -2; done
