# A/B of attention kernel variants under recsys-examples_amd/lib/var (MI355_LIB) against the default build
R=$GRAFT_REPO_ROOT
for L in 512 4096; do
  echo -n "default L=$L: "; python $R/tools/bench_hstu.py --seqlen $L --reps 10 2>&1 | grep fwd
  for f in $R/recsys-examples_amd/lib/var/*.so; do echo -n "$(basename $f) L=$L: "; MI355_LIB=$f python $R/tools/bench_hstu.py --seqlen $L --reps 10 2>&1 | grep fwd; done
done
