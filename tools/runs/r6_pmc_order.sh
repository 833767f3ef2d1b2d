#!/bin/bash
# round 6, review item 2: HBM-side traffic of bwd_kernel with the unique rows in hash order (path (c), MI355_FUSED=1) and in
# first-occurrence = bag order (the per-op index path, MI355_FUSED=0): separate rocprofv3 --pmc passes, counters only
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06_pmc_unique_order.txt
cd /tmp && export TMPDIR=/tmp
: > $O
for F in 1 0; do
  i=0
  for SET in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1)); rm -rf /tmp/po$F$i
    MI355_FUSED=$F rocprofv3 --pmc $SET -d /tmp/po$F$i -o p -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-hstu --no-extra > /tmp/po$F$i.log 2>&1
    DB=$(ls /tmp/po$F$i/*/*.db /tmp/po$F$i/*.db 2>/dev/null | head -1)
    echo "## MI355_FUSED=$F pass $i: $SET" >> $O
    python $R/tools/pmc_dump.py $DB mi355 2>&1 | grep "bwd_kernel" >> $O
  done
done
cat $O
