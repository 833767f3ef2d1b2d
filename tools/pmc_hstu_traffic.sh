#!/bin/bash
# HBM traffic of the attention kernels from the TCC counters (separate counters-only passes, as MI355X_MICROARCH.md prescribes):
#   bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB; FETCH_SIZE counts 64 B per 128-B request on gfx950).
# Usage on the GPU box: bash tools/pmc_hstu_traffic.sh [batch] [seqlen] [out]
B=${1:-8}; L=${2:-4096}; OUT=${3:-gpurun_out/pmc_hstu_traffic.txt}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
: > $ROOT/$OUT
i=0
for SET in "FETCH_SIZE WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/pmct$i
  timeout 400 rocprofv3 --pmc $SET -d /tmp/pmct$i -o p -- python $ROOT/tools/bench_hstu.py --batch $B --seqlen $L --reps 1 > /tmp/pmct$i.log 2>&1
  DB=$(ls /tmp/pmct$i/*/*.db /tmp/pmct$i/*.db 2>/dev/null | head -1)
  echo "## pass $i (batch $B x L $L): $SET" >> $ROOT/$OUT
  python $ROOT/tools/pmc_dump.py $DB hstu >> $ROOT/$OUT 2>&1
done
cat $ROOT/$OUT | cut -c1-150
