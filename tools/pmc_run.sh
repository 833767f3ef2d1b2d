#!/bin/bash
# PMC passes over the C2 bench (separate runs, counters only: no --sys-trace / hip-trace with --pmc).
# Usage (on the GPU box): bash tools/pmc_run.sh <outfile>
set -u
OUT=${1:-gpurun_out/pmc_c2.txt}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
: > $ROOT/$OUT
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  rocprofv3 --pmc $SET -d /tmp/pmc$i -o p -- python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-hstu > /tmp/pmc$i.log 2>&1
  DB=$(ls /tmp/pmc$i/*/*.db /tmp/pmc$i/*.db 2>/dev/null | head -1)
  echo "## pass $i: $SET" >> $ROOT/$OUT
  python $ROOT/tools/pmc_dump.py $DB mi355 >> $ROOT/$OUT 2>&1
done
