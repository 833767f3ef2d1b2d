#!/bin/bash
# SQ counters of the HSTU kernels (one counters-only pass per set).  Usage on the GPU box: bash tools/pmc_hstu.sh [seqlen] [out]
L=${1:-4096}
OUT=${2:-gpurun_out/pmc_hstu.txt}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
: > $ROOT/$OUT
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pmch$i
  rocprofv3 --pmc $SET -d /tmp/pmch$i -o p -- python $ROOT/tools/bench_hstu.py --seqlen $L --reps 3 > /tmp/pmch$i.log 2>&1
  DB=$(ls /tmp/pmch$i/*/*.db /tmp/pmch$i/*.db 2>/dev/null | head -1)
  echo "## pass $i (L=$L): $SET" >> $ROOT/$OUT
  python $ROOT/tools/pmc_dump.py $DB hstu >> $ROOT/$OUT 2>&1
done
