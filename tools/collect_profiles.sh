#!/bin/bash
# Everything profiles/ holds for a round, in one GPU-box call (results land in gpurun_out/<tag>_*; copy them to profiles/):
#   <tag>_c2_bench.json               the default bench line (un-profiled)
#   <tag>_c2_kernel_trace_stats.txt   rocprofv3 --kernel-trace --stats summary of the same command (+ HSTU kernels)
#   <tag>_step_timeline.txt           launch timeline of the last steps (tools/rocpd_timeline.py)
#   <tag>_pmc_c2.txt / _pmc_traffic.json   PMC passes (counters only) and the HBM traffic derived from them
#   <tag>_pmc_hstu.txt                SQ counters of the attention kernels at C3 (L = 512)
#   <tag>_extended.json               tools/bench_extended.py (secondary configurations of SURVEY 8(d))
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( time timeout 600 python $R/bench.py ) > $O/${TAG}_c2_bench.json 2> $O/${TAG}_c2_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB > $O/${TAG}_c2_kernel_trace_stats.txt
python $R/tools/rocpd_timeline.py $DB 80 > $O/${TAG}_step_timeline.txt
rm -rf /tmp/kt
timeout 600 bash $R/tools/pmc_run.sh gpurun_out/${TAG}_pmc_c2.txt
python $R/tools/pmc_traffic.py $O/${TAG}_pmc_c2.txt $O/${TAG}_pmc_traffic.json
timeout 400 bash $R/tools/pmc_hstu.sh 512 gpurun_out/${TAG}_pmc_hstu.txt
timeout 600 python $R/tools/bench_extended.py --out $O/${TAG}_extended.json > /dev/null 2> $O/${TAG}_extended.err
timeout 300 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.log 2>&1
tail -1 $O/${TAG}_smoke.log
