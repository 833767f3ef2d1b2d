#!/bin/bash
# round-4 evidence, second half (after the 64-rows-per-wave forward): attention shapes, phase stamps, SQ counters, kernel trace, bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; L=$R/recsys-examples_amd/lib
cd $R
export MASTER_ADDR=127.0.0.1
timeout 300 python tools/hstu_shapes.py --seeds 4 2>&1 | grep -v amdgpu.ids > $O/r04_hstu_shapes.txt
( echo "== forward, 64 rows per wave (hstu_fwd_q2_kernel), 8 x 4096"; MI355_HSTU_PAIR=0 MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --q2 --batch 8 --seqlen 4096
  echo "== forward, 32 rows per wave (hstu_fwd_pc_kernel), 8 x 4096"; MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 8 --seqlen 4096
  echo "== forward, 32 rows per wave (hstu_fwd_pc_kernel), C3"; MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 32 --seqlen 512
  echo "== backward dK pass (hstu_bwd_kv_pc_kernel), 8 x 4096"; MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --bwdpc --batch 8 --seqlen 4096
  echo "== backward dK pass (hstu_bwd_kv_pc_kernel), C3"; MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --bwdpc --batch 32 --seqlen 512 ) 2>&1 | grep -v amdgpu.ids > $O/r04_hstu_phase_stamps.txt
timeout 400 bash tools/pmc_hstu.sh 4096 gpurun_out/r04_pmc_hstu_l4096.txt
timeout 300 bash tools/pmc_hstu.sh 512 gpurun_out/r04_pmc_hstu.txt
( cd /tmp && export TMPDIR=/tmp
  for cfg in "c3 32 512" "l4096 8 4096"; do
    set -- $cfg; rm -rf /tmp/prof_$1
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o t -- python $R/tools/bench_hstu.py --batch $2 --seqlen $3 --reps 10 > /dev/null 2>&1
    db=$(find /tmp/prof_$1 -name '*.db' | head -1)
    echo "== attention kernels, batch $2 x L $3 (rocprofv3 --kernel-trace)"; python $R/tools/rocpd_stats.py $db | grep -i "kernel \|hstu\|total" | cut -c1-160
  done ) > $O/r04_hstu_kernel_trace_stats.txt 2>&1
timeout 900 python bench.py > $O/r04_c2_bench.json 2> $O/r04_c2_bench.err
tail -c 600 $O/r04_c2_bench.json
ls -la $O/r04_* | awk '{print $5, $9}'
