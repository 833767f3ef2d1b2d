#!/bin/bash
# round-4 evidence in one box call: tools/collect_profiles.sh r04 + attention shapes / phase stamps / kernel trace + gather ubench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; L=$R/recsys-examples_amd/lib
cd $R
export MASTER_ADDR=127.0.0.1
bash tools/collect_profiles.sh r04
timeout 400 bash tools/pmc_hstu.sh 4096 gpurun_out/r04_pmc_hstu_l4096.txt
timeout 300 python tools/hstu_shapes.py --seeds 4 > $O/r04_hstu_shapes.txt 2>&1
( MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 8 --seqlen 4096
  MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --pc --batch 32 --seqlen 512
  MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --bwdpc --batch 8 --seqlen 4096
  MI355_LIB=$L/librecsys_amd_tim.so timeout 120 python tools/hstu_phase_cycles.py --bwdpc --batch 32 --seqlen 512 ) > $O/r04_hstu_phase_stamps.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  for cfg in "c3 32 512" "l4096 8 4096"; do
    set -- $cfg; rm -rf /tmp/prof_$1
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o t -- python $R/tools/bench_hstu.py --batch $2 --seqlen $3 --reps 10 > /dev/null 2>&1
    db=$(find /tmp/prof_$1 -name '*.db' | head -1)
    echo "== attention kernels, batch $2 x L $3 (rocprofv3 --kernel-trace)"; python $R/tools/rocpd_stats.py $db | grep -i "kernel \|hstu\|total" | cut -c1-160
  done ) > $O/r04_hstu_kernel_trace_stats.txt 2>&1
timeout 600 python bench.py --force-sharded --no-hstu --no-cpu-baseline > $O/r04_sharded_w1_bench.json 2> /dev/null
timeout 300 python tools/bench_model_shapes.py > $O/r04_model_shapes.txt 2>&1
ls -la $O/r04_* | awk '{print $5, $9}'
