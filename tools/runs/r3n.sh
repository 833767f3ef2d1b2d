#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3n; mkdir -p $O; L=$R/recsys-examples_amd/lib
cd $R
( timeout 900 python -m pytest tests/test_fused_fwd_gpu.py tests/test_module_gpu.py tests/test_twin_gpu.py tests/test_plugin_surface_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python tools/bench_extended.py --only c2_fwd_only_eval > $O/ext_ev.json 2> $O/err.txt
MI355_LIB=$L/librecsys_amd_evform2.so timeout 300 python tools/bench_extended.py --only c2_fwd_only_eval > $O/ext_ev_form2.json 2>> $O/err.txt
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$O/ext_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), {k:(round(v['ms_per_step']*1e3,1) if 'ms_per_step' in v else v) for k,v in d.items() if isinstance(v,dict)})
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t -- python $R/tools/bench_extended.py --only c2_fwd_only_eval > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB | head -4 | cut -c1-150
