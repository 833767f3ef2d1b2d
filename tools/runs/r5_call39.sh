#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in 0 1; do MI355_HSTU_FUNC_DENSE=$v timeout 300 python tools/bench_hstu_func.py 2>&1 | grep "func \|Error"; done
for v in 0 1; do MI355_HSTU_FUNC_DENSE=$v timeout 300 python tools/bench_hstu_func.py --batch 32 --seqlen 512 2>&1 | grep "func \|Error"; done
