cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/d
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_pipe -o t -- python $R/tools/pipeline_step.py --steps 40 > $R/gpurun_out/d/prof.log 2>&1
db=$(find $R/gpurun_out/prof_pipe -name '*.db' | head -1)
python - $db > $R/gpurun_out/d/timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall() if "queue_id" in cols else c.execute("select * from kernels order by start limit 3").fetchall()
# the pipelined section: between the two serial sections; print a window in the middle of the run
n = len(rows)
lo = n // 2 - 30
t0 = rows[lo][1]
for r in rows[lo:lo + 60]:
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:7.1f}  q{r[3]} s{r[4]}  {r[0][:60]}")
PY
rm -rf $R/gpurun_out/prof_pipe
cat $R/gpurun_out/d/timeline.txt | head -70
