#!/bin/bash
# per-kernel times of the grounding loss (config-2 then shipped layout): bash tools/prof_ground.sh TAG
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02_ground}
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python tools/run_ground.py > gpurun_out/${tag}.log 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
cat gpurun_out/${tag}.log | tail -3
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f}us")
PY
