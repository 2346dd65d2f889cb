#!/bin/bash
# per-kernel times of the grounding loss (config-2 widths, bf16) with a variant library:
#   bash tools/prof_ground_variant.sh TAG tools/variants/lib_X.so -> gpurun_out/TAG_ground_kernels.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}; export VLGAE_AMD_LIB=$2
mkdir -p gpurun_out
rm -rf gpurun_out/prof_${tag}_g
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_g -- python tools/attic/time_ground_bwd.py > gpurun_out/${tag}_ground.log 2>&1
f=$(find gpurun_out/prof_${tag}_g -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/${tag}_ground_kernels.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:120]:120s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f}us")
PY
rm -rf gpurun_out/prof_${tag}_g
tail -2 gpurun_out/${tag}_ground.log; cat gpurun_out/${tag}_ground_kernels.txt
