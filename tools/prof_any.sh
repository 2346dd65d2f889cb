#!/bin/bash
# per-kernel times of any tool script: bash tools/prof_any.sh TAG script.py [args] -> gpurun_out/TAG_kernel_stats.csv (+ TAG.log)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python "$@" > gpurun_out/${tag}.log 2>&1
f=$(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_${tag}
python tools/kstats.py gpurun_out/${tag}_kernel_stats.csv | head -40
