#!/bin/bash
# per-kernel times of the chained training step: bash tools/prof_train_step.sh TAG -> gpurun_out/TAG_train_step_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02}
mkdir -p gpurun_out
rm -rf gpurun_out/prof_${tag}_ts; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_ts -- python tools/time_train_step.py ${@:2} > gpurun_out/${tag}_train_step.log 2>&1
f=$(find gpurun_out/prof_${tag}_ts -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_train_step_kernel_stats.csv
grep "ms/step" gpurun_out/${tag}_train_step.log
