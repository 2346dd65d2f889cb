#!/bin/bash
# Everything the round's measurement section cites, in one call on the GPU box: bash tools/prof_round.sh TAG  -> gpurun_out/TAG_*
cd $GRAFT_REPO_ROOT
tag=${1:-r06_z}
mkdir -p gpurun_out
bash tools/prof_headline.sh $tag > gpurun_out/${tag}_prof_headline.log 2>&1
# (the traffic passes run first and land in profiles/: bench.py labels its roofline.traffic with the newest profiles/*_pmc_traffic.json and says whether
#  it was taken at THIS kernel source)
python bench.py > gpurun_out/${tag}_bench_full.json 2> gpurun_out/${tag}_bench_full.err
bash tools/prof_dp4.sh $tag > /dev/null 2>&1
bash tools/prof_train_step.sh $tag > gpurun_out/${tag}_prof_train_step.log 2>&1
python tools/step_kernel_sequence.py gpurun_out/prof_${tag}_ts --all > gpurun_out/${tag}_train_step_sequence.txt 2>&1
bash tools/prof_kernels.sh $tag > gpurun_out/${tag}_prof_kernels.log 2>&1
# the shipped factor layout (1369 columns): per-kernel stats + launch sequence of the step, the key-split attention fuse alone (stats + counters)
bash tools/prof_train_step.sh ${tag}_shipped 64 40 36 --shipped > gpurun_out/${tag}_prof_train_step_shipped.log 2>&1
python tools/step_kernel_sequence.py gpurun_out/prof_${tag}_shipped_ts --all > gpurun_out/${tag}_shipped_train_step_sequence.txt 2>&1
bash tools/prof_any.sh ${tag}_attn_wide tools/time_attn_wide.py 64 > /dev/null 2>&1
bash tools/prof_attn_pmc.sh ${tag} 64 > /dev/null 2>&1
python tools/time_headline.py 256 80 > gpurun_out/${tag}_dp_L80.log 2>&1
python tools/time_train_step.py --f32 > gpurun_out/${tag}_train_step_f32.log 2>&1
tail -1 gpurun_out/${tag}_train_step_sequence.txt; tail -1 gpurun_out/${tag}_shipped_train_step_sequence.txt; tail -1 gpurun_out/${tag}_train_step_f32.log
ls gpurun_out | grep "^${tag}" | head -40
