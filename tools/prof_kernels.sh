#!/bin/bash
# Per-kernel evidence for every hot kernel: rocprofv3 --kernel-trace --stats, three SQ counter passes and the two HBM-traffic passes
# (separate --pmc runs, --kernel-trace only: no other trace domain), then tools/pmc_summary.py -> gpurun_out/TAG_kernels.json.
# usage (on the GPU box): bash tools/prof_kernels.sh TAG [program args...]   (default program: python3 tools/run_all_kernels.py)
# The program goes directly after `--` (never through env / bash -c: the profiler's preloaded library has initialised the GPU).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}; shift
cmd=${*:-python3 tools/run_all_kernels.py}
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pk_${tag}_stats -- $cmd > gpurun_out/pk_${tag}_stats.log 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pk_${tag}_pmc$i -- $cmd > gpurun_out/pk_${tag}_pmc$i.log 2>&1
done
python3 tools/pmc_summary.py $tag "$cmd"
