#!/bin/bash
# PMC passes (MFMA-busy etc.) for the matrix-core kernels: alignment (align_max / align_mfma), attention-fuse, trilinear.
# usage: bash tools/prof_mfma.sh TAG "python tools/time_align.py"   -> gpurun_out/pmcm_TAG_summary.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}; shift
cmd=${*:-python tools/time_align.py}
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcm_${tag}_$i -- $cmd > gpurun_out/pmcm_${tag}_$i.log 2>&1
done
python - <<PY
import csv, glob, collections, re
out = open('gpurun_out/pmcm_${tag}_summary.txt', 'w')
for d in sorted(glob.glob('gpurun_out/pmcm_${tag}_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            m=re.search(r'(ground_bwd_dense_kernel<[01]|ground_transpose_kernel|align_full_kernel|align_argmax_kernel|align_max_kernel|align_mfma_kernel|attn_fuse_mfma_kernel|attn_fuse_bwd_words_kernel|attn_fuse_bwd_regions_kernel|attn_fuse\w*kernel|tri_kernel|tri_dw_kernel|align_bwd\w*kernel)', n)
            if m:
                key=m.group(1) + ('' if 'align_mfma' not in n else ('<TILE>' if 'true, false' in n or 'Lb1ELb0' in n else ''))
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key in sorted(acc):
            for k,v in sorted(acc[key].items()):
                line = '%s %s n=%d avg=%.0f' % (key, k, len(v), sum(v)/len(v))
                print(line); out.write(line + '\n')
PY
