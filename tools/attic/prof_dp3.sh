#!/bin/bash
# PMC passes for the DP kernels (inside-only and fused) under tools/time_fw.py; run on the GPU box.
# usage: bash tools/prof_dp3.sh TAG   -> gpurun_out/pmc3_TAG_*.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc3_${tag}_$i -- python tools/time_fw.py > gpurun_out/pmc3_${tag}_$i.log 2>&1
done
python - <<PY
import csv, glob, collections
out = open('gpurun_out/pmc3_${tag}_summary.txt', 'w')
for d in sorted(glob.glob('gpurun_out/pmc3_${tag}_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if 'dmv1o_kernel' in n:
                key='fused' if ', true,' in n else 'inside'
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key in sorted(acc):
            for k,v in sorted(acc[key].items()):
                line = '%s %s avg=%.0f per-wave=%.0f' % (key, k, sum(v)/len(v), sum(v)/len(v)/2048)
                print(line); out.write(line + '\n')
PY
