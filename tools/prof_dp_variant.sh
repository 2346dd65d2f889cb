#!/bin/bash
# SQ counters of the headline DP launch of a variant library: bash tools/prof_dp_variant.sh TAG LIB.so -> gpurun_out/TAG_dp_pmc.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-x}; export VLGAE_AMD_LIB=$2
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcv_${tag}_$i -- python tools/time_headline.py > gpurun_out/pmcv_${tag}_$i.log 2>&1
done
python - <<PY > gpurun_out/${tag}_dp_pmc.txt
import csv, glob, collections
print("library: $2")
for d in sorted(glob.glob('gpurun_out/pmcv_${tag}_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if 'dmv1o_kernel' in n:
                key='fused' if ', true,' in n else 'inside'
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key in sorted(acc):
            for k,v in sorted(acc[key].items()):
                print(key, k, 'avg=%.0f'%(sum(v)/len(v)), 'per-wave=%.0f'%(sum(v)/len(v)/2048))
PY
rm -rf gpurun_out/pmcv_${tag}_*
cat gpurun_out/${tag}_dp_pmc.txt
