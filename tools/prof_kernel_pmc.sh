#!/bin/bash
# SQ / TCC counters of the kernels whose name contains PATTERN, in any tool script:
#   bash tools/prof_kernel_pmc.sh TAG PATTERN script.py [args]  -> gpurun_out/TAG_pmc.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; pat=$2; shift; shift
mkdir -p gpurun_out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmck_${tag}_$i -- python "$@" > gpurun_out/pmck_${tag}_$i.log 2>&1
done
python - <<PY > gpurun_out/${tag}_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob('gpurun_out/pmck_${tag}_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if '$pat' in n:
                acc[n.split('(')[0][:100]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print('   %-28s avg %.4g  (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
rm -rf gpurun_out/pmck_${tag}_*
cat gpurun_out/${tag}_pmc.txt
