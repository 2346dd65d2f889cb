#!/bin/bash
# per-kernel counters for the DMV kernel (separate passes per guide)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-align > gpurun_out/pmc_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'dmv1o_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            print(k, 'n=%d'%len(v), 'avg=%.1f'%(sum(v)/len(v)))
PY
