#!/bin/bash
# instruction-cache counters of the headline kernel: bash tools/prof_icache.sh TAG
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-ic}
mkdir -p gpurun_out
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_${tag} -- python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-secondary > gpurun_out/pmc_${tag}.log 2>&1
python - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_${tag}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dmv1o_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, 'n=%d avg=%.0f' % (len(v), sum(v)/len(v)))
PY
tail -3 gpurun_out/pmc_${tag}.log | cut -c1-200
