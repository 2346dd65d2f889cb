#!/bin/bash
# rocprofv3 evidence for the headline bench command (run on the GPU box): kernel-trace stats + the two HBM-traffic PMC passes.
# usage: bash tools/prof_headline.sh TAG   -> gpurun_out/TAG_kernel_stats.csv, TAG_pmc_traffic.json, TAG_bench_line.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02}
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-secondary > gpurun_out/pmc_${tag}_$c.log 2>&1
done
# the secondary DP lines that carry a roofline of their own: configs[3] (L = 80, overlay placement: value charts copied to the workspace) and the
# fp32-stored headline -- the same two passes each, program directly behind `--`
for w in "256 80 bf16" "256 40 f32"; do
  key=$(echo $w | awk '{print "B"$1"_L"$2"_"$3}')
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_${key}_$c -- python tools/dp_workload.py $w 20 > gpurun_out/pmc_${tag}_${key}_$c.log 2>&1
  done
done
python - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
import bench
res = {"kernel_source_id": bench.kernel_source_id(), "command": "python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-secondary",
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; KB per launch; gfx950: FETCH_SIZE x2 before comparing with a byte count"}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/pmc_${tag}_{c}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c and 'dmv1o_kernel' in r['Kernel_Name']:
                acc['dmv1o_B256_L40_bf16'].append(float(r['Counter_Value']))
        for k, v in acc.items():
            res.setdefault(k, {})[c] = {'n': len(v), 'avg_KB': sum(v) / len(v), 'min_KB': min(v), 'max_KB': max(v)}
for key in ('B256_L80_bf16', 'B256_L40_f32'):
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        for f in glob.glob(f'gpurun_out/pmc_${tag}_{key}_{c}/**/*counter_collection.csv', recursive=True):
            v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == c and 'dmv1o_kernel' in r['Kernel_Name']]
            if v:
                res.setdefault('dmv1o_' + key, {})[c] = {'n': len(v), 'avg_KB': sum(v) / len(v), 'min_KB': min(v), 'max_KB': max(v)}
print(json.dumps(res, indent=1))
json.dump(res, open('gpurun_out/${tag}_pmc_traffic.json', 'w'), indent=1)
PY
# the traffic figure into profiles/ BEFORE the stats run, so that the bench line it prints cites the passes taken at this kernel source
mkdir -p profiles; cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-secondary > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv; head -4 "$f" | cut -c1-220
