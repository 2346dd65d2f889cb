cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# kernel-trace stats for the default bench command
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1g -- python bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-align > gpurun_out/bench_r1g.log 2>&1
tail -1 gpurun_out/bench_r1g.log | cut -c1-300
# HBM traffic counters, one pass each (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcg_$c -- python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-align > gpurun_out/pmcg_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'gpurun_out/pmcg_{c}/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c:
                k = 'dmv1o_kernel' if 'dmv1o_kernel' in r['Kernel_Name'] else 'align_mfma_kernel' if 'align_mfma' in r['Kernel_Name'] else None
                if k: acc[k].append(float(r['Counter_Value']))
        for k, v in acc.items():
            res.setdefault(k, {})[c] = {'n': len(v), 'avg_KB': sum(v) / len(v), 'min_KB': min(v), 'max_KB': max(v)}
print(json.dumps(res, indent=1))
json.dump(res, open('gpurun_out/pmc_traffic_r1g.json', 'w'), indent=1)
PY
f=$(find gpurun_out/prof_r1g -name "*kernel_stats.csv" | head -1); head -5 "$f" | cut -c1-200
