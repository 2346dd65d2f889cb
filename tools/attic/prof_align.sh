cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_align.py 2>&1 | tail -6
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmca_$tag -- python tools/bench_align.py > gpurun_out/pmca_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmca_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'align_mfma' in r['Kernel_Name']:
                key='TILE' if 'ELb1EEE' in r['Kernel_Name'] else 'NOTILE'
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key in acc:
            for k,v in acc[key].items():
                print(key, k, 'n=%d'%len(v), 'avg=%.0f'%(sum(v)/len(v)))
PY
