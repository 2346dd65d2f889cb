cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VLGAE_AMD_LIB=$PWD/tools/variants/lib_nt256.so
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmcd_$tag -- python tools/time_fw.py > gpurun_out/pmcd_$tag.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmcd_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if 'dmv1o_kernel' in n:
                key='fused' if ', true,' in n else 'inside'
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key in sorted(acc):
            for k,v in sorted(acc[key].items()):
                print(key, k, 'avg=%.0f'%(sum(v)/len(v)), 'per-wave=%.0f'%(sum(v)/len(v)/1024))
PY
