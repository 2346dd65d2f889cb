#!/bin/bash
# Measured ceilings of align_argmax_kernel by removal: bash tools/time_argmax_ablation.sh [shipped]
#   variants: the full kernel | without S^T (maxima over regions) | without S (maxima over queries) | without the first-equal searches;
#   each with ragged masks and with all-true masks (the mask extracts run on masked tiles only).  Average kernel time from rocprofv3.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out tools/variants
# (the variants are built in the build container -- hipcc cross-compiles -- and travel with the snapshot: bash tools/time_argmax_ablation.sh build)
build() { [ -f tools/variants/lib_$1.so ] || bash tools/build_variant_ground.sh "$@" > /dev/null; }
build am_full; build am_nost -DVLG_ABL_AM_NOST; build am_nos -DVLG_ABL_AM_NOS; build am_nosearch -DVLG_ABL_AM_NOSEARCH
build am_mfma_only -DVLG_ABL_AM_NOSEARCH -DVLG_ABL_AM_NOBARRIER
[ "$1" = build ] && exit 0
for v in am_full am_nost am_nos am_nosearch am_mfma_only; do
  for m in masked unmasked; do
    rm -rf gpurun_out/prof_abl
    VLGAE_AMD_LIB=$PWD/tools/variants/lib_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_abl -- python tools/argmax_fwd.py $m $1 > gpurun_out/abl_$v.log 2>&1
    f=$(find gpurun_out/prof_abl -name "*kernel_stats.csv" | head -1)
    t=$(python tools/kstats.py "$f" | grep align_argmax_kernel | head -1 | awk '{print $1}')
    echo "$v $m ${1:-config2}: align_argmax_kernel $t us"
  done
done
rm -rf gpurun_out/prof_abl
