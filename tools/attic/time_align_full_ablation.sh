#!/bin/bash
# Measured ceilings of align_full_kernel by removal: bash tools/time_align_full_ablation.sh   (variants are built in the build container:
# bash tools/time_align_full_ablation.sh build)   full | no products / LDS writes (the store stream alone) | no global stores (the compute side alone)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out tools/variants
build() { [ -f tools/variants/lib_$1.so ] || bash tools/build_variant_ground.sh "$@" > /dev/null; }
build af_full; build af_nomfma -DVLG_ABL_AF_NOMFMA; build af_nostore -DVLG_ABL_AF_NOSTORE; build af_seq -DVLG_ABL_AF_NOMFMA -DVLG_ABL_AF_SEQ; build af_plain -DVLG_ABL_AF_NOMFMA -DVLG_ABL_AF_PLAINST
build af_f_ldsbar -DVLG_ABL_AF_LDSBAR; build af_f_ldsbar_plain -DVLG_ABL_AF_LDSBAR -DVLG_ABL_AF_PLAINST; build af_f_plain -DVLG_ABL_AF_PLAINST; build af_s_ldsbar -DVLG_ABL_AF_NOMFMA -DVLG_ABL_AF_LDSBAR
build af_rt3 -DVLG_AF_RT=3 -DVLG_AF_WPE=4; build af_rt3_ldsbar -DVLG_AF_RT=3 -DVLG_AF_WPE=4 -DVLG_ABL_AF_LDSBAR; build af_rt3_s -DVLG_AF_RT=3 -DVLG_AF_WPE=4 -DVLG_ABL_AF_NOMFMA
build af_rt3_apb16 -DVLG_AF_RT=3 -DVLG_AF_WPE=4 -DVLG_AF_APB=16; build af_rt3_apb16_ldsbar -DVLG_AF_RT=3 -DVLG_AF_WPE=4 -DVLG_AF_APB=16 -DVLG_ABL_AF_LDSBAR; build af_rt3_apb16_s -DVLG_AF_RT=3 -DVLG_AF_WPE=4 -DVLG_AF_APB=16 -DVLG_ABL_AF_NOMFMA
build af_seq_al -DVLG_ABL_AF_NOMFMA -DVLG_ABL_AF_SEQ -DVLG_ABL_AF_SEQ_STRIDE=2944; build af_seq_al_full -DVLG_ABL_AF_SEQ -DVLG_ABL_AF_SEQ_STRIDE=2944
[ "$1" = build ] && exit 0
for v in ${VARIANTS:-af_full af_nomfma af_nostore af_seq af_plain}; do
  echo "== $v"; VLG_SKIP_CHECK=1 VLGAE_AMD_LIB=$PWD/tools/variants/lib_$v.so python tools/time_align_full.py 2>&1 | grep "direct-store\|fill_"
done
