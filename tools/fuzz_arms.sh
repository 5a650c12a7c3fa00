#!/bin/bash
# the random-shape sweeps once per A/B arm of the library (the fallback kernels are real code paths for the
# shapes the fast ones do not cover).  The product build has no run-time knobs: this script rebuilds the library
# with -DAPA_ABLATION into libapa_hip_ablate.so (knob() then reads the environment) and points APA_LIB_PATH at it.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
make -C attentionalpoolingaction_amd/csrc -j16 ABLATE=1 >/dev/null 2>&1 || exit 1     # libapa_hip_ablate.so, beside the product library
export APA_LIB_PATH=$R/attentionalpoolingaction_amd/custom_ops/libapa_hip_ablate.so
m1() { echo "== $*"; env "$@" timeout 300 python tools/fuzz_attn_pool.py 120 13 2>&1 | grep -E "^FAIL|cases," | cut -c1-260 | head -6
       env "$@" timeout 300 python tools/fuzz_all.py 50 13 step,bf16m1 2>&1 | grep -E "^FAIL|cases," | cut -c1-260 | head -6; }
dense() { echo "== $*"; env "$@" timeout 300 python tools/fuzz_all.py 70 13 perclass,pose 2>&1 | grep -E "^FAIL|cases," | cut -c1-260 | head -6; }
m1 APA_M1_BWD_HEAD=0
m1 APA_M1_STREAM=0
m1 APA_M1_LOGITS2=0
m1 APA_M1_LOGITS_XENT=0
m1 APA_M1_GEMV_BWD2=0 APA_M1_KEEP_BITS=0
dense APA_PC_FUSED=0
dense APA_GEMM_RING=0
dense APA_GEMM_WIDE=0
dense APA_POSE_STEP_FUSED=0 APA_PC_XENT_FOLD=0
dense APA_PC_ACT_FOLD=0
dense APA_PC_DX_FUSED=0
dense APA_GEMM_TWIN=0 APA_GEMM_NT=0
dense APA_PC_CAT=0
dense APA_GEMM_GLDS=0
dense APA_GEMM_FAST=0
dense APA_POSE_BWD_ROWS=0 APA_POSE_PL_FAST=0
# round 5: the caller-kept-state family (weight images + tagged keep bits) on the arms it crosses
wimg() { echo "== wimg $*"; env "$@" timeout 300 python tools/fuzz_all.py 60 13 wimg 2>&1 | grep -E "^FAIL|cases," | cut -c1-260 | head -6; }
wimg APA_PC_TAGGED_BITS=0
wimg APA_PC_DX_FUSED=0
wimg APA_PC_ACT_FOLD=0 APA_PC_XENT_FOLD=0
wimg APA_PC_FUSED=0
