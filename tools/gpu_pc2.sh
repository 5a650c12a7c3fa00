#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
APA_PC_EXP=1 bash tools/prof_dense.sh pcdx --workload perclass 2>&1 | grep -E "pc_dx|gemm_bf16"
APA_PC_EXP=1 APA_PC_DX=0 bash tools/prof_dense.sh pcdx --workload perclass 2>&1 | grep -E "pc_dx|gemm_bf16"
