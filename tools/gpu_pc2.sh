#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
APA_PC_EXP=4 bash tools/prof_dense.sh pcdw --workload perclass 2>&1 | grep -E "pc_bwd_dw|pc_dw_reduce"
for s in 8 32; do echo "splits=$s"; APA_PC_DW_SPLITS=$s bash tools/prof_dense.sh pcdw --workload perclass 2>&1 | grep -E "pc_bwd_dw|pc_dw_reduce"; done
