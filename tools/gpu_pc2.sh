#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
for e in 0 16 32 64 48 112; do
echo "exp=$e"; APA_PC_EXP=$e bash tools/prof_dense.sh pcdma --workload perclass 2>&1 | grep -E "pc_fwd_zt"
done
