#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/c10_dx.log
for e in 0 128; do
bash tools/prof_variant.sh pc_dwe$e "--workload perclass" APA_PC_DW_EXP=$e 2>&1 | grep -E "==|pc_" | tee -a gpurun_out/c10_dx.log
done
