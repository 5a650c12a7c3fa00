#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/p3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p3 -- python $R/tools/bench_dense.py "$@" > /dev/null 2>&1
f=$(find $R/gpurun_out/p3 -name "*kernel_stats.csv"|head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"])>=50: print("%-100s %5s %9.2f us %s"%(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
