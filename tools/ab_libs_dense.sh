#!/bin/bash
# interleaved A/B/C of library builds on tools/bench_dense.py workloads: ab_libs_dense.sh "<wl1> <wl2>" <reps> lib1.so lib2.so ...
WLS=$1; REPS=$2; shift 2
for r in $(seq $REPS); do
  for lib in "$@"; do
    line="$(basename $lib)"
    for w in $WLS; do
      out=$(APA_LIB_PATH=$lib python3 tools/bench_dense.py --workload $w 2>/dev/null | tail -1)
      line="$line | $w $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%.2f' % (d['ms_per_step']*1e3))" "$out" 2>/dev/null || echo ERR)"
    done
    echo "$line"
  done
done
