#!/bin/bash
# A/B of one knob of the development build on ONE box, interleaved: tools/ab_env.sh <workload> <VAR> <valA> <valB> [reps]
# workload: a tools/bench_dense.py --workload name (+ optional args in quotes)
W=$1; VAR=$2; A=$3; B=$4; REPS=${5:-3}
L=attentionalpoolingaction_amd/custom_ops/libapa_hip_ablate.so
for r in $(seq $REPS); do
  for v in $A $B; do
    out=$(env APA_LIB_PATH=$L $VAR=$v python3 tools/bench_dense.py --workload $W 2>/dev/null | tail -1)
    echo "$VAR=$v $(python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('%.2f us' % (d['ms_per_step']*1e3))" "$out")"
  done
done
