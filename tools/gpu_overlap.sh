#!/bin/bash
# 1-rank cost of the data-parallel schedules (collectives are no-ops on a 1-rank group): what the N > 1
# choreography itself costs per step on one MI355X
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
run() { label=$1; shift
  python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('| %-44s | %.2f | %s |' % ('$label', d['ms_per_step']*1e3, d['config']['allreduce']))"
}
for i in 1 2; do
run "no process group (plain 1-GPU bench)"
run "1-rank group, one in-stream bucket (--overlap off)" --force-dist --overlap off
run "1-rank group, overlapped schedule (--overlap on)" --force-dist --overlap on
run "1-rank group, torch.distributed RCCL, one bucket" --force-dist --comm torch --overlap off
done
