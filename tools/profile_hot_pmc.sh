#!/bin/bash
# SQ counters of the cfg 002 hot path (bench.py default workload), one rocprofv3 --pmc pass (kernel-trace
# only) -> profiles/<tag>_hot_pmc.md.  Run through gpurun, then copy gpurun_out/profiles_out/* into profiles/.
tag=${1:-r01}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/profiles_out
O=$R/gpurun_out/pmc_${tag}_hot; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $O -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --repeats 2 --min-ms 1 > $O/bench.log 2>&1
f=$(find $O -name "*counter_collection.csv" | head -1)
python - "$f" "$R/gpurun_out/profiles_out/${tag}_hot_pmc.md" <<'PY'
import csv, sys, glob, os
from collections import defaultdict
src, dst = sys.argv[1:3]
def short(n):
    n = n.replace('(anonymous namespace)::', '').split('(')[0]
    return n[5:] if n.startswith('void ') else n
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for r in csv.DictReader(open(glob.glob(os.path.join(os.path.dirname(src), '*kernel_trace.csv'))[0])):
    dur[short(r['Kernel_Name'])].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for r in csv.DictReader(open(src)):
    n = short(r['Kernel_Name'])
    if n.startswith('apa::'): acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
with open(dst, 'w') as o:
    o.write('# SQ counters of the cfg 002 step, `python bench.py` (MI355X, rocprofv3 --pmc, one pass)\n\n'
            'Per launch, averaged.  `active` / `wait_any` / `wait_inst` are fractions of SQ_WAVE_CYCLES '
            '(issuing / parked on s_waitcnt or a barrier / issue-stalled; disjoint, sum ~ 1).  The two streaming '
            'kernels spend their wave-cycles parked on memory (`wait_any`), as an HBM-bound kernel should; '
            '`VALU insts` and `VMEM reads` are wave-level instruction counts per launch; `MFMA util` as in '
            'the dense tables (the small exact-fp32 MFMA kernels are latency-, not MFMA-bound).\n\n')
    o.write('| kernel | launches | avg us (this run) | active | wait_any | wait_inst | VALU insts | VMEM reads | MFMA util |\n|---|---|---|---|---|---|---|---|---|\n')
    for n, d in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
        m = {k: sum(v) / len(v) for k, v in d.items()}
        wc = max(m.get('SQ_WAVE_CYCLES', 0), 1); dn = sum(dur[n]) / max(len(dur[n]), 1)
        o.write('| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.3g | %.3g | %.3f |\n' % (
            n[:80], len(d.get('SQ_WAVE_CYCLES', [])), dn / 1e3, m.get('SQ_ACTIVE_INST_ANY', 0) / wc,
            m.get('SQ_WAIT_ANY', 0) / wc, m.get('SQ_WAIT_INST_ANY', 0) / wc, m.get('SQ_INSTS_VALU', 0),
            m.get('SQ_INSTS_VMEM_RD', 0), m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(dn * 2.4 * 1024, 1)))
print(open(dst).read())
PY
