#!/bin/bash
# SQ counters of the MFMA-bound rows (cfg 003 pose head, per-class maps), one rocprofv3 --pmc pass per
# workload (kernel-trace only; 8 SQ slots on gfx950) -> profiles/<tag>_<workload>_pmc.md.
# Run through gpurun, then copy gpurun_out/profiles_out/* into profiles/.   Usage: profile_dense_pmc.sh <tag>
tag=${1:-r01}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/profiles_out
for wl in cfg003 perclass; do
O=$R/gpurun_out/pmc_${tag}_$wl; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $O -- python $R/tools/bench_dense.py --workload $wl --steps 10 --warmup 2 > $O/bench.log 2>&1
f=$(find $O -name "*counter_collection.csv" | head -1)
python - "$f" "$R/gpurun_out/profiles_out/${tag}_${wl}_pmc.md" "$wl" <<'PY'
import csv, sys
from collections import defaultdict
src, dst, wl = sys.argv[1:4]
import glob, os
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)     # kernel -> launch durations (ns) in this same (counter-collecting) run
def short(n):
    n = n.replace('(anonymous namespace)::', '').split('(')[0]
    return n[5:] if n.startswith('void ') else n
for r in csv.DictReader(open(glob.glob(os.path.join(os.path.dirname(src), '*kernel_trace.csv'))[0])):
    dur[short(r['Kernel_Name'])].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for r in csv.DictReader(open(src)):
    n = short(r['Kernel_Name'])
    if 'gemm' in n or 'pc_' in n or 'pose_' in n:
        acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
with open(dst, 'w') as o:
    o.write('# SQ counters, `python tools/bench_dense.py --workload %s` (MI355X, rocprofv3 --pmc, one pass)\n\n' % wl)
    o.write('Per launch, averaged over the launches of the run.  `MFMA util` = SQ_VALU_MFMA_BUSY_CYCLES / '
            '(launch duration in this run x 2.4 GHz x 1024 SIMDs): the share of all SIMD-cycles of the chip with '
            'the matrix pipe busy (durations under counter collection are a few percent longer than un-profiled, '
            'so this is a slight under-estimate).  `active` / `wait_any` / `wait_inst` are fractions of '
            'SQ_WAVE_CYCLES (disjoint, sum ~ 1): issuing, parked on s_waitcnt / barrier, issue-stalled.  '
            '`lds conflict` = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.\n\n')
    o.write('| kernel | launches | avg us (this run) | MFMA util | active | wait_any | wait_inst | lds conflict |\n|---|---|---|---|---|---|---|---|\n')
    for n, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
        m = {k: sum(v) / len(v) for k, v in d.items()}
        wc = max(m.get('SQ_WAVE_CYCLES', 0), 1)
        lds = m.get('SQ_LDS_IDX_ACTIVE', 0)
        dn = sum(dur[n]) / max(len(dur[n]), 1)
        o.write('| `%s` | %d | %.1f | %.3f | %.2f | %.2f | %.2f | %s |\n' % (
            n[:90], len(d.get('SQ_WAVE_CYCLES', [])), dn / 1e3,
            m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(dn * 2.4 * 1024, 1),
            m.get('SQ_ACTIVE_INST_ANY', 0) / wc, m.get('SQ_WAIT_ANY', 0) / wc, m.get('SQ_WAIT_INST_ANY', 0) / wc,
            ('%.4f' % (m.get('SQ_LDS_BANK_CONFLICT', 0) / lds)) if lds else '-'))
print(open(dst).read())
PY
done
