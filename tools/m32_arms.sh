#!/bin/bash
# the 32x32x16-MFMA form of the ring GEMM, measured (development build).
# arms: 0 = shipped (160 x 128 tiles, 16x16x32), 16 = 192 x 128 tiles with 16x16x32, 6 = 192 x 128 with 32x32x16, 8 = 256 x 128 with 32x32x16
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
L=$R/attentionalpoolingaction_amd/custom_ops/libapa_hip_ablate.so
echo "== correctness of the 32x32x16 arms (float64 products of the bf16-rounded operands, tests/test_gemm_ring_gpu.py + the cfg 003 parity tests)"
for m in 6 8; do
  APA_LIB_PATH=$L APA_GEMM_M32=$m python -m pytest $R/tests/test_gemm_ring_gpu.py $R/tests/test_bf16_parity_gpu.py -q -x -k "not perclass" 2>&1 | tail -1
done
for m in 0 16 6 8; do
  $R/tools/prof_variant.sh m32_$m "--workload cfg003" APA_GEMM_M32=$m 2>&1 | grep -i "ring\|=="
done
echo "== SQ counters"
cd /tmp; export TMPDIR=/tmp
for m in 0 16 6 8; do
  O=$R/gpurun_out/pmc_m32_$m; rm -rf $O; mkdir -p $O
  APA_LIB_PATH=$L APA_GEMM_M32=$m timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
     --output-format csv -d $O -- python $R/tools/bench_dense.py --workload cfg003 --steps 10 --warmup 2 > $O/bench.log 2>&1
  python3 - "$O" "$m" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O, m = sys.argv[1:3]
f = glob.glob(os.path.join(O, '**', '*counter_collection.csv'), recursive=True)[0]
tr = glob.glob(os.path.join(O, '**', '*kernel_trace.csv'), recursive=True)[0]
dur = defaultdict(list)
for r in csv.DictReader(open(tr)):
    dur[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'ring' in r['Kernel_Name']:
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for n, d in acc.items():
    mm = {k: sum(v) / len(v) for k, v in d.items()}
    wc = max(mm.get('SQ_WAVE_CYCLES', 0), 1); dn = sum(dur[n]) / len(dur[n]); lds = mm.get('SQ_LDS_IDX_ACTIVE', 0)
    print('arm %s | %s | %.1f us | MFMA util %.3f | active %.2f | wait_any %.2f | wait_inst %.2f | lds conflict %s' % (
        m, n.split('(')[0].replace('void apa::(anonymous namespace)::', '')[:60], dn / 1e3,
        mm.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(dn * 2.4 * 1024, 1), mm.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        mm.get('SQ_WAIT_ANY', 0) / wc, mm.get('SQ_WAIT_INST_ANY', 0) / wc,
        ('%.4f' % (mm.get('SQ_LDS_BANK_CONFLICT', 0) / lds)) if lds else '-'))
PY
done
