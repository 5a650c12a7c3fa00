#!/bin/bash
# round-end evidence collection (through gpurun): headline stats + PMC traffic, dense summaries, SQ counters on the
# final binaries, the vendor-library yardstick, the default bench line.   Usage: gpu_final.sh <tag>
tag=${1:-r05}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
timeout 600 bash tools/profile_round.sh $tag > gpurun_out/final_round.log 2>&1; echo "profile_round rc=$?"
timeout 600 bash tools/profile_dense.sh $tag > gpurun_out/final_dense.log 2>&1; echo "profile_dense rc=$?"
timeout 600 bash tools/profile_dense_pmc.sh $tag > gpurun_out/final_dense_pmc.log 2>&1; echo "profile_dense_pmc rc=$?"
timeout 300 bash tools/profile_hot_pmc.sh $tag > gpurun_out/final_hot_pmc.log 2>&1; echo "profile_hot_pmc rc=$?"
python tools/hipblaslt_ref.py > gpurun_out/${tag}_hipblaslt_ref.log 2>&1; cat gpurun_out/${tag}_hipblaslt_ref.log
bash tools/prof_variant.sh posebwd_beta0 "--workload posebwd" | tee gpurun_out/${tag}_posebwd_beta0.log
bash tools/prof_variant.sh posebwd_beta1 "--workload posebwd_acc" | tee gpurun_out/${tag}_posebwd_beta1.log
timeout 600 bash tools/profile_e2e.sh $tag > gpurun_out/final_e2e.log 2>&1; echo "profile_e2e rc=$?"
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fwd', d['roofline_fwd']['frac'], 'step', d['step_roofline_frac'])
for k, v in d['extra'].items():
    print(k, {kk: v[kk] for kk in ('ms_per_step', 'images_per_sec', 'error') if kk in v})
PY
