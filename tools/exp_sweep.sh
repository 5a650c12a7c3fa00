#!/bin/bash
# A/B sweep of the headline step: env knobs of the streaming kernels + an alternative library build
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%-28s step %.2f us  fwd %.2f  bwd %.2f  (min %.2f max %.2f)' % ('$label', d['ms_per_step']*1e3, d['roofline_fwd']['kernel_avg_us'], d['roofline']['kernel_avg_us'], d['ms_per_step_min_max'][0]*1e3, d['ms_per_step_min_max'][1]*1e3))"
}
run base A=1
run base_again A=1
run nt_loads APA_LIB_PATH=$R/attentionalpoolingaction_amd/custom_ops/libapa_hip_nt.so
run blocks384 APA_M1_TARGET_BLOCKS=384
run blocks768 APA_M1_TARGET_BLOCKS=768
run blocks1024 APA_M1_TARGET_BLOCKS=1024
run pix1 APA_M1S_PIX=1
run pix4 APA_M1S_PIX=4
