#!/bin/bash
t() { APA_DBG_SKIP=$1 python bench.py --steps 400 --warmup 40 --no-cpu-baseline "${@:3}" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=%-4s %-16s step %.2f us'%('$1','$2',d['ms_per_step']*1e3))"; }
t 0 none "$@"; t 0 none "$@"
t 1 pool "$@"; t 2 finalize "$@"; t 4 logits_partial "$@"; t 8 logits_reduce "$@"; t 16 xent "$@"; t 32 bwd_small "$@"; t 64 bwd_main "$@"; t 128 colsum "$@"
t 62 only_streams+colsum "$@"; t 190 only_stream_kernels "$@"; t 255 nothing "$@"
