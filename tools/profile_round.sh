#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. --kernel-trace --stats of the default bench command  -> gpurun_out/prof/stats
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (kernel-trace only) -> HBM traffic of
#      the dominant kernel per launch (gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide
#      coalesced stream -> doubled; calibrated on the forward pass, which reads exactly X)
# then tools/profile_summarise.py turns the CSVs into profiles/<tag>_*.  Usage: profile_round.sh <tag> [bench args]
tag=${1:-r01}; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/prof_$tag; rm -rf $O; mkdir -p $O
cd $R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra "$@" > $O/bench_under_rocprof.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --repeats 2 --min-ms 1 "$@" > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --repeats 2 --min-ms 1 "$@" > $O/pmc_write.log 2>&1
python tools/profile_summarise.py $O $tag "$@"
