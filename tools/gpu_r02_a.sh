#!/bin/bash
# round-2 GPU pass A: full GPU test suite, the default bench line, rocprofv3 evidence (N=32 + N=128/512)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_a.log
tail -5 gpurun_out/pytest_a.log
timeout 600 python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"
cat gpurun_out/bench_a.json
timeout 900 bash tools/profile_round.sh r02 > gpurun_out/profile_a.log 2>&1; echo "profile rc=$?"
for n in 128 512; do
  O=$R/gpurun_out/prof_r02_n$n; rm -rf $O; mkdir -p $O
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --batch $n --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --batch $n --steps 10 --warmup 2 --repeats 2 --min-ms 1 --no-cpu-baseline --no-extra > $O/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --batch $n --steps 10 --warmup 2 --repeats 2 --min-ms 1 --no-cpu-baseline --no-extra > $O/pmc_write.log 2>&1)
  find $O -name "*kernel_trace.csv" -delete   # keep the merge-back small
  find $O -name "*.db" -delete
done
find $R/gpurun_out/prof_r02 -name "*kernel_trace.csv" -delete; find $R/gpurun_out -name "*.db" -delete
du -sh gpurun_out
