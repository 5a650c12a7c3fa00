#!/bin/bash
# round-2 evidence pass: full GPU test suite, the default bench line, rocprofv3 evidence (N=32 + N=128/512,
# cfg003, per-class)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_final.log
tail -4 gpurun_out/pytest_final.log
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
timeout 900 bash tools/profile_round.sh r02 > gpurun_out/profile_final.log 2>&1; echo "profile rc=$?"
for n in 128 512; do
  O=$R/gpurun_out/prof_r02_n$n; rm -rf $O; mkdir -p $O
  echo "python bench.py --batch $n --steps 50 --warmup 5 --no-cpu-baseline --no-extra" > $O/cmd.txt
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --batch $n --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $O/bench.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --batch $n --steps 10 --warmup 2 --repeats 2 --min-ms 1 --no-cpu-baseline --no-extra > $O/pmc_fetch.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --batch $n --steps 10 --warmup 2 --repeats 2 --min-ms 1 --no-cpu-baseline --no-extra > $O/pmc_write.log 2>&1)
done
bash tools/prof_dense.sh r02_cfg003 --workload cfg003 > gpurun_out/prof_r02_cfg003.txt 2>&1
bash tools/prof_dense.sh r02_perclass --workload perclass > gpurun_out/prof_r02_perclass.txt 2>&1
bash tools/prof_dense.sh r02_perclass393 --workload perclass --classes 393 > gpurun_out/prof_r02_perclass393.txt 2>&1
bash tools/prof_dense.sh r02_eval002 --workload eval002 > gpurun_out/prof_r02_eval002.txt 2>&1
find $R/gpurun_out -name "*kernel_trace.csv" -delete; find $R/gpurun_out -name "*.db" -delete
cat gpurun_out/bench_final.json | cut -c1-600
