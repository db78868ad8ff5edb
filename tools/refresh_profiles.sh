set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python3 $R/bench.py > $R/gpurun_out/bench_line.json 2> $R/gpurun_out/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -o r01 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o r01 --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o r01 --output-format csv -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1
cat $R/gpurun_out/bench_line.json
