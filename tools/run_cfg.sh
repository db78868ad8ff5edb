set -e
export TMPDIR=/tmp
python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "bitwise or baseline_config or unipc or spc" > gpurun_out/pytest_quick.log 2>&1 || { tail -30 gpurun_out/pytest_quick.log; exit 1; }
tail -2 gpurun_out/pytest_quick.log
python tools/bench_configs.py > gpurun_out/bench_configs2.log 2>&1
cat gpurun_out/bench_configs2.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg2 -o n --output-format csv -- python3 tools/bench_configs.py > gpurun_out/prof_cfg2.log 2>&1
