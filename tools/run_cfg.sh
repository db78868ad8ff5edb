set -e
export TMPDIR=/tmp
python -m pytest tests/test_noise_gpu.py -x -q -m gpu -k "color" > gpurun_out/pytest_quick.log 2>&1 || { tail -40 gpurun_out/pytest_quick.log; exit 1; }
tail -2 gpurun_out/pytest_quick.log
python tools/bench_noise.py > gpurun_out/bench_noise2.log 2>&1 || tail -5 gpurun_out/bench_noise2.log
grep Colored gpurun_out/bench_noise2.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_noise2 -o n --output-format csv -- python3 tools/prof_noise.py > gpurun_out/prof_noise2.log 2>&1
