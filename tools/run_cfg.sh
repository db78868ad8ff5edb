set -e
export TMPDIR=/tmp
python -m pytest tests/test_noise_gpu.py tests/test_step_gpu.py -x -q -m gpu -k "color or structured_noise" > gpurun_out/pytest_quick.log 2>&1 || { tail -40 gpurun_out/pytest_quick.log; exit 1; }
tail -2 gpurun_out/pytest_quick.log
python tools/bench_noise.py 2>&1 | grep Colored
SKR_FFT_NO_SPLIT=1 python tools/bench_noise.py 2>&1 | grep Colored
python tools/bench_configs.py 2>&1 | grep cfg3
