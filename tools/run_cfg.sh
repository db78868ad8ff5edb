set -e
export TMPDIR=/tmp
python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "one_trip or baseline_config or two_output or indexed" > gpurun_out/pytest_quick.log 2>&1 || { tail -30 gpurun_out/pytest_quick.log; exit 1; }
tail -1 gpurun_out/pytest_quick.log
tools/tune/tune_r2 place=torch > gpurun_out/tune_x.log 2>&1
cat gpurun_out/tune_x.log
for i in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_us_per_launch'], d['roofline']['frac'], d['graph_loop_cfg2']['graph_steps_per_s'])"; done
