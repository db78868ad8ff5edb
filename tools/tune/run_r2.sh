#!/bin/bash
set -e
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r2tune; mkdir -p $O
T=tools/tune/tune_r2
python -m pytest tests/test_step_gpu.py -x -q -m gpu -k "one_trip or tile_layout or runge_kutta_stage or baseline_config" > $O/pytest_quick.log 2>&1 || { tail -30 $O/pytest_quick.log; exit 1; }
$T place=torch > $O/torch9.log 2>&1
python bench.py --steps 400 --warmup 40 --no-cpu-baseline > $O/bench_e.json 2> $O/bench_e.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f.json 2> $O/bench_f.err
cat $O/torch9.log
