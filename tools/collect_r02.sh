# Round-2 evidence, one gpurun call from the repo root:  bash tools/collect_r02.sh
# (headline bench + rocprofv3 kernel trace + PMC passes, per-config / noise / fp32 kernel stats, plan microbenchmarks)
set -e
R=$GRAFT_REPO_ROOT
bash $R/tools/refresh_profiles.sh > $R/gpurun_out/refresh.log 2>&1 || { tail -5 $R/gpurun_out/refresh.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cfg -o n --output-format csv -- python3 $R/tools/bench_configs.py > $R/gpurun_out/prof_cfg.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_noise -o n --output-format csv -- python3 $R/tools/prof_noise.py > $R/gpurun_out/prof_noise.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp32 -o n --output-format csv -- python3 $R/tools/prof_fp32.py > $R/gpurun_out/prof_fp32.log 2>&1
python3 $R/tools/bench_configs.py > $R/gpurun_out/bench_configs.log 2>&1
python3 $R/tools/bench_plan.py > $R/gpurun_out/bench_plan.log 2>&1
python3 $R/tools/bench_noise.py > $R/gpurun_out/bench_noise.log 2>&1
python3 $R/tools/bench_graph.py > $R/gpurun_out/bench_graph.log 2>&1
cd $R && python3 -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/smoke.log 2>&1 && echo SMOKE OK
tail -2 $R/gpurun_out/smoke.log
cat $R/gpurun_out/bench_configs.log $R/gpurun_out/bench_graph.log
