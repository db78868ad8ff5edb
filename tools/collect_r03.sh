# Round-3 evidence, one gpurun call from the repo root:  bash tools/collect_r03.sh [part ...]
#   parts: bench  (headline bench lines + rocprofv3 kernel trace + PMC / SQ passes of bench.py)
#          plans  (cold-buffer plan launches: kernel trace + FETCH / WRITE / SQ / TCC passes -> k2 / rk1 / k1 counters)
#          configs (per-config kernel stats and wrapper rates), micro (bench_plan / bench_graph), noise, colored (SQ counters of the Colored kernels)
# The raw rocprofv3 output stays in /tmp on the box (hundreds of MB); only the condensed files come back, under
# gpurun_out/r3c/profiles/ -- copy what is to be judged from there into profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
RAW=/tmp/r3raw
mkdir -p $O/profiles $RAW
PARTS="${@:-bench plans configs micro noise}"
cd /tmp && export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
T="timeout -k 10"
stats_csv() { ls $1/*kernel_stats.csv $1/*/*kernel_stats.csv 2>/dev/null | head -1; }
if has bench; then
  B="python3 $R/bench.py"
  $T 400 $B > $O/profiles/r03_bench_line.json 2> $O/bench.err
  $T 300 $B --steps 20 --warmup 5 > $O/profiles/r03_bench_line_k20.json 2>> $O/bench.err
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_trace -o prof --output-format csv -- $B --no-cpu-baseline --no-traffic > $O/prof_trace.log 2>&1
  $T 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/prof_fetch -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_fetch.log 2>&1
  $T 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/prof_write -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_write.log 2>&1
  $T 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/prof_sq -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_sq.log 2>&1
  $T 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $RAW/prof_misc -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_misc.log 2>&1 || echo "misc pass failed"
  (cd $R && SKR_PROF_RAW=$RAW SKR_PROF_OUT=$O/profiles python3 tools/summarize_profile.py r03 > $O/summarize_bench.log 2>&1) || { echo "summarize_profile failed"; tail -5 $O/summarize_bench.log; }
  echo "bench part done"; tail -c 300 $O/profiles/r03_bench_line_k20.json
fi
if has plans; then
  P="python3 $R/tools/prof_plans.py"
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/plans_trace -o n --output-format csv -- $P > $O/plans_trace.log 2>&1
  $T 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/plans_fetch -o n --output-format csv -- $P > $O/plans_fetch.log 2>&1
  $T 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/plans_write -o n --output-format csv -- $P > $O/plans_write.log 2>&1
  $T 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/plans_sq -o n --output-format csv -- $P > $O/plans_sq.log 2>&1
  $T 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $RAW/plans_tcc -o n --output-format csv -- $P > $O/plans_tcc.log 2>&1 || echo "tcc pass failed"
  for k in k2 rk1 k1; do
    python3 $R/tools/summarize_counters.py $O/profiles/r03_${k}_counters.json step_kernel_$k $RAW/plans_trace $RAW/plans_fetch $RAW/plans_write $RAW/plans_sq $RAW/plans_tcc > $O/summarize_$k.log 2>&1 || echo "summarize $k failed"
  done
  f=$(stats_csv $RAW/plans_trace); [ -n "$f" ] && cp $f $O/profiles/r03_plans_kernel_stats.csv
  grep -v amdgpu.ids $O/plans_trace.log > $O/profiles/r03_plans_cold_timings.txt || true
  echo "plans part done"
fi
if has configs; then
  $T 500 rocprofv3 --kernel-trace --stats -d $RAW/prof_cfg -o n --output-format csv -- python3 $R/tools/bench_configs.py > $O/prof_cfg.log 2>&1
  f=$(stats_csv $RAW/prof_cfg); [ -n "$f" ] && cp $f $O/profiles/r03_configs_kernel_stats.csv
  $T 300 python3 $R/tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > $O/profiles/r03_bench_configs.txt
  echo "configs part done"; cat $O/profiles/r03_bench_configs.txt
fi
if has micro; then
  $T 900 python3 $R/tools/bench_plan.py 2>&1 | grep -v amdgpu.ids > $O/profiles/r03_bench_plan.txt
  SKR_BENCH_PLACEMENT=torch $T 300 python3 $R/tools/bench_plan.py rk 2>&1 | grep -v amdgpu.ids > $O/profiles/r03_bench_plan_rk_torch_placement.txt
  $T 300 python3 $R/tools/bench_plan.py b64 2>&1 | grep -v amdgpu.ids > $O/profiles/r03_bench_plan_b64.txt
  $T 300 python3 $R/tools/bench_graph.py 2>&1 | grep -v amdgpu.ids > $O/profiles/r03_bench_graph.txt
  echo "micro part done"
fi
if has noise; then
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_noise -o n --output-format csv -- python3 $R/tools/prof_noise.py > $O/prof_noise.log 2>&1
  f=$(stats_csv $RAW/prof_noise); [ -n "$f" ] && cp $f $O/profiles/r03_noise_kernel_stats.csv
  $T 300 python3 $R/tools/bench_noise.py 2>&1 | grep -v amdgpu.ids > $O/profiles/r03_bench_noise.txt
  echo "noise part done"
fi
if has colored; then
  C="python3 $R/tools/prof_colored.py"
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/col_trace -o n --output-format csv -- $C > $O/col_trace.log 2>&1
  $T 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/col_sq -o n --output-format csv -- $C > $O/col_sq.log 2>&1
  $T 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace -d $RAW/col_sq2 -o n --output-format csv -- $C > $O/col_sq2.log 2>&1 || echo "second colored SQ pass failed"
  python3 $R/tools/summarize_counters.py $O/profiles/r03_colored_sq_counters.json colored_ $RAW/col_trace $RAW/col_sq $RAW/col_sq2 > $O/summarize_colored.log 2>&1 || echo "summarize colored failed"
  echo "colored part done"
fi
du -sh $R/gpurun_out || true
