# Round-5 evidence, one gpurun call from the repo root:  bash tools/collect_r05.sh [part ...]
#   parts: bench    (bench lines of EVERY BASELINE config: headline at K=400 and at the driver's K=20, cfg2, cfg3, cfg3c, cfg4, cfg5)
#          trace    (rocprofv3 --kernel-trace --stats of the same bench commands -> profiles/r05_kernel_stats_<config>.csv)
#          pmc      (headline PMC / SQ passes -> r05_pmc_traffic.json, r05_sq_counters.json)
#          plans    (cold-buffer plan launches: kernel trace + counters)
#          micro    (bench_plan / bench_graph), noise, colored / pyramid (SQ counters of those generators' kernels), fftown (awkward Colored shapes: own transforms vs hipFFT),
#          margins (measured parity maxima), rehearsal (two torchrun ranks on the one card at the driver's K = 20 / W = 5)
# The raw rocprofv3 output stays in /tmp on the box (hundreds of MB); only the condensed files come back, under
# gpurun_out/r5c/profiles/ -- copy what is to be judged from there into profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5c
RAW=/tmp/r5raw
mkdir -p $O/profiles $RAW
PARTS="${@:-bench trace}"
cd /tmp && export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
T="timeout -k 10"
CONFIGS="${R5_CONFIGS:-headline cfg2 cfg3 cfg3c cfg4 cfg5}"   # R5_CONFIGS="cfg5" refreshes one configuration
stats_csv() { ls $1/*kernel_stats.csv $1/*/*kernel_stats.csv 2>/dev/null | head -1; }
B="python3 $R/bench.py"
if has bench; then
  case " $CONFIGS " in *" headline "*)
    $T 400 $B > $O/profiles/r05_bench_line.json 2> $O/bench.err
    $T 300 $B --steps 20 --warmup 5 > $O/profiles/r05_bench_line_k20.json 2>> $O/bench.err;;
  esac
  for c in $CONFIGS; do
    [ $c = headline ] && continue
    $T 400 $B --config $c > $O/profiles/r05_bench_line_$c.json 2>> $O/bench.err || { echo "bench $c failed"; tail -5 $O/bench.err; }
    echo "bench $c done"
  done
fi
if has trace; then
  for c in $CONFIGS; do
    $T 300 rocprofv3 --kernel-trace --stats -d $RAW/trace_$c -o prof --output-format csv -- $B --config $c --no-cpu-baseline --no-traffic --no-extras > $O/trace_$c.log 2>&1 || { echo "trace $c failed"; tail -3 $O/trace_$c.log; }
    (cd $R && python3 tools/kernel_stats.py $RAW/trace_$c $O/profiles/r05_kernel_stats_$c.csv > $O/kernel_stats_$c.log 2>&1) || { echo "kernel_stats $c failed"; tail -3 $O/kernel_stats_$c.log; }
  done
  echo "trace part done"
fi
if has pmc; then
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_trace -o prof --output-format csv -- $B --no-cpu-baseline --no-traffic > $O/prof_trace.log 2>&1
  $T 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/prof_fetch -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_fetch.log 2>&1
  $T 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/prof_write -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_write.log 2>&1
  $T 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/prof_sq -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_sq.log 2>&1
  $T 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $RAW/prof_misc -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $O/prof_misc.log 2>&1 || echo "misc pass failed"
  (cd $R && SKR_PROF_RAW=$RAW SKR_PROF_OUT=$O/profiles python3 tools/summarize_profile.py r05 > $O/summarize_bench.log 2>&1) || { echo "summarize_profile failed"; tail -5 $O/summarize_bench.log; }
  echo "pmc part done"
fi
if has plans; then
  P="python3 $R/tools/prof_plans.py"
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/plans_trace -o n --output-format csv -- $P > $O/plans_trace.log 2>&1
  $T 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/plans_fetch -o n --output-format csv -- $P > $O/plans_fetch.log 2>&1
  $T 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/plans_write -o n --output-format csv -- $P > $O/plans_write.log 2>&1
  $T 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/plans_sq -o n --output-format csv -- $P > $O/plans_sq.log 2>&1
  for k in k2 rk1 k1; do
    python3 $R/tools/summarize_counters.py $O/profiles/r05_${k}_counters.json step_kernel_$k $RAW/plans_trace $RAW/plans_fetch $RAW/plans_write $RAW/plans_sq > $O/summarize_$k.log 2>&1 || echo "summarize $k failed"
  done
  f=$(stats_csv $RAW/plans_trace); [ -n "$f" ] && cp $f $O/profiles/r05_plans_kernel_stats.csv
  echo "plans part done"
fi
if has micro; then
  $T 900 python3 $R/tools/bench_plan.py 2>&1 | grep -v amdgpu.ids > $O/profiles/r05_bench_plan.txt
  $T 300 python3 $R/tools/bench_graph.py 2>&1 | grep -v amdgpu.ids > $O/profiles/r05_bench_graph.txt
  echo "micro part done"
fi
if has noise; then
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/prof_noise -o n --output-format csv -- python3 $R/tools/prof_noise.py > $O/prof_noise.log 2>&1
  (cd $R && python3 tools/kernel_stats.py $RAW/prof_noise $O/profiles/r05_noise_kernel_stats.csv > $O/kernel_stats_noise.log 2>&1) || echo "kernel_stats noise failed"
  $T 300 python3 $R/tools/bench_noise.py 2>&1 | grep -v amdgpu.ids > $O/profiles/r05_bench_noise.txt
  echo "noise part done"
fi
if has colored; then
  C="python3 $R/tools/prof_colored.py"
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/col_trace -o n --output-format csv -- $C > $O/col_trace.log 2>&1
  $T 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/col_sq -o n --output-format csv -- $C > $O/col_sq.log 2>&1
  $T 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace -d $RAW/col_sq2 -o n --output-format csv -- $C > $O/col_sq2.log 2>&1 || echo "second colored SQ pass failed"
  python3 $R/tools/summarize_counters.py $O/profiles/r05_colored_sq_counters.json colored_ $RAW/col_trace $RAW/col_sq $RAW/col_sq2 > $O/summarize_colored.log 2>&1 || echo "summarize colored failed"
  echo "colored part done"
fi
if has pyramid; then
  C="python3 $R/tools/prof_pyramid.py"
  $C 2>&1 | grep Pyramid > $O/profiles/r05_prof_pyramid.txt
  $T 300 rocprofv3 --kernel-trace --stats -d $RAW/pyr_trace -o n --output-format csv -- $C > $O/pyr_trace.log 2>&1
  $T 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $RAW/pyr_sq -o n --output-format csv -- $C > $O/pyr_sq.log 2>&1
  $T 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace -d $RAW/pyr_sq2 -o n --output-format csv -- $C > $O/pyr_sq2.log 2>&1 || echo "second pyramid SQ pass failed"
  python3 $R/tools/summarize_counters.py $O/profiles/r05_pyramid_sq_counters.json pyramid_pass1 $RAW/pyr_trace $RAW/pyr_sq $RAW/pyr_sq2 > $O/summarize_pyramid.log 2>&1 || echo "summarize pyramid failed"
  (cd $R && python3 tools/kernel_stats.py $RAW/pyr_trace $O/profiles/r05_pyramid_kernel_stats.csv > $O/kernel_stats_pyr.log 2>&1) || echo "kernel_stats pyramid failed"
  echo "pyramid part done"
fi
if has fftown; then
  { echo "# python tools/bench_fft_own.py (one MI355X, round 5): whole Colored draws, fp32 results; rows with 0 transforms run on the LDS plane kernels whatever the setting"; $T 300 python3 $R/tools/bench_fft_own.py 2>&1 | grep -v amdgpu.ids; } > $O/profiles/r05_bench_fft_own.txt
  echo "fftown part done"
fi
if has margins; then
  rm -f /tmp/r5_margins.jsonl
  (cd $R && SKR_PARITY_MARGINS=/tmp/r5_margins.jsonl $T 900 python3 -m pytest tests -m gpu -q -p no:cacheprovider > $O/margins_pytest.log 2>&1) || { echo "gpu tests failed"; tail -15 $O/margins_pytest.log; }
  tail -3 $O/margins_pytest.log
  (cd $R && python3 tools/summarize_margins.py /tmp/r5_margins.jsonl $O/profiles/r05_parity_margins.txt > /dev/null) || echo "summarize_margins failed"
  echo "margins part done"
fi
if has rehearsal; then
  # the N > 1 path on ONE card (both ranks on device 0, gloo instead of RCCL, which refuses two ranks on one device), at the driver's K = 20 / W = 5
  { echo "# bench.py under python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ... bench.py --gpus 2 --steps 20 --warmup 5 [--config cfg4] on ONE MI355X"
    echo "# (SKR_BENCH_DEVICE=0 SKR_BENCH_BACKEND=gloo).  A rehearsal of the N > 1 code path -- shards through BatchShard.from_env, opening barrier, every rank's wall read after ITS OWN"
    echo "# synchronise and before the closing barrier (sharding.TimedRegion), MAX over ranks, per-rank clocks in roofline.ranks -- NOT a scaling number: the two ranks share one GPU."
    for c in headline cfg4; do
      (cd $R && SKR_BENCH_DEVICE=0 SKR_BENCH_BACKEND=gloo $T 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --config $c --no-traffic 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print(json.dumps({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling')} | {'config': {k: d['config'][k] for k in ('name', 'global_batch', 'per_gpu_batch', 'parallelism')}, 'roofline.us_per_step': r['us_per_step'], 'roofline.ranks': r['ranks'], 'cpu_baseline': d.get('cpu_baseline')}))")
    done
    echo "# the RCCL code path itself (backend nccl: communicator, barriers, the per-rank all_gather) with ONE rank under torch.distributed.run:"
    (cd $R && $T 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print(json.dumps({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling')} | {'backend': 'nccl (RCCL), world 1', 'roofline.us_per_step': r['us_per_step'], 'roofline.ranks': r['ranks']}))")
  } > $O/profiles/r05_two_rank_rehearsal.txt
  echo "rehearsal part done"
fi
du -sh $R/gpurun_out || true
