# Collects the round's headline evidence on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel-trace stats, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes), SQ counters.
# tools/summarize_profile.py <round> then condenses gpurun_out/prof_* into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py"
timeout -k 10 300 $B > $R/gpurun_out/bench_line.json 2> $R/gpurun_out/bench.err
timeout -k 10 300 $B --steps 20 --warmup 5 > $R/gpurun_out/bench_line_k20.json 2>> $R/gpurun_out/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -o prof --output-format csv -- $B --no-cpu-baseline --no-traffic > $R/gpurun_out/prof_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/prof_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/prof_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/prof_sq -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/prof_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace -d $R/gpurun_out/prof_sq2 -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/prof_sq2.log 2>&1 || echo "second SQ pass failed (counter names)"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/prof_misc -o prof --output-format csv -- $B --steps 100 --warmup 10 --no-cpu-baseline --no-traffic > $R/gpurun_out/prof_misc.log 2>&1 || echo "misc pass failed"
cat $R/gpurun_out/bench_line.json
