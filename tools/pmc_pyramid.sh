# SQ counters of the Pyramid kernels: bash tools/pmc_pyramid.sh  (one gpurun call, from the repo root)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/prof_pyramid.py
SKR_PYR_MODE=1 python3 $R/tools/prof_pyramid.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pyr_trace -o n --output-format csv -- python3 $R/tools/prof_pyramid.py > $R/gpurun_out/pyr_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/pyr_sq -o prof --output-format csv -- python3 $R/tools/prof_pyramid.py > $R/gpurun_out/pyr_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/pyr_sq2 -o prof --output-format csv -- python3 $R/tools/prof_pyramid.py > $R/gpurun_out/pyr_sq2.log 2>&1
head -12 $R/gpurun_out/pyr_trace/n_kernel_stats.csv | cut -c1-150
