set -e
export TMPDIR=/tmp
for k in 20 20 400; do python bench.py --steps $k --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['steps'], round(d['value']), round(d['ms_per_step']*1e3,2), round(r['kernel_us_per_launch'],2), round(r['event_span_us_per_launch'],2), round(r['frac'],4))"; done
cd /tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_k20 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_k20.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_k20.log | cut -c1-50
