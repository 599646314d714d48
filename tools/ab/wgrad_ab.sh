mkdir -p gpurun_out/pk
timeout -k 10 500 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -k "mlp_wgrad_kernels" > gpurun_out/pk/test3.log 2>&1; tail -3 gpurun_out/pk/test3.log
timeout -k 10 300 python tools/wgrad_probe.py > gpurun_out/pk/probe3.log 2>&1; cat gpurun_out/pk/probe3.log
for wl in "C*" C3 C5 C2 "C*-cov"; do
  for old in 1 0; do
    PSVO_WGRAD_OLD=$old timeout -k 10 300 python bench.py --workload "$wl" --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', 'old' if $old else 'new', round(d['ms_per_step'],4), d['step_ms']['median'], d['config']['native_ms_per_step'].get('psvo_mlp_wgrad'))" || exit 1
  done
done
