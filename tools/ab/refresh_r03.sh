# Refresh of the round-3 bench lines and the C* kernel statistics after a kernel change (run from the repo root on the GPU box).
mkdir -p gpurun_out/fin
bash tools/bench_all.sh fin/r03 || exit 1
for w in "C*-cov" "C2-cov" "C*wR-cov" "C5-cov"; do
  n=$(echo "$w" | sed 's/\*/star/')
  timeout -k 10 900 python3 bench.py --workload "$w" > gpurun_out/fin/r03_train_${n}_bench.json 2> gpurun_out/fin/${n}.err
  echo "$w rc=$?"
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/fin/r03_train_${n}_bench.json') if l.startswith('{')][-1])
print(round(d['ms_per_step'],3), d['step_ms'], round(d['roofline']['frac'],4), d['roofline']['kernel'])" || exit 1
done
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fin/kt -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $R/gpurun_out/fin/kt.log 2>&1
echo "kernel trace rc=$?"
cd $R
cp $(ls gpurun_out/fin/kt/*/*kernel_stats.csv | head -1) gpurun_out/fin/r03_train_Cstar_kernel_stats.csv
find gpurun_out/fin/kt -type f -delete
head -12 gpurun_out/fin/r03_train_Cstar_kernel_stats.csv | cut -c1-150
