# Every workload of BASELINE.json (+ the PSVOwR line) through bench.py on one box: bash tools/bench_all.sh <tag>
# -> gpurun_out/<tag>_train_<workload>_bench.json   (copy the ones to keep into profiles/)
TAG=${1:-r}
for w in "C*" C2 C3 C4 C5 "C*wR"; do
  n=$(echo "$w" | sed 's/\*/star/')
  timeout -k 10 500 python3 bench.py --workload "$w" --no-cpu-baseline > gpurun_out/${TAG}_train_${n}_bench.json 2> gpurun_out/${TAG}_${n}.err
  echo "$w rc=$?"
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_train_${n}_bench.json') if l.startswith('{')][-1])
print(round(d['ms_per_step'],3), d['config']['native_ms_per_step'], d['config'].get('launch'))" || exit 1
done
