# Every workload of BASELINE.json (+ the PSVOwR and two-layer lines) through bench.py on one box, WITH the cpu_baseline and
# elbo_vs_oracle legs:   bash tools/bench_all.sh <tag>   -> gpurun_out/<tag>_train_<workload>_bench.json  (copy into profiles/)
TAG=${1:-r}
for w in "C*" C2 C3 C4 C5 "C*wR" "C*-2x32" "C*-2x64"; do
  n=$(echo "$w" | sed 's/\*/star/')
  timeout -k 10 900 python3 bench.py --workload "$w" > gpurun_out/${TAG}_train_${n}_bench.json 2> gpurun_out/${TAG}_${n}.err
  echo "$w rc=$?"
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/${TAG}_train_${n}_bench.json') if l.startswith('{')][-1])
print(round(d['ms_per_step'],3), d['step_ms'], round(d['roofline']['frac'],4), d['roofline']['kernel'], d.get('cpu_baseline',{}).get('value'), {k: d.get('elbo_vs_oracle',{}).get(k) for k in ('rel_err','flipped_draws','draws','max_abs_trajectory_err')})" || exit 1
done
