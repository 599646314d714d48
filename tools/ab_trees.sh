# Same-box A/B of two source trees: bash tools/ab_trees.sh <old_tree> [rounds]   (run through gpurun; trees built beforehand)
# Alternates `bench.py --no-cpu-baseline` of the old tree and of this one and prints ms_per_step of each run.
OLD=$1; N=${2:-3}
for i in $(seq $N); do
  for t in "$OLD" "."; do
    ( cd $t && timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$t', round(d['ms_per_step'],4))" ) || exit 1
  done
done
