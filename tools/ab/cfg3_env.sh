# development: cfg 3 (HIP-graph replay) under settings of one environment variable:  tools/ab/cfg3_env.sh NAME v1 v2 ...
NAME=$1; shift
for i in 1 2; do for v in "$@"; do
env $NAME=$v python bench.py --workload cfg3 --steps 20 --warmup 5 --graph --no-cpu-baseline --no-other 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NAME=$v', d['value'], d['ms_per_step'], d['final_loss'])"
done; done
