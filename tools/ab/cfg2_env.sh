# development: cfg 2 under settings of one environment variable:  tools/ab/cfg2_env.sh NAME v1 v2 ...
NAME=$1; shift
for i in 1 2; do for v in "$@"; do
env $NAME=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other --no-alt --no-traffic 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NAME=$v', d['value'], d['ms_per_step'])"
done; done
