# same-box A/B of an environment switch on the cfg-3 step:  tools/ab/cfg3_env.sh VAR=off_value   (development tool)
for r in 1 2 3; do for v in off on; do
if [ $v = off ]; then export $1; else unset ${1%%=*}; fi
python bench.py --workload cfg3 --steps 20 --warmup 5 --graph --no-cpu-baseline --no-other 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg3 $1 $v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
