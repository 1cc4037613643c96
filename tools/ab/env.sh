# same-box A/B of two environments with one build: bench.py alternately with and without the variables given (development tool)
# usage: tools/ab/env.sh VAR=value [VAR=value ...]
for r in 1 2 3; do for v in base var; do
if [ $v = var ]; then pre="env $@"; else pre=""; fi
$pre python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); c=d['roofline']['conv_ms_per_step']; print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'fwd', round(c['conv_l16_fwd_kernel'],3), 'wgrad', round(c['conv_l16_wgrad_kernel'],3))"
done; done
