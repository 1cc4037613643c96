# development: how often the cfg-3 bench ends in a non-finite loss under the given environment
#   tools/nan_hunt.sh RUNS [VAR=value ...]
runs=$1; shift
bad=0
for i in $(seq 1 $runs); do
  env "$@" python bench.py --workload cfg3 --steps 20 --warmup 5 > /tmp/nh.json 2> /tmp/nh.err || bad=$((bad+1))
done
echo "env [$@]: $bad of $runs runs failed; last error: $(grep -v amdgpu /tmp/nh.err | tail -1)"
