# development: cfg 3 (HIP-graph replay) with the one-workgroup BatchNorm units off (0), 256-thread units only (2), all (1)
for i in 1 2; do for f in ${MODES:-0 2 1}; do
FSC_BN_FUSED=$f python bench.py --workload cfg3 --steps 20 --warmup 5 --graph --no-cpu-baseline --no-other 2> /dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FSC_BN_FUSED=$f', d['value'], d['ms_per_step'], d['final_loss'])"
done; done
