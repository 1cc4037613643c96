# same-box A/B on the cfg-3 step (graph replay): tools/ab/libfsc_old.so against libfsc_new.so (development tool)
for r in 1 2 3; do for v in old new; do cp tools/ab/libfsc_$v.so freesound_classification_amd/libfsc_hip.so
python bench.py --workload cfg3 --steps 20 --warmup 5 --graph --no-cpu-baseline --no-other 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg3 $v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
cp tools/ab/libfsc_new.so freesound_classification_amd/libfsc_hip.so
