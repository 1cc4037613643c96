# same-box A/B of two builds of the library: bench.py alternately with tools/ab/libfsc_old.so and libfsc_new.so (development tool)
for r in 1 2 3; do for v in ${AB:-old new}; do cp tools/ab/libfsc_$v.so freesound_classification_amd/libfsc_hip.so
python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); c=d['roofline']['conv_ms_per_step']; print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'fwd', round(c['conv_l16_fwd_kernel'],3), 'wgrad', round(c['conv_l16_wgrad_kernel'],3))"
done; done
cp tools/ab/libfsc_new.so freesound_classification_amd/libfsc_hip.so
