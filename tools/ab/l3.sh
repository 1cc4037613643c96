# same-box A/B of conv_l3_fwd_kernel variants: tools/l3_check.py --arith 10 with each library given, two rounds (development tool)
#   tools/ab/l3.sh "base defer" [layer names]
LIBS=$1; shift
cp freesound_classification_amd/libfsc_hip.so /tmp/libfsc_keep.so
for r in 1 2; do for v in $LIBS; do cp tools/ab/libfsc_$v.so freesound_classification_amd/libfsc_hip.so; echo "== $v (round $r)"
timeout 300 python tools/l3_check.py --arith 10 --iters 20 ${@:-b0c2 b1e b2e b2c2 b3c2 b0c1} 2>&1 | grep -E "fwd |dgrad" | sed -E "s/max\|diff\|.*\| new/| new/"; done; done
cp /tmp/libfsc_keep.so freesound_classification_amd/libfsc_hip.so
