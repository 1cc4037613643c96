# development: per-layer TF of the L16 forward / dgrad kernels for several library variants on one box
#   tools/ab/layers.sh "v0 v1 v2" [l16_check args]
for r in 1 2; do for v in $1; do cp tools/ab/libfsc_$v.so freesound_classification_amd/libfsc_hip.so
echo "== $v (round $r)"; python tools/l16_check.py --iters 20 ${2:-b0c2 b1e b1c2 b2e b2c2 b3c2 b4c2} 2>/dev/null | awk '{print $1, $2, $3, $(NF-6), $(NF-5), $(NF-4), $(NF-3)}'
done; done
