# same-box A/B of the L16 forward kernel: tools/l16_check.py with each library given (development tool)
for v in "$@"; do cp tools/ab/libfsc_$v.so freesound_classification_amd/libfsc_hip.so; echo "== $v"
timeout 300 python tools/l16_check.py b0c2 b1e b1c2 b2c2 b3c2 b0c1 2>&1 | grep -E "fwd |dgrad" | sed -E "s/roundtrip.*\| new/| new/"; done
