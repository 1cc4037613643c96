# same-box A/B of the L16 weight-gradient kernel: tools/l16_check.py --wgrad with each library given (development tool)
for v in "$@"; do cp tools/ab/libfsc_$v.so freesound_classification_amd/libfsc_hip.so; echo "== $v"
timeout 300 python tools/l16_check.py --wgrad b0c1 b0c2 b1e b1c2 b2e b2c2 b3c2 b4c2 2>&1 | grep wgrad | sed -E "s/\| old.*\| new/| new/"; done
