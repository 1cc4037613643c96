for r in 1 2; do for v in $1; do cp tools/ab/libfsc_$v.so freesound_classification_amd/libfsc_hip.so; echo "== $v"; python tools/frontend_bench.py 2>/dev/null | head -1; done; done
