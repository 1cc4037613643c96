# development: a variant of conv_l16.hip (cfg-2 instantiations only) linked with the objects of the current build
#   tools/ab/build_variant.sh NAME "-DFSC_L16_DEV -DFSC_L16_EARLY_A=1 ..." [SOURCE]  ->  tools/ab/libfsc_NAME.so   (objects under csrc/build/var_*)
set -e
NAME=$1; FLAGS=$2; SRC=${3:-conv_l16}
cd freesound_classification_amd/csrc; mkdir -p build/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -Wno-inline-asm -Wno-unused-value $FLAGS -c $SRC.hip -o build/var/${SRC}_$NAME.o
OBJS=""; for o in frontend conv conv_s1d conv_l16 conv_l3 conv_l16_wgrad stem_pooled norm_act pool dense loss optim misc rnn; do
  if [ $o = $SRC ]; then OBJS="$OBJS build/var/${SRC}_$NAME.o"; else OBJS="$OBJS build/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab/libfsc_$NAME.so $OBJS
