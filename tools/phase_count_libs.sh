#!/bin/bash
# Builds the profiling variants of the library that leave phases of the persistent decode kernel out (JDA_EXP_SKIP, see
# jda_kernels.hip): ab/lib_skip{1,3,7,15,31}.so = no P4 / + no P3 / + no P2 / + no lists / + no P1.  Run HERE (hipcc cross-
# compiles); then on the GPU box: tools/gpu_counts2.sh jpegdec_amd/libjpegdec_amd.so ab/lib_skip*.so  (differences of
# consecutive SQ_INSTS_VALU figures = the phases' instruction counts per tile; 4:4:4: tools/gpu_counts444.sh).
set -e
mkdir -p ab
cp jpegdec_amd/libjpegdec_amd.so ab/lib_full.so
for k in 1 3 7 15 31; do
  touch jpegdec_amd/csrc/jda_kernels.hip
  make lib EXTRA="-DJDA_EXP_SKIP=$k" > /dev/null
  cp jpegdec_amd/libjpegdec_amd.so ab/lib_skip$k.so
done
touch jpegdec_amd/csrc/jda_kernels.hip
make lib > /dev/null
cmp jpegdec_amd/libjpegdec_amd.so ab/lib_full.so && echo "product library restored"
