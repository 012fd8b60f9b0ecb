#!/bin/bash
# Determinism A/B of the fused AA+conv kernel on the GPU box (see aa_race.hip, aa_r2/, opsel_coexec.hip):
#   r2        round-2 sources as shipped (one sample per channel, xv[1], is read as `v_pk_fma_f32 ... op_sel:[0,1,0]`)
#   r2_hi20   the same sources with sample xv[20] ALSO read through that encoding (AA_R2_FORCE_HI=20): if the encoding is the
#             cause, differing rows appear at t%16 in 10..15 in addition to 0..1
#   r3        the product sources (channel pairs: no VGPR operand is broadcast)
# each at two workgroups per CU (MI355TTS_AACONV_LDS_MIN=0) and, for r2, one workgroup per CU as the control.
cd "$(dirname "$0")/../.."
CS=text-to-speech-tts-onnx_amd/csrc
# usage: aa_race.sh build   (here: hipcc cross-compiles, binaries land in tools/ubench/bin/, git-ignored, shipped by gpurun)
#        aa_race.sh run     (GPU box)
O=tools/ubench/bin
mkdir -p $O gpurun_out/aa_race
if [ "$1" != "run" ]; then
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNDEBUG -w"
$HC -I tools/ubench/aa_r2 -I $CS tools/ubench/aa_race.hip tools/ubench/aa_r2/aa_conv.hip $CS/aa_act.hip $CS/runtime.hip -o $O/aa_race_r2 2> gpurun_out/aa_race/build_r2.log || { echo "build r2 failed"; cat gpurun_out/aa_race/build_r2.log; exit 1; }
$HC -DAA_R2_FORCE_HI=20 -I tools/ubench/aa_r2 -I $CS tools/ubench/aa_race.hip tools/ubench/aa_r2/aa_conv.hip $CS/aa_act.hip $CS/runtime.hip -o $O/aa_race_r2_hi20 2> gpurun_out/aa_race/build_r2hi.log || { echo "build r2_hi20 failed"; cat gpurun_out/aa_race/build_r2hi.log; exit 1; }
$HC -I $CS tools/ubench/aa_race.hip $CS/aa_conv.hip $CS/aa_act.hip $CS/runtime.hip -o $O/aa_race_r3 2> gpurun_out/aa_race/build_r3.log || { echo "build r3 failed"; cat gpurun_out/aa_race/build_r3.log; exit 1; }
$HC tools/ubench/opsel_coexec.hip -o $O/opsel 2> gpurun_out/aa_race/build_opsel.log || { echo "build opsel failed"; cat gpurun_out/aa_race/build_opsel.log; exit 1; }
[ "$1" = "build" ] && exit 0
fi
REPS=${REPS:-120}
for c in 64 48 96 24; do
  case $c in 96) T=32768;; 64) T=32768;; 48) T=65536;; 24) T=131072;; esac
  for v in r2 r2_hi20 r3; do
    echo "== $v C=$c, two workgroups per CU (MI355TTS_AACONV_LDS_MIN=0)"; MI355TTS_AACONV_LDS_MIN=0 timeout 300 $O/aa_race_$v $c $T $REPS
  done
  echo "== r2 C=$c, ONE workgroup per CU (MI355TTS_AACONV_LDS_MIN=83968)"; MI355TTS_AACONV_LDS_MIN=83968 timeout 300 $O/aa_race_r2 $c $T $REPS
done
echo "== opsel_coexec (standalone)"; timeout 300 $O/opsel
