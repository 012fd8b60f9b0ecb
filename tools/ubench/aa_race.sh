#!/bin/bash
# build + run the aa_conv determinism stress on the GPU box (see aa_race.hip)
cd "$(dirname "$0")/../.."
CS=text-to-speech-tts-onnx_amd/csrc
mkdir -p gpurun_out/aa_race
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNDEBUG -I $CS tools/ubench/aa_race.hip $CS/aa_conv.hip $CS/aa_act.hip $CS/runtime.hip -o gpurun_out/aa_race/aa_race 2> gpurun_out/aa_race/build.log || { echo "build failed"; exit 1; }
for c in 64 48; do
  echo "== C=$c, two workgroups per CU (MI355TTS_AACONV_LDS_MIN=0)"; MI355TTS_AACONV_LDS_MIN=0 gpurun_out/aa_race/aa_race $c $((c==64?32768:65536)) ${REPS:-120}
  echo "== C=$c, one workgroup per CU (default)"; gpurun_out/aa_race/aa_race $c $((c==64?32768:65536)) ${REPS:-120}
done
rm -f gpurun_out/aa_race/aa_race
