set -u
V=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_bar2.so
MI355TTS_LIB=$V timeout 900 python -m pytest tests/test_gpu_f5.py -m gpu -q -x --timeout 600 -k "adaln and f32 or fixture and fp16-pairs or coexist" 2>&1 | tail -3
mkdir -p /tmp/d0 /tmp/d1
B="python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3"
for i in 1 2; do
 $B --dump-dir /tmp/d0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', round(d['ms_per_step'],2), [(k['kernel'][:30], round(k['avg_launch_us'],1)) for k in d['roofline']['kernels'][:2]])"
 MI355TTS_LIB=$V $B --dump-dir /tmp/d1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bar2', round(d['ms_per_step'],2), [(k['kernel'][:30], round(k['avg_launch_us'],1)) for k in d['roofline']['kernels'][:2]])"
done
python -c "
import numpy as np, glob
a=np.load(glob.glob('/tmp/d0/*.npy')[0]); b=np.load(glob.glob('/tmp/d1/*.npy')[0]); print('bit-identical waveforms:', np.array_equal(a,b), np.abs(a.astype(int)-b.astype(int)).max())"
