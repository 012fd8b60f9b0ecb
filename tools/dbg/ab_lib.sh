#!/bin/bash
# same-box A/B of the in-tree library against a variant build (build.py --variant NAME): tools/dbg/ab_lib.sh NAME [bench args]
# also says whether the two builds produce bit-identical waveforms (kernel variants that only reschedule must)
V=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_$1.so; shift
mkdir -p /tmp/ab_d0 /tmp/ab_d1
B="python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3 $*"
S='import json,sys; d=json.load(open("bench_detail.json")); print(sys.argv[1], round(d["ms_per_step"],2), [(k["kernel"][-28:], round(k["avg_launch_us"],1)) for k in (d["roofline"].get("instantiations") or d["roofline"]["kernels"])[:4]], [(k["kernel"][:12], round(k["avg_launch_us"],1)) for k in d["roofline"]["kernels"] if "attn" in k["kernel"]])'
for i in 1 2 3; do $B --dump-dir /tmp/ab_d0 > /dev/null 2>&1; python -c "$S" base; MI355TTS_LIB=$V $B --dump-dir /tmp/ab_d1 > /dev/null 2>&1; python -c "$S" variant; done
python -c "
import numpy as np, glob
a=np.load(sorted(glob.glob('/tmp/ab_d0/*.npy'))[0]); b=np.load(sorted(glob.glob('/tmp/ab_d1/*.npy'))[0]); print('bit-identical waveforms:', np.array_equal(a,b), 'max |diff|', int(np.abs(a.astype(int)-b.astype(int)).max()))"
