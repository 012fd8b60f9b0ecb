import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/text-to-speech-tts-onnx_amd')
import numpy as np
from mi355tts.config import BigVGANConfig
from mi355tts import weights as W, _lib
from mi355tts import bigvgan as BV
cfg=BigVGANConfig(); st=W.synth_state(W.bigvgan_spec(cfg),9527)
mel8=W.bigvgan_synthetic_mel(cfg,8,512,0)
def run(tag, env=None, opts=None, dtype="bf16", n=8, B=8, F=512, floatout=False):
    for k in ("MI355TTS_NO_FUSED_AA","MI355TTS_FUSED_MAX_C"): os.environ.pop(k,None)
    for k,v in (env or {}).items(): os.environ[k]=v
    base={"gemm_use_dma3":1,"gemm_use_dma":1,"gemm_big_tiles":1,"gemm_n192":1,"gemm_buf":1}
    base.update(opts or {})
    for k,v in base.items(): _lib.set_option(k,v)
    v=BV.BigVGANVocoder(cfg,st,dtype=dtype)
    m=np.ascontiguousarray(mel8[:B,:,:F])
    outs=[v.run_float(m) if floatout else v.run(m) for _ in range(n)]
    ref=outs[0]; bad=[]
    for i,o in enumerate(outs[1:],1):
        if not np.array_equal(o,ref):
            d=np.abs(o.astype(np.float64)-ref.astype(np.float64)); idx=np.argwhere(d>0)
            bad.append((i,float(d.max()),len(idx),idx[0].tolist(),idx[-1].tolist()))
    print(tag,"mismatching reruns:",len(bad),bad[:4],flush=True)
    v.close()
run("bf16 default")
run("bf16 unfused",env={"MI355TTS_NO_FUSED_AA":"1"})
run("bf16 fused<=24",env={"MI355TTS_FUSED_MAX_C":"24"})
run("bf16 fused<=48",env={"MI355TTS_FUSED_MAX_C":"48"})
run("bf16 no_dma",opts={"gemm_use_dma":0,"gemm_use_dma3":0})
run("bf16 no_dma unfused",env={"MI355TTS_NO_FUSED_AA":"1"},opts={"gemm_use_dma":0,"gemm_use_dma3":0})
run("bf16 no_dma3",opts={"gemm_use_dma3":0})
run("bf16 no_buf",opts={"gemm_buf":0})
run("f16 default",dtype="f16")
run("f32 default B=2",dtype="f32",B=2,n=5)
run("bf16 B=1",B=1)
run("bf16 F=64",F=64)
