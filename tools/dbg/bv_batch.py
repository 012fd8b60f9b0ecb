import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/text-to-speech-tts-onnx_amd')
import numpy as np
from mi355tts.config import BigVGANConfig
from mi355tts import weights as W, _lib
from mi355tts import bigvgan as BV
cfg=BigVGANConfig(); st=W.synth_state(W.bigvgan_spec(cfg),9527)
mel8=W.bigvgan_synthetic_mel(cfg,8,512,0)
def run(tag, opts, F=512, B=8, dtype="f16"):
    for k,v in opts.items(): _lib.set_option(k,v)
    v=BV.BigVGANVocoder(cfg,st,dtype=dtype)
    m=np.repeat(mel8[:1,:,:F],B,axis=0)
    w=v.run(m)
    w2=v.run(m)
    out=[]
    for b in range(1,B):
        d=np.abs(w[b,0].astype(int)-w[0,0].astype(int))
        nz=np.nonzero(d)[0]
        out.append((b,int(d.max()),len(nz),(int(nz.min()),int(nz.max())) if len(nz) else None))
    print(tag, "rerun_equal", np.array_equal(w,w2), out, flush=True)
    v.close()
base={"gemm_use_dma3":1,"gemm_use_dma":1,"gemm_big_tiles":1,"gemm_n192":1,"gemm_buf":1}
run("default",base)
run("default F=64 B=8",base,F=64)
run("default F=512 B=2",base,B=2)
run("no_buf",{**base,"gemm_buf":0})
run("no_dma3",{**base,"gemm_use_dma3":0})
run("no_big",{**base,"gemm_big_tiles":0})
run("no_n192",{**base,"gemm_n192":0})
run("no_dma",{**base,"gemm_use_dma":0,"gemm_use_dma3":0})
run("bf16",base,dtype="bf16")
