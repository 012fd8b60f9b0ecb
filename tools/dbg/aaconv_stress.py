import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/text-to-speech-tts-onnx_amd')
import numpy as np
from mi355tts import weights as W
from mi355tts import bigvgan as BV
def run(C,k,d,B,T,dtype,n=6,res=True):
    x=W.synth_normal(1,"x",(B,C,T)); w=W.synth_normal(2,"w",(C,C,k),std=1/np.sqrt(C*k)); b=W.synth_normal(3,"b",(C,),std=0.1)
    al=W.synth_normal(4,"a",(C,),std=0.1); be=W.synth_normal(5,"be",(C,),std=0.1)
    r=W.synth_normal(6,"r",(B,C,T)) if res else None
    outs=[BV.aa_conv1d(x,al,be,w,b,dilation=d,res=r,dtype=dtype) for _ in range(n)]
    bad=[]
    for i,o in enumerate(outs[1:],1):
        if not np.array_equal(o,outs[0]):
            dd=np.abs(o-outs[0]); idx=np.argwhere(dd>0)
            bad.append((i,float(dd.max()),len(idx),idx[0].tolist(),idx[-1].tolist()))
    print(f"C={C} k={k} d={d} B={B} T={T} {dtype}: mismatching {len(bad)}/{n-1}",bad[:3],flush=True)
for dtype in ("bf16","f16","f32"):
    for (C,k,d) in [(48,11,5),(48,11,3),(48,3,1),(96,11,5),(96,3,1),(24,11,5)]:
        run(C,k,d,8,65536 if C<=48 else 32768,dtype)
