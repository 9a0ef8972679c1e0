#!/usr/bin/env python3
"""Randomised cross-check of yume_attn_fwd's automatic kernel selection (8-wave kernel, query split, key-range split of the tail)
against the 4-wave kernel on 60 shapes. Not a test file: prints mismatches."""
import torch, sys, random
sys.path.insert(0, ".")
from yume_amd import ops
random.seed(1)
DEV="cuda"
bad=0
shapes=[]
for _ in range(60):
    H=random.choice([1,2,3,5,8,9,16,24])
    Lk=random.choice([1536,1537,1600,2047,2048,2111,3000,4096-7])
    Lq=random.choice([1,31,255,256,257,511,513,1000,2049,4097,8192+13, 8500, 9460])
    if Lq*Lk*H > 9460*4096*8: Lq = min(Lq, 2049)
    shapes.append((Lq,Lk,H))
for (Lq,Lk,H) in shapes:
    q=(torch.randn(Lq,H*128,device=DEV)*0.7).bfloat16(); k=(torch.randn(Lk,H*128,device=DEV)*0.7).bfloat16()
    Lp=(Lk+7)//8*8
    vt=torch.full((H*128,Lp),float("nan"),device=DEV,dtype=torch.bfloat16); vt[:,:Lk]=(torch.randn(H*128,Lk,device=DEV)).bfloat16()
    outs={}
    for var,ws in ((2,False),(0,True),(0,False),(4,False)):
        if var==4 and Lq<1: continue
        o=torch.full((Lq,H*128),7.0,device=DEV,dtype=torch.bfloat16)
        ops.attn_fwd(q,k,vt,o,Lq,Lk,H,variant=var,use_workspace=ws)
        outs[(var,ws)]=o.float()
    ref=outs[(2,False)]
    for key,o in outs.items():
        if not torch.isfinite(o).all(): print("NONFINITE",(Lq,Lk,H),key); bad+=1; continue
        e=((o-ref).norm()/ref.norm()).item(); m=(o-ref).abs().max().item()
        if e>4e-3 or m>0.05: print("MISMATCH",(Lq,Lk,H),key,e,m); bad+=1
print("shapes",len(shapes),"bad",bad)
