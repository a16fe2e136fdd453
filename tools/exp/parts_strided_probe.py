import sys, torch
sys.path.insert(0, "/root/repo")
from sglang_npu_amd import ops
dev="cuda:0"; g=torch.Generator(device=dev).manual_seed(0)
Hq,Hkv,D,L,P=32,8,128,128,4096
n_tok=16*6144+1
kb=torch.randn(n_tok,Hkv,D,device=dev,generator=g).bfloat16(); vb=torch.randn(n_tok,Hkv,D,device=dev,generator=g).bfloat16()
perm=(torch.randperm(n_tok-1,device=dev,generator=g)+1).to(torch.int32)
qkv=torch.randn(L,(Hq+2*Hkv)*D,device=dev,generator=g).bfloat16()
q=qkv[:,:Hq*D].view(L,Hq,D); ke=qkv[:,Hq*D:(Hq+Hkv)*D].view(L,Hkv,D); ve=qkv[:,(Hq+Hkv)*D:].view(L,Hkv,D)
o=torch.zeros(L,Hq,D,dtype=torch.bfloat16,device=dev)
qo=torch.tensor([0,L],dtype=torch.int32,device=dev); kvp=torch.tensor([0,P],dtype=torch.int32,device=dev)
for name,idx in (("random page table",perm[:P].contiguous()),("contiguous slots",torch.arange(1,P+1,device=dev,dtype=torch.int32))):
    sc=ops.ExtendPartsScratch(dev)
    for label,kw in (("plain",{}),("parts",dict(max_prefix_len=P,parts_scratch=sc))):
        f=lambda: ops.extend_attention_fwd(q,ke,ve,o,kb,vb,qo,kvp,idx,None,True,None,L,D**-0.5,0.0,-1,**kw)
        for _ in range(3): f()
        torch.cuda.synchronize()
        ts=[]
        for _ in range(20):
            st,en=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            st.record(); f(); en.record(); torch.cuda.synchronize(); ts.append(st.elapsed_time(en)*1e3)
        ts.sort(); print(name,label,"eager single launch us",round(ts[len(ts)//2],1))
