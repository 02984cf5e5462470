import sys, os, torch
sys.path.insert(0, os.getcwd())
from cleantransformer_amd import ops
from cleantransformer_amd.models.modeling_bloom import alibi_slopes
from torch.profiler import profile, ProfilerActivity
DEV="cuda:0"
B,S,nh,hd=4,2048,32,128
H=nh*hd; T=B*S
qkv=(torch.randn(T,3*H,device=DEV)*0.5).bfloat16(); go=(torch.randn(T,H,device=DEV)*0.5).bfloat16()
out=torch.empty((T,H),dtype=torch.bfloat16,device=DEV)
mask=ops.MaskInfo(torch.ones(B,S,dtype=torch.long,device=DEV)); slopes=alibi_slopes(nh).to(DEV)
desc=ops.fused_qkv_desc(B,S,nh,hd,True)
sm,sl=ops.attn_fwd(qkv,qkv[:,hd:],qkv[:,2*hd:],out,desc,slopes,mask)
dq=torch.empty_like(qkv)
def bwd(): ops.attn_bwd(qkv,qkv[:,hd:],qkv[:,2*hd:],out,go,sm,sl,dq,dq[:,hd:],dq[:,2*hd:],desc,slopes,mask)
for _ in range(2): bwd(); ops.attn_fwd(qkv,qkv[:,hd:],qkv[:,2*hd:],out,desc,slopes,mask)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        ops.attn_fwd(qkv,qkv[:,hd:],qkv[:,2*hd:],out,desc,slopes,mask); bwd()
    torch.cuda.synchronize()
for e in prof.key_averages():
    if "attn" in e.key: print(f"   {e.key[:60]:60s} {e.device_time:9.1f} us avg")
