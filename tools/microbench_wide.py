import sys, os, torch
sys.path.insert(0, os.getcwd())
from cleantransformer_amd import ops
DEV="cuda:0"
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b)/it
for T,H in ((8192,4096),(8192,2048),(8192,1024)):
    x=(torch.randn(T,H,device=DEV)*0.5).bfloat16(); g=(torch.randn(T,H,device=DEV)*0.5).bfloat16()
    w=torch.ones(H,device=DEV); b=torch.zeros(H,device=DEV)
    y,mean,rstd=ops.layernorm_fwd(x,w,b,1e-5)
    t=timeit(lambda: ops.layernorm_fwd(x,w,b,1e-5)); print(f"[{T},{H}] fwd {t*1e3:8.1f} us {2*T*H*2/t/1e6:8.1f} GB/s")
    t=timeit(lambda: ops.layernorm_bwd(g,x,w,mean,rstd,dres=g)); print(f"[{T},{H}] bwd {t*1e3:8.1f} us {4*T*H*2/t/1e6:8.1f} GB/s")
    for N in (H, 3*H, 4*H):
        gg=(torch.randn(T,N,device=DEV)).bfloat16()
        t=timeit(lambda: ops.colsum(gg)); print(f"colsum [{T},{N}] {t*1e3:8.1f} us {T*N*2/t/1e6:8.1f} GB/s")
