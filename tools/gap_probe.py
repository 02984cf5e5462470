import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
from cleantransformer_amd.optimizer import AdamW
m = bench.build_model(dev, "bf16")
opt = AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
ids = torch.randint(0, bench.V, (8, 1024), device=dev); am = torch.ones(8, 1024, dtype=torch.long, device=dev)
def step(ev=None):
    outputs, _ = m(input_ids=ids, attention_mask=am, labels=ids)
    opt.zero_grad()
    outputs[0].backward()
    if ev: ev[0].record()
    t0 = time.perf_counter()
    opt.step()
    t1 = time.perf_counter()
    if ev: ev[1].record()
    return t1 - t0
for _ in range(3): step()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
host = []
t0 = time.perf_counter()
for e in evs: host.append(step(e))
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 8 * 1e3)
print("GPU time between end of backward and end of optimizer (ms):", [round(a.elapsed_time(b), 3) for a, b in evs])
print("host time inside opt.step() (ms):", [round(h * 1e3, 3) for h in host])
