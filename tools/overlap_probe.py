"""Can the HBM-bound optimizer pass hide under MFMA-bound GEMMs on disjoint CUs?  (AdamW of step t under the forward of step t+1.)
GEMM sequence = the forward GEMMs of the Bloom-560M step (24 x qkv/dense/h4h/4hh + LM head) on the current stream with R CUs left
out of every persistent launch (ctmi_set_launch_policy reserve), AdamW over 559 M parameters on a side stream.  Prints the two
alone, back to back, and concurrent for each R."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleantransformer_amd import ops  # noqa: E402

DEV, BF = "cuda:0", torch.bfloat16
T, H, V = 8192, 1024, 250880


def rnd(*s):
    return (torch.randn(*s, device=DEV) * 0.05).to(BF)


def main():
    x, x4 = rnd(T, H), rnd(T, 4 * H)
    wq, wd, w1, w2, wv = rnd(3 * H, H), rnd(H, H), rnd(4 * H, H), rnd(H, 4 * H), rnd(V, H)
    n = 559_214_592
    p, g = torch.zeros(n, device=DEV), torch.ones(n, device=DEV) * 1e-3
    m, v, sh = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.empty(n, dtype=BF, device=DEV)
    kw = dict(lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3, decoupled=True)
    # 13 chunks like the optimizer's packs, so the side stream holds several kernels
    ch = [(i * n // 13, (i + 1) * n // 13) for i in range(13)]

    def gemms():
        for _ in range(24):
            ops.linear_fwd(x, wq, None); ops.linear_fwd(x, wd, None); ops.linear_fwd(x, w1, None); ops.linear_fwd(x4, w2, None)
        ops.linear_fwd(x, wv, None)

    def adam():
        for a, b in ch:
            ops.adamw_step([p[a:b]], [g[a:b]], [m[a:b]], [v[a:b]], [sh[a:b]], **kw)

    side = torch.cuda.Stream()

    def timed(fn, it=3):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it

    def both():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            adam()
        gemms()
        torch.cuda.current_stream().wait_stream(side)

    def serial():
        adam(); gemms()

    for R in [int(a) for a in sys.argv[1:]] or [0, 16, 32, 48]:
        ops.set_launch_policy(False, R)
        tg, ta, ts, tb = timed(gemms), timed(adam), timed(serial), timed(both)
        print(f"reserve {R:3d} CUs: gemms {tg:7.3f}  adamw {ta:7.3f}  back-to-back {ts:7.3f}  concurrent {tb:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
