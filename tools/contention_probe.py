"""How sensitive is the SFT step to a few CUs being held by somebody else?  (Stand-in for an RCCL all-reduce kernel running
under backward on a multi-GPU job: K single-wave spin kernels on K side streams each squat on a CU for the whole timed
region.  A workgroup that needs a whole CU — the ping-pong GEMM tiles: all LDS, all VGPRs — cannot share a CU with one.)

usage: python tools/contention_probe.py [K ...]      env: CTMI_GEMM_* as usual
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ks = [int(a) for a in sys.argv[1:]] or [0, 8, 32]
    dev = torch.device("cuda:0")
    from cleantransformer_amd.optimizer import AdamW
    model = bench.build_model(dev, "bf16")
    opt = AdamW(model.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True)
    ids = torch.randint(0, bench.V, (8, 1024), device=dev)
    am = torch.ones(8, 1024, dtype=torch.long, device=dev)

    def step():
        outputs, _ = model(input_ids=ids, attention_mask=am, labels=ids)
        opt.zero_grad()
        outputs[0].backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(max(ks))]
    for k in ks:
        torch.cuda.synchronize()
        for s in streams[:k]:
            with torch.cuda.stream(s):
                torch.cuda._sleep(int(2.0e9))                    # ~1 s of spinning: covers the timed region
        time.sleep(0.05)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(6):
            step()
        ev1.record()
        ev1.synchronize()
        print(f"squatters={k:3d}: {ev0.elapsed_time(ev1) / 6:8.3f} ms/step", flush=True)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
