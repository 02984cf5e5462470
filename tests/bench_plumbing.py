"""Test-only launcher of bench.py's N > 1 code path on a ONE-GPU box (tests/test_gpu_bloom.py::test_bench_two_rank_code_path_executes).
bench.py itself has no switches that shrink its model or change its backend; this wrapper patches the module from outside: both
ranks share cuda:0, the process group is gloo (RCCL refuses two ranks on one device), the model is shrunk to (layers, vocab) from
argv[1], and the JSON line is re-labelled so it can never be mistaken for a measurement.
Usage (under torch.distributed.run): tests/bench_plumbing.py LAYERS,VOCAB [bench.py arguments...]"""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

layers, vocab = (int(x) for x in sys.argv[1].split(","))
os.environ["LOCAL_RANK"] = "0"                                   # every rank on cuda:0
import torch.distributed as dist

import bench

bench.L, bench.V = layers, vocab
_init = dist.init_process_group
dist.init_process_group = lambda backend=None, *a, **k: _init("gloo", *a, **k)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(sys.argv[2:])
for line in buf.getvalue().splitlines():
    if line.startswith("{"):
        doc = json.loads(line)
        doc["metric"] = "PLUMBING RUN (not a measurement): " + doc["metric"]
        doc["config"]["plumbing_override"] = {"layers": layers, "vocab": vocab, "one_device": True, "backend": "gloo"}
        line = json.dumps(doc)
    print(line, flush=True)
