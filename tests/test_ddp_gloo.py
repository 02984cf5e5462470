"""CPU, world_size 2 and 4 over gloo: the DistributedDataParallel wrapper (trainer/ddp.py) around the product Bloom model
(kernel contracts emulated on CPU) reproduces torch-DDP's result on the reference model: rank-0 broadcast at wrap time,
bucketed all-reduce from autograd hooks, gradients AVERAGED over ranks.  Golden: tests/golden/ddp_tiny.npz, produced by
torch.nn.parallel.DistributedDataParallel(gloo) around the *reference* model (tests/golden/make_golden.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")


class _Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bucket_mb, ret):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_kernel_emulation as emu
    from oracle import bloom_ref as R
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
    emu.install(_Patch())
    V, H, L, nh, B, S = 211, 64, 2, 8, 2, 16
    m = BloomForCausalLM(BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh))
    m._tie_weight()
    sd = dict(R.det_init(R.BloomShape(V, H, L, nh)))
    sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
    m.load_state_dict(sd)
    m._tie_weight()
    if rank != 0:                                              # wrong weights on the other ranks: the wrapper must broadcast rank 0's
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
    ddp = DDP(m, device_ids=None, bucket_cap_mb=bucket_mb)
    assert list(ddp.state_dict().keys())[0] == "module.bloom.word_embeddings.weight"
    ids = torch.randint(0, V, (world * B, S), generator=torch.Generator().manual_seed(7))[rank * B:(rank + 1) * B]
    am = torch.ones(B, S, dtype=torch.long)
    if rank == 1:
        am[0, 11:] = 0
    ddp.train()
    (loss, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
    loss.backward()
    # every rank must hold the same averaged gradient
    chk = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(torch.equal(lo, hi))
    # no_sync(): local gradients only
    for p in m.parameters():
        p.grad = None
    with ddp.no_sync():
        (l2, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
        l2.backward()
    local = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    lo2, hi2 = local.clone(), local.clone()
    dist.all_reduce(lo2, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi2, op=dist.ReduceOp.MAX)
    # gradient accumulation: a synced backward on top of the local gradients reduces the SUM of both passes (torch-DDP
    # semantics); the tied [V,H] gradient must take its bucket fallback here (its early path needs .grad to be None)
    early_before = ddp._tied_sync.steps
    (l4, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
    l4.backward()
    accum = {n: p.grad.numpy().copy() for n, p in m.named_parameters()}
    early_in_accum = ddp._tied_sync.steps - early_before
    if rank == 0:
        ret["same_on_all_ranks"] = same
        ret["no_sync_differs"] = not bool(torch.equal(lo2, hi2))
        ret["loss0"] = float(loss)
        ret["buckets"] = ddp.bucket_summary()
        ret["early_in_accum"] = early_in_accum
        for n, a in accum.items():
            ret["acc_" + n] = a
    for p in m.parameters():
        p.grad = None
    (l3, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
    l3.backward()
    if rank == 0:
        for n, p in m.named_parameters():
            ret["g_" + n] = p.grad.numpy().copy()
        ret["early_steps"] = ddp._tied_sync.steps
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bucket_mb", [(2, 25), (2, 0.05), (4, 0.2), (2, -64)])
def test_ddp_matches_torch_ddp_on_reference_model(world, bucket_mb, monkeypatch):
    """bucket_mb < 0: the dense part of the tied [V,H] gradient in row pieces of -bucket_mb rows (V = 211: 64 + 64 + 64 + 19), each
    piece produced by its own weight-gradient GEMM over a column window of dlogits and handed to its own all-reduce."""
    if bucket_mb < 0:
        monkeypatch.setenv("CTMI_DDP_TIED_CHUNK_ROWS", str(-bucket_mb))
        bucket_mb = 25
    gold = np.load(os.path.join(G, "ddp_tiny.npz"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), bucket_mb, ret), nprocs=world, join=True)
    assert ret["same_on_all_ranks"] and ret["no_sync_differs"]
    assert abs(ret["loss0"] - float(gold[f"w{world}___loss0"])) < 1e-5
    nb = len(ret["buckets"])
    assert nb >= (1 if bucket_mb >= 25 else 3), ret["buckets"]
    for k in gold.files:
        if k.startswith(f"w{world}_bloom") or k.startswith(f"w{world}_lm_head"):
            name = k[len(f"w{world}_"):]
            a, b = ret["g_" + name], gold[k]
            assert a.shape == b.shape
            assert np.allclose(a, b, rtol=1e-4, atol=1e-8), (name, float(np.abs(a - b).max()))
            acc = ret["acc_" + name]                              # no_sync pass + synced pass on the same batch = 2 x the average
            assert np.allclose(acc, 2.0 * b, rtol=1e-4, atol=2e-8), ("accumulated", name, float(np.abs(acc - 2.0 * b).max()))
    # the tied embedding / LM-head gradient went through the early dense all-reduce + row exchange in every plain synced
    # backward (3 of them), and through its bucket in the accumulation step
    assert ret["early_steps"] == 2 and ret["early_in_accum"] == 0, (ret["early_steps"], ret["early_in_accum"])


def _trainer_worker(rank, world, port, out_dir, ret):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_kernel_emulation as emu
    emu.install(_Patch())
    from test_host_logic_cpu import TINY, T, build
    from cleantransformer_amd.trainer import Trainer, TrainingArguments
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    ids, am = T(TINY["ids"]), T(TINY["mask"])
    rows = slice(rank * 2, rank * 2 + 2)                         # each rank: half of the golden batch, as TWO micro-batches of one row

    class Data:
        def __iter__(self):
            for r in range(rows.start, rows.stop):
                yield {"input_ids": ids[r:r + 1], "attention_mask": am[r:r + 1], "labels": ids[r:r + 1].clone()}

        def __len__(self):
            return 2
    args = TrainingArguments(output_dir=out_dir, device="cpu", max_steps=1, gradient_accumulation_steps=2, per_device_train_batch_size=1,
                             learning_rate=1e-5, weight_decay=0.01, lr_scheduler_type="constant", max_grad_norm=1e9, logging_steps=1, save_steps=1)
    tr = Trainer(model=build(V, H, L, nh), args=args, train_dataset=Data())
    tr.train()
    if rank == 0:
        ret["log"] = dict(tr.state.log_history[0])
        ret["files"] = sorted(os.listdir(os.path.join(out_dir, "checkpoint-1")))
        ret["wrapped"] = type(tr.model_wrapped).__name__
        for n, p in tr.model.named_parameters():
            ret["p_" + n] = p.detach().numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_under_ddp_accumulation_equals_the_golden_full_batch_step(tmp_path):
    """world 2 x gradient_accumulation 2 x micro-batch 1 = the golden batch of 4 rows: one all-reduce per optimizer step
    (micro-step 1 runs under no_sync), averaged loss / grad-norm in the log, rank-0-only checkpoint, parameters after the step
    equal to the single-process full-batch step."""
    from test_host_logic_cpu import TINY
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_trainer_worker, args=(2, _free_port(), str(tmp_path), ret), nprocs=2, join=True)
    assert ret["wrapped"] == "DistributedDataParallel"
    assert abs(ret["log"]["loss"] - float(TINY["traj"][0, 0])) <= 1.01e-4
    assert abs(ret["log"]["grad_norm"] - float(TINY["traj"][0, 1])) <= 1e-4 * float(TINY["traj"][0, 1])
    assert ret["files"] == sorted(["pytorch_model.bin", "optimizer.pt", "scheduler.pt", "trainer_state.json", "rng_state_0.pth",
                                   "training_args.bin"])
    import cpu_kernel_emulation as emu
    from test_host_logic_cpu import T, build
    from cleantransformer_amd.examples.ft_bloom import train_step
    from cleantransformer_amd.optimizer import AdamW
    emu.install(_Patch())                                          # (process-wide; this module's other tests spawn their own workers)
    V, H, L, nh, B, S = [int(v) for v in TINY["cfg"]]
    m = build(V, H, L, nh)
    train_step(m, {"input_ids": T(TINY["ids"]), "attention_mask": T(TINY["mask"]), "labels": T(TINY["ids"]).clone()},
               AdamW(m.parameters(), lr=1e-5, weight_decay=0.01, decoupled=True))
    for n, p in m.named_parameters():
        assert np.allclose(ret["p_" + n], p.detach().numpy(), rtol=1e-6, atol=5e-8), n


def _bf16_comm_worker(rank, world, port, ret):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_kernel_emulation as emu
    emu.install(_Patch())
    from test_host_logic_cpu import build
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
    V, H, L, nh, B, S = 211, 64, 2, 8, 2, 16
    m = build(V, H, L, nh)
    ddp = DDP(m, device_ids=None, bucket_cap_mb=0.05, comm_dtype=torch.bfloat16)
    ids = torch.randint(0, V, (world * B, S), generator=torch.Generator().manual_seed(7))[rank * B:(rank + 1) * B]
    am = torch.ones(B, S, dtype=torch.long)
    if rank == 1:
        am[0, 11:] = 0
    for _ in range(2):                                         # second pass: gradients are views of the fp32 buckets
        for p in m.parameters():
            p.grad = None
        (loss, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
        loss.backward()
    chk = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        ret["same"] = bool(torch.equal(lo, hi))
        ret["wire"] = [str(b.comm.dtype) for b in ddp._buckets if b.comm is not None]
        ret["dtypes"] = sorted({str(p.grad.dtype) for p in m.parameters()})
        for n, p in m.named_parameters():
            ret["g_" + n] = p.grad.numpy().copy()
    dist.destroy_process_group()


def test_ddp_bf16_bucket_communication_is_opt_in_and_within_bf16_rounding():
    """comm_dtype=torch.bfloat16 ("bf16 grads/buckets"): the buckets travel as bf16, the gradients the optimizer
    sees stay fp32 views of the fp32 buckets, every rank holds the same values, and they equal the torch-DDP golden to bf16
    rounding (NOT to the 1e-4 bar of the default fp32 path — hence opt-in).  The tied [V,H] gradient is reduced in fp32."""
    gold = np.load(os.path.join(G, "ddp_tiny.npz"))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bf16_comm_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["same"] and ret["dtypes"] == ["torch.float32"] and ret["wire"] and set(ret["wire"]) == {"torch.bfloat16"}
    worst_default_bar = 0.0
    for k in gold.files:
        if k.startswith("w2_bloom") or k.startswith("w2_lm_head"):
            name = k[len("w2_"):]
            a, b = ret["g_" + name], gold[k]
            scale = float(np.abs(b).max())
            assert np.allclose(a, b, rtol=2 ** -7, atol=2 ** -8 * scale), (name, float(np.abs(a - b).max()), scale)
            if name not in ("bloom.word_embeddings.weight", "lm_head.weight"):
                worst_default_bar = max(worst_default_bar, float(np.abs(a - b).max() / scale))
            else:                                                  # the tied gradient took the fp32 early path
                assert np.allclose(a, b, rtol=1e-4, atol=1e-8), name
    assert worst_default_bar > 1e-4                                 # i.e. the bf16 wire really was used for the bucketed gradients


def _uneven_worker(rank, world, port, ret):
    """Ranks with DIFFERENT sequence lengths (the reference's collate pads per rank), and a backward that raises mid-way."""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    import cpu_kernel_emulation as emu
    from oracle import bloom_ref as R
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
    emu.install(_Patch())
    V, H, L, nh, B = 211, 64, 2, 8, 2
    seqs = [16, 11, 7, 13][:world]                               # T = B*S differs on every rank
    m = BloomForCausalLM(BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh))
    m._tie_weight()
    sd = dict(R.det_init(R.BloomShape(V, H, L, nh)))
    sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
    m.load_state_dict(sd)
    m._tie_weight()
    plain = copy.deepcopy(m)
    plain._tie_weight()
    ddp = DDP(m, device_ids=None, bucket_cap_mb=0.05).train()
    batches = [torch.randint(0, V, (B, s), generator=torch.Generator().manual_seed(40 + r)) for r, s in enumerate(seqs)]
    # expected: the mean over ranks of each rank's local gradient, computed without the wrapper
    want = None
    for r, ids in enumerate(batches):
        for p in plain.parameters():
            p.grad = None
        (l, _, _), _ = plain(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids.clone())
        l.backward()
        g = torch.cat([p.grad.reshape(-1) for p in plain.parameters()]) / world
        want = g if want is None else want + g
    ids = batches[rank]

    def run():
        for p in m.parameters():
            p.grad = None
        (loss, _, _), _ = ddp(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids.clone())
        loss.backward()
        return torch.cat([p.grad.reshape(-1) for p in m.parameters()])

    early0 = ddp._tied_sync.steps
    got = run()
    ok_uneven = bool(torch.allclose(got, want, rtol=1e-4, atol=1e-8))
    took_early = ddp._tied_sync.steps - early0
    # a backward that raises after the first gradient hooks fired (every rank raises at the same point: the collectives that
    # were issued match up), then a normal step: the wrapper must start from a clean slate
    victim = m.bloom.blocks[0].mlp.dense_h_to_4h.weight

    def boom(g):
        raise RuntimeError("boom")
    h = victim.register_hook(boom)
    raised = False
    for p in m.parameters():
        p.grad = None
    try:
        (loss, _, _), _ = ddp(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids.clone())
        loss.backward()
    except RuntimeError as e:
        raised = "boom" in str(e)
    h.remove()
    for b in ddp._buckets:                                       # drain what the failed pass left in flight
        if b.work is not None:
            b.work.wait()
    for w in ddp._tied_sync.works:
        w.wait()
    dist.barrier()
    again = run()
    ok_after_failure = bool(torch.allclose(again, want, rtol=1e-4, atol=1e-8))
    if rank == 0:
        ret["ok_uneven"], ret["took_early"], ret["raised"], ret["ok_after_failure"] = ok_uneven, took_early, raised, ok_after_failure
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ddp_uneven_sequence_lengths_and_failed_backward_recovery(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_uneven_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["took_early"] == 1, "the tied [V,H] gradient must still take its early dense + row-exchange path"
    assert ret["ok_uneven"], "averaged gradients with rank-dependent T differ from the mean of the local gradients"
    assert ret["raised"] and ret["ok_after_failure"]


def test_tied_chunk_rows_policy(monkeypatch):
    """<= 64 MiB of fp32 per piece, whole 256-row tiles: 16384 rows at H = 1024 (one round of 256 x 256 tiles on 256 CUs), the
    whole tensor when it is smaller than a piece, overrides honoured."""
    from cleantransformer_amd.trainer.ddp import _TiedGradSync
    t = _TiedGradSync.__new__(_TiedGradSync)
    monkeypatch.delenv("CTMI_DDP_TIED_CHUNK_MB", raising=False)
    monkeypatch.delenv("CTMI_DDP_TIED_CHUNK_ROWS", raising=False)
    assert t.chunk_rows(250880, 1024) == 16384 and 16384 * 1024 * 4 == 64 * 1024 * 1024
    assert t.chunk_rows(250880, 4096) == 4096
    assert t.chunk_rows(211, 64) == 211
    monkeypatch.setenv("CTMI_DDP_TIED_CHUNK_MB", "0")
    assert t.chunk_rows(250880, 1024) == 250880
    monkeypatch.setenv("CTMI_DDP_TIED_CHUNK_ROWS", "64")
    assert t.chunk_rows(211, 64) == 64


def test_direct_rccl_backend_refuses_a_group_it_cannot_use(monkeypatch):
    """CTMI_DDP_BACKEND=rccl needs GPU parameters and an RCCL (nccl) process group to hand the communicator's id around: asked for on
    a gloo / CPU group it fails at construction with a sentence, not later inside a collective."""
    from cleantransformer_amd.trainer import ddp as D

    class _Owner:
        process_group, world_size = None, 2
        module = torch.nn.Linear(4, 4)
    monkeypatch.setattr(D.dist, "get_backend", lambda g=None: "gloo")
    monkeypatch.setenv("CTMI_DDP_BACKEND", "rccl")
    with pytest.raises(RuntimeError, match="nccl"):
        D.DistributedDataParallel._make_direct_comm(_Owner())
    monkeypatch.setenv("CTMI_DDP_BACKEND", "torch")
    assert D.DistributedDataParallel._make_direct_comm(_Owner()) is None


def _loop_order_worker(rank, world, port, ret):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_kernel_emulation as emu
    from oracle import bloom_ref as R
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    from cleantransformer_amd.optimizer import AdamW
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
    emu.install(_Patch())
    V, H, L, nh, B, S = 211, 64, 2, 8, 2, 16
    m = BloomForCausalLM(BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh))
    m._tie_weight()
    sd = dict(R.det_init(R.BloomShape(V, H, L, nh)))
    sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
    m.load_state_dict(sd)
    m._tie_weight()
    ddp = DDP(m, device_ids=None, bucket_cap_mb=0.05)
    opt = AdamW(ddp.parameters(), lr=1e-3, weight_decay=0.0, decoupled=True)
    ids = torch.randint(0, V, (world * B, S), generator=torch.Generator().manual_seed(7))[rank * B:(rank + 1) * B]
    am = torch.ones(B, S, dtype=torch.long)
    losses = []
    for _ in range(3):                                         # examples/ft_bloom.py:84-90 order: forward, zero_grad, backward, step
        (loss, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    # an evaluation forward with grad enabled on ONE rank only (no backward): must not issue any collective
    if rank == 0:
        ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
    chk = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        ret["losses"], ret["early"], ret["same"] = losses, ddp._tied_sync.steps, bool(torch.equal(lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_reference_loop_order_and_one_sided_forward():
    """The reference loop calls zero_grad() BETWEEN forward and backward (ft_bloom.py:84-90): at forward time the tied weight still
    holds last step's gradient, at backward time it does not — the early tied-gradient path is chosen, and its row capacity agreed,
    at backward time.  A grad-enabled forward that only one rank runs (rank-0 evaluation) issues no collective."""
    ret = mp.Manager().dict()
    mp.spawn(_loop_order_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["early"] == 3 and ret["same"], (ret["early"], ret["same"])
    assert ret["losses"][0] > ret["losses"][1] > ret["losses"][2], ret["losses"]


def _stub_worker(rank, world, port, ret):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_kernel_emulation as emu
    from oracle import bloom_ref as R
    from cleantransformer_amd.models.modeling_bloom import BloomConfig, BloomForCausalLM
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel as DDP
    emu.install(_Patch())
    V, H, L, nh, B, S = 211, 64, 2, 8, 2, 16
    m = BloomForCausalLM(BloomConfig(vocab_size=V, hidden_size=H, n_layer=L, num_attention_heads=nh))
    m._tie_weight()
    sd = dict(R.det_init(R.BloomShape(V, H, L, nh)))
    sd["lm_head.weight"] = sd["bloom.word_embeddings.weight"]
    m.load_state_dict(sd)
    m._tie_weight()
    ddp = DDP(m, device_ids=None, bucket_cap_mb=0.05)
    ids = torch.randint(0, V, (world * B, S), generator=torch.Generator().manual_seed(7))[rank * B:(rank + 1) * B]
    am = torch.ones(B, S, dtype=torch.long)

    def grads():
        for p in m.parameters():
            p.grad = None
        (loss, _, _), _ = ddp(input_ids=ids, attention_mask=am, labels=ids.clone())
        loss.backward()
        return {n: p.grad.clone() for n, p in m.named_parameters()}

    synced = grads()
    # stubbed: not ONE collective may be issued — every entry point of torch.distributed that moves data raises while the switch is on
    real = {k: getattr(dist, k) for k in ("all_reduce", "all_gather", "all_gather_into_tensor", "broadcast")}

    def forbidden(*a, **k):
        raise AssertionError("a collective was issued with stub_collectives = True")
    ddp.stub_collectives = True
    for k in real:
        setattr(dist, k, forbidden)
    try:
        stubbed = grads()
    finally:
        for k, f in real.items():
            setattr(dist, k, f)
        ddp.stub_collectives = False
    again = grads()
    ok_same = all(torch.equal(synced[n], again[n]) for n in synced)            # the switch leaves no state behind
    # a non-tied gradient under the stub is the LOCAL gradient / world (the pre-division rides on the bucket copy, nothing is summed)
    with ddp.no_sync():
        local = grads()
    name = "bloom.blocks.0.mlp.dense_h_to_4h.weight"
    ok_local = bool(torch.allclose(stubbed[name], local[name] / world, rtol=1e-6, atol=1e-9))
    differs = not bool(torch.allclose(stubbed[name], synced[name], rtol=1e-3, atol=1e-9))
    flags = torch.tensor([float(ok_same), float(ok_local), float(differs), float(all(bool(torch.isfinite(g).all()) for g in stubbed.values()))])
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        ret["flags"] = flags.tolist()
    dist.destroy_process_group()


def test_stub_collectives_skips_every_collective_and_leaves_no_state():
    """bench.py's `comm.exposed_ms` (round 6) = step with collectives - step with DistributedDataParallel.stub_collectives: under the switch a
    backward issues no collective at all (buckets, the tied table's early reduction, its capacity agreement and row exchange), the bucket
    gradients are the local ones / world, and the next normal step is bit-identical to the one before."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_stub_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["flags"] == [1.0, 1.0, 1.0, 1.0], ret["flags"]
