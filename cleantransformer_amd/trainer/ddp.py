"""DistributedDataParallel for one-process-per-GPU data parallelism over RCCL/xGMI.

The reference wraps its model in ``torch.nn.parallel.DistributedDataParallel`` (examples/ft_bloom_DDP.py:99,
kwargs funnel at trainer/trainer.py:1187-1207).  This class keeps that constructor / attribute surface
(``DDP(model, device_ids=[local_rank])``, ``.module``, ``state_dict`` keys prefixed ``module.``, ``no_sync()``) and
reproduces torch-DDP's semantics — parameters and buffers broadcast from rank 0 at construction; gradients bucketed in
reverse parameter order (first bucket 1 MiB, then ``bucket_cap_mb`` = 25 MiB); each bucket pre-divided by the world
size and sum-all-reduced as soon as its last gradient is produced, asynchronously, while backward keeps running — but
is built MI355X-first:

  * one flat fp32 buffer per bucket, allocated once (sized for 288 GB HBM: no per-step allocation, no re-bucketing);
    a gradient is touched ONCE on its way through: the copy into the bucket carries the 1/world pre-division, and after the
    all-reduce ``p.grad`` simply becomes the view of its bucket slot (no copy back);
  * the all-reduce is issued from the autograd hook through ``torch.distributed`` (backend "nccl" == RCCL on ROCm):
    RCCL runs it on its own HIP stream, fenced against the compute stream by events, so buckets overlap with the
    remaining backward kernels; the compute stream only waits at the end of backward;
  * on a fully connected xGMI mesh large buckets are what keeps all 7 links busy, hence the 25 MiB default stays; the
    oversized tied embedding / LM-head gradient does NOT travel as one bucket: its dense part is produced and all-reduced in
    <= 64 MiB row windows right after the LM-head backward, its embedding part as gathered rows (``_TiedGradSync``); the
    bucket it owns is only the fallback for accumulation steps.

It is transport-agnostic (``gloo`` on CPU works and is how the semantics are tested: tests/test_ddp_gloo.py).

``CTMI_DDP_BACKEND=rccl`` (round 3, opt-in): the gradient collectives of the step — bucket all-reduces, the pieces of the tied
gradient, the row exchange — go straight to an RCCL communicator owned by libctmi355 (``ctmi_ddp_*``, ops.DirectComm) whose
channel cap is the CU budget of the launch policy; torch.distributed keeps the control plane (the construction-time broadcast,
the capacity agreement, handing the communicator's id around).
"""
from __future__ import annotations

import os
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist

_MiB = 1024 * 1024


class _DirectWork:
    """what `dist.all_reduce(async_op=True)` returns, for a collective issued on the library's communicator"""

    def __init__(self, comm):
        self.comm = comm

    def wait(self):
        self.comm.wait()                                            # current stream waits for everything issued on the communicator so far


class _NullWork:
    """stands for a collective that was NOT issued (DistributedDataParallel.stub_collectives: the compute path of a rank timed alone)"""

    def wait(self):
        return None


class _Bucket:
    __slots__ = ("params", "offsets", "numel", "flat", "comm", "pending", "work", "index")

    def __init__(self, index: int, params: List[torch.nn.Parameter]):
        self.index = index
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        self.flat: Optional[torch.Tensor] = None
        self.comm: Optional[torch.Tensor] = None                 # bf16 wire copy (comm_dtype=torch.bfloat16 only)
        self.pending = 0
        self.work = None


class _TiedGradSync:
    """Early reduction of a tied embedding / LM-head gradient.  The [V,H] parameter is the FIRST parameter,
    so under plain bucketing its 1 GB gradient is the last all-reduce of the step and fully exposed.  Its value is
    ``dW_lm_head`` (dense, produced by the first backward op) + ``scatter(d_embedding_rows)`` (T rows, produced by the last).
    The dense part is pre-divided and all-reduced as soon as the LM-head weight gradient exists — under the whole rest of
    backward — and the sparse part is exchanged as rows: ``all_gather`` of every rank's token ids and row gradients
    (world·T·H elements instead of V·H), scatter-added with the 1/world factor.  The sum is linear, so this equals
    torch-DDP's average of the dense per-rank gradients up to fp32 summation order."""

    def __init__(self, owner: "DistributedDataParallel"):
        self.owner = owner
        self.works = []
        self.active = False
        self.steps = 0                                               # how many backward passes took the early path
        self._tmax = None                                            # agreed row capacity of this step's exchange (announce)
        self._announced = False
        self._tbuf = self._thost = self._tstream = self._tevent = self._tall = None

    def _single_rank(self) -> bool:
        """One rank has nothing to exchange: the tied gradient takes the generic path.  (A method so that the one-rank GPU tests can run the
        early path and the padded row exchange on the real backend by overriding it and _row_capacity — tests/test_gpu_bloom.py; the product
        carries no test switches.)"""
        return self.owner.world_size == 1

    def prescale(self, weight: torch.nn.Parameter):
        """1/world if this backward pass reduces the dense part early (the LM-head weight-gradient GEMM then applies it as
        its alpha: torch-DDP's pre-division at no cost), None for the generic bucket path (no sync, a single rank, or
        gradient accumulation pending in ``weight.grad``)."""
        o = self.owner
        if not o.require_backward_grad_sync or self._single_rank() or weight.grad is not None:
            return None
        return 1.0 / o.world_size

    def announce(self, n_tokens: int, device) -> None:
        """Called from the LM-head backward of a step that takes the early path (every rank takes it or none: the conditions —
        syncing, no accumulated gradient pending — are rank-uniform), with the token count its embedding forward recorded: agree on
        max_r T_r, the row capacity of this step's exchange.  (Announcing from the forward, as round 2 did, issued a collective for
        forwards whose backward took the bucket path or never ran — and the reference loop's zero_grad() sits BETWEEN forward and
        backward, so the forward cannot know which path the backward will take.)  The reference's collate pads each rank's batch to ITS longest sample
        (examples/ft_bloom_DDP.py:45-60, padding=True), so T = B*S differs between ranks; all_gather needs equal extents.
        The maximum is taken by a tiny all-reduce issued now and read at the end of backward; on RCCL it is copied to pinned
        host memory on a side stream, so reading it never waits for the compute stream."""
        o = self.owner
        self._tmax = None
        self._announced = True
        if self._single_rank() or o.stub_collectives:
            self._tmax = int(n_tokens)                               # single rank: nothing to agree on, no collective
            return
        on_rccl = dist.get_backend(o.process_group) == "nccl" and torch.device(device).type == "cuda"
        if o.world_size == 1 and not on_rccl:
            self._tmax = int(n_tokens)
            return
        if not on_rccl:
            t = torch.tensor([int(n_tokens)], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=o.process_group)
            self._tmax = int(t[0])
            return
        if self._tbuf is None or self._tbuf.device != torch.device(device):
            self._tbuf = torch.empty(1, dtype=torch.int64, device=device)
            self._thost = torch.empty(1, dtype=torch.int64, pin_memory=True)
            self._tstream = torch.cuda.Stream(device=device)
            self._tevent = torch.cuda.Event()
        self._tbuf.fill_(int(n_tokens))
        if o._direct is not None:
            # ONE communicator carries every collective of the step (round-3 advisor): two RCCL communicators with kernels in
            # flight on the same device are only guaranteed to make progress if both can be co-resident, which the "reserve"
            # policy (persistent GEMMs on 256 - R CUs, R channels) does not promise.  all_gather of one int64 per rank, max
            # taken on the side stream that also carries the result to pinned host memory.
            if self._tall is None or self._tall.device != self._tbuf.device or self._tall.numel() != o.world_size:
                self._tall = torch.empty(o.world_size, dtype=torch.int64, device=device)
            o._direct.all_gather(self._tall, self._tbuf)
            with torch.cuda.stream(self._tstream):
                o._direct.wait()                                     # the side stream waits for the collective; the host does not
                self._thost.copy_(self._tall.max().reshape(1), non_blocking=True)
                self._tevent.record(self._tstream)
            return
        work = dist.all_reduce(self._tbuf, op=dist.ReduceOp.MAX, group=o.process_group, async_op=True)
        with torch.cuda.stream(self._tstream):
            work.wait()                                              # the side stream waits for the collective; the host does not
            self._thost.copy_(self._tbuf, non_blocking=True)
            self._tevent.record(self._tstream)

    def _row_capacity(self, n_local: int) -> int:
        if not self._announced:
            raise RuntimeError("tied-gradient row exchange: no row capacity was agreed for this step (the LM-head backward announces it "
                               "when it starts the early dense reduction)")
        if self._tmax is None:
            self._tevent.synchronize()
            self._tmax = int(self._thost[0])
        if self._tmax < n_local:
            raise RuntimeError(f"tied-gradient row exchange: announced capacity {self._tmax} < local rows {n_local}")
        return self._tmax

    CHUNK_ROWS_BYTES = 64 * _MiB                                   # one piece of the dense [V,H] reduction (fp32 bytes)

    def chunk_rows(self, V: int, H: int) -> int:
        """Rows per piece of the chunked dense reduction: <= 64 MiB of fp32, a multiple of 256 rows (whole GEMM tiles; at
        H = 1024 that is 16384 rows = 64 x 4 tiles of 256 x 256 = exactly one round of the 256 CUs).  CTMI_DDP_TIED_CHUNK_MB
        overrides; 0 = one piece."""
        explicit = os.environ.get("CTMI_DDP_TIED_CHUNK_ROWS")        # (tests: any row count, so that tiny vocabularies take the path too)
        if explicit is not None:
            return max(1, min(int(explicit), V))
        mb = os.environ.get("CTMI_DDP_TIED_CHUNK_MB")
        nbytes = self.CHUNK_ROWS_BYTES if mb is None else int(float(mb) * _MiB)
        if nbytes <= 0:
            return V
        rows = max(256, (nbytes // (4 * H)) // 256 * 256)
        # whole rounds of the CUs the GEMM launcher may use: a piece is a [rows, H] weight gradient on 256 x 256 tiles (csrc/gemm.hip
        # pick_tile sends such pieces to the LM-head tile), i.e. rows/256 * ceil(H/256) tiles over 256 - reserved CUs — 64 MiB at
        # H = 1024 is 256 tiles: one round of 256 CUs, but 1.07 rounds of the 240 a "reserve 16" policy leaves (round-3 advisor)
        try:
            from .. import ops
            shared, reserved = ops.get_launch_policy()
        except Exception:                                            # CPU / gloo semantics tests: no library
            shared, reserved = True, 0
        tiles_n = max(1, -(-H // 256))
        slots = 256 - (0 if shared else reserved)
        if slots % tiles_n == 0:
            per_round = slots // tiles_n * 256                       # rows of the table per full round of tiles
            if rows >= per_round:
                rows = rows // per_round * per_round
        return min(rows, V)

    def begin(self, dw: torch.Tensor) -> None:
        """dw = dW_lm_head / world (contiguous fp32 [V,H] or a row range of it): start its all-reduce now, under the rest of
        backward.  Called once per piece when the LM-head weight gradient is produced in row chunks (each piece's
        collective is enqueued as soon as its GEMM is — communication starts after 1/16 of the weight-gradient work instead
        of after all of it, and no single collective is larger than 64 MiB)."""
        o = self.owner
        if o.stub_collectives:
            self.works.append(_NullWork())
        elif o._direct is not None and dw.is_cuda:
            o._direct.all_reduce(dw)
            self.works.append(_DirectWork(o._direct))
        else:
            self.works.append(dist.all_reduce(dw, op=dist.ReduceOp.SUM, group=o.process_group, async_op=True))
        if o._launch_events is not None and dw.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            o._launch_events.append(("tied", ev))
        if not self.active:
            self.active = True
            self.steps += 1

    def finish(self, dw: torch.Tensor, drows: torch.Tensor, ids: torch.Tensor) -> None:
        o = self.owner
        W = o.world_size
        ids = ids.reshape(-1).contiguous()
        drows = drows.contiguous()
        n, Hc = ids.numel(), drows.shape[-1]
        cap = self._row_capacity(n)
        if cap != n:
            # pad to the agreed capacity: id -1 (the scatter kernel skips ids outside [0, V)) and zero rows
            pid = torch.full((cap,), -1, dtype=ids.dtype, device=ids.device)
            pid[:n] = ids
            prow = torch.zeros((cap, Hc), dtype=drows.dtype, device=drows.device)
            prow[:n] = drows.view(n, Hc)
            ids, drows = pid, prow
        drows = drows.view(cap, Hc)
        all_ids = torch.empty((W, cap), dtype=ids.dtype, device=ids.device)
        all_rows = torch.empty((W, cap, Hc), dtype=drows.dtype, device=drows.device)
        if o.stub_collectives:
            # timing only: every rank's slot holds the LOCAL rows (the copies and the scatter below cost what they cost with real data)
            all_ids.copy_(ids.unsqueeze(0).expand(W, cap))
            all_rows.copy_(drows.unsqueeze(0).expand(W, cap, Hc))
        elif ids.is_cuda and dist.get_backend(o.process_group) == "gloo":
            # gloo has no device all_gather (several ranks sharing one GPU): stage through the host; RCCL takes the direct path
            hi, hr = ids.cpu(), drows.float().cpu()
            li, lr = [torch.empty_like(hi) for _ in range(W)], [torch.empty_like(hr) for _ in range(W)]
            dist.all_gather(li, hi, group=o.process_group)
            dist.all_gather(lr, hr, group=o.process_group)
            all_ids.copy_(torch.stack(li))
            all_rows.copy_(torch.stack(lr))
        elif ids.is_cuda and o._direct is not None:                  # the library's communicator
            o._direct.all_gather(all_ids, ids)
            o._direct.all_gather(all_rows, drows)
            o._direct.wait()
        elif ids.is_cuda:                                            # RCCL: straight into the [world, ...] buffers
            dist.all_gather_into_tensor(all_ids, ids, group=o.process_group)
            dist.all_gather_into_tensor(all_rows, drows, group=o.process_group)
        else:
            dist.all_gather(list(all_ids.unbind(0)), ids, group=o.process_group)
            dist.all_gather(list(all_rows.unbind(0)), drows, group=o.process_group)
        for w in self.works:
            w.wait()
        self.works = []
        self._tmax, self._announced = None, False
        _embed_scatter(all_rows.view(-1, Hc), all_ids.view(-1), dw, 1.0 / W)


def build_buckets(params: List[torch.nn.Parameter], bucket_cap_bytes: int, first_bucket_bytes: int = _MiB) -> List[List[int]]:
    """Reverse-order, size-capped bucket assignment (indices into `params`), torch-DDP style: gradients become ready
    roughly in reverse parameter order, so bucket 0 holds the LAST parameters.  A parameter larger than the cap gets a
    bucket of its own."""
    buckets, cur, cur_bytes = [], [], 0
    cap = first_bucket_bytes
    for i in reversed(range(len(params))):
        nbytes = params[i].numel() * 4
        if cur and cur_bytes + nbytes > cap:
            buckets.append(cur)
            cur, cur_bytes = [], 0
            cap = bucket_cap_bytes
        cur.append(i)
        cur_bytes += nbytes
    if cur:
        buckets.append(cur)
    return buckets


class DistributedDataParallel(torch.nn.Module):
    def __init__(self, module: torch.nn.Module, device_ids=None, output_device=None, dim=0, broadcast_buffers=True,
                 process_group=None, bucket_cap_mb=None, find_unused_parameters=False, check_reduction=False,
                 gradient_as_bucket_view=False, static_graph=False, comm_dtype=None, **kwargs):
        super().__init__()
        # comm_dtype=torch.bfloat16 (or CTMI_DDP_COMM_DTYPE=bf16): buckets travel as bf16 — half the xGMI bytes of the fp32
        # default — and are widened back into the fp32 bucket the gradients view ("O2"-style bf16 gradient communication,
        # SURVEY §8(f)2).  Opt-in: the averaged gradients then carry bf16 rounding (~2^-9 relative), outside the 1e-4 parity
        # bar of the default path.  The tied [V,H] gradient's early reduction stays fp32.
        if comm_dtype is None and os.environ.get("CTMI_DDP_COMM_DTYPE", "").lower() in ("bf16", "bfloat16"):
            comm_dtype = torch.bfloat16
        if comm_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be None / torch.float32 / torch.bfloat16")
        self.comm_dtype = None if comm_dtype == torch.float32 else comm_dtype
        # Measurement switch (bench.py `comm.exposed_ms`): True = every gradient collective of the step is SKIPPED — bucket copies, pre-division,
        # wire casts, the tied table's chunked weight gradient and row scatter all still run, under the same launch policy — so that
        # step(collectives) - step(stubbed) is the communication this rank could not hide.  The gradients are then NOT averaged: never
        # set while training.
        self.stub_collectives = False
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("DistributedDataParallel needs torch.distributed.init_process_group(...) first "
                               "(examples/ft_bloom_DDP.py:183 does init_process_group('nccl'), i.e. RCCL on ROCm)")
        if find_unused_parameters:
            raise NotImplementedError("find_unused_parameters=True is not supported: every parameter of the SFT model "
                                      "receives a gradient each step")
        self.module = module
        self.process_group = process_group if process_group is not None else dist.group.WORLD
        self.world_size = dist.get_world_size(self.process_group)
        self._arm_launch_policy()
        self._direct = self._make_direct_comm()
        self.device_ids = device_ids
        self.broadcast_buffers = broadcast_buffers
        self.bucket_cap_mb = 25 if bucket_cap_mb is None else bucket_cap_mb
        self.require_backward_grad_sync = True
        self._params = [p for p in module.parameters() if p.requires_grad]
        if not self._params:
            raise RuntimeError("DistributedDataParallel is not needed when a module doesn't have any parameter that requires a gradient.")
        self._sync_module_states()
        cap = int(self.bucket_cap_mb * _MiB)
        layout = build_buckets(self._params, cap, first_bucket_bytes=min(_MiB, cap))
        # a tied embedding / LM-head parameter gets a bucket of its own: most steps it is reduced early (_TiedGradSync) and
        # its bucket is skipped; the bucket is the fallback (gradient accumulation, non-contiguous gradient)
        tied = getattr(module, "ct_tied_weight", lambda: None)()
        # (stored through __dict__: assigning a Parameter attribute on an nn.Module would register it as a new parameter)
        self.__dict__["_tied_param"] = tied if (tied is not None and tied.requires_grad) else None
        self._tied_sync = None
        if self._tied_param is not None:
            ti = next(i for i, p in enumerate(self._params) if p is self._tied_param)
            layout = [[i for i in idxs if i != ti] for idxs in layout]
            layout = [idxs for idxs in layout if idxs] + [[ti]]
            self._tied_sync = _TiedGradSync(self)
            self._tied_param._ct_tied_sync = self._tied_sync
        self._buckets = [_Bucket(bi, [self._params[i] for i in idxs]) for bi, idxs in enumerate(layout)]
        self._where = {}
        for b in self._buckets:
            for j, p in enumerate(b.params):
                self._where[id(p)] = (b, j)
        self._callback_queued = False
        self._launch_events = None                                    # record_launch_events(): [(kind, event)] of this step's collectives
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad_ready) for p in self._params]
        # the bucket copies run in these hooks, on the compute stream: every gradient handed to autograd must be complete there (explicit
        # contract, ops.hold_deferred_wgrad_join — not something for the block backward to guess from torch's private hook lists)
        from .. import ops as _ops
        _ops.hold_deferred_wgrad_join()
        self._holds_join = True

    def _make_direct_comm(self):
        """CTMI_DDP_BACKEND=rccl: an RCCL communicator of libctmi355 for the step's gradient collectives (module docstring).  Its
        channel cap: CTMI_DDP_MAX_CHANNELS, else the reserved-CU count under CTMI_DDP_LAUNCH_POLICY=reserve, else RCCL's default."""
        if os.environ.get("CTMI_DDP_BACKEND", "torch").lower() not in ("rccl", "rccl_direct", "direct"):
            return None
        if dist.get_backend(self.process_group) != "nccl" or not any(p.is_cuda for p in self.module.parameters()):
            raise RuntimeError("CTMI_DDP_BACKEND=rccl needs parameters on the GPU and a torch.distributed group on the nccl (RCCL) backend "
                               "(it carries the communicator's id to the other ranks)")
        from .. import ops
        rank = dist.get_rank(self.process_group)
        box = [ops.DirectComm.unique_id() if rank == 0 else None]
        src = dist.get_global_rank(self.process_group, 0) if hasattr(dist, "get_global_rank") else 0
        dist.broadcast_object_list(box, src=src, group=self.process_group)
        cap = int(os.environ.get("CTMI_DDP_MAX_CHANNELS", "0"))
        if cap <= 0 and os.environ.get("CTMI_DDP_LAUNCH_POLICY", "flow").lower() == "reserve":
            cap = max(1, min(128, int(os.environ.get("CTMI_DDP_COMM_CUS", "16"))))
        return ops.DirectComm(box[0], rank, self.world_size, cap)

    def _arm_launch_policy(self):
        """world > 1: the all-reduce kernels will hold CUs under backward — switch the GEMM launcher to a policy that tolerates it
        (ctmi_set_launch_policy; csrc/gemm.hip).  An explicit library call, re-made at every training forward, so it also takes
        effect for a model that already ran GEMMs before it was wrapped.
          CTMI_DDP_LAUNCH_POLICY = "flow" is the default since round 6 (below);
          "shared" (also "1"; the default of rounds 2-5): no persistent launches — one workgroup per tile, the hardware
              dispatcher flows them over whatever CUs the collectives leave free;
          "reserve": persistent launches on 256 - R CUs, R = CTMI_DDP_COMM_CUS (default 16) — meant to be paired with
              NCCL_MAX_NCHANNELS = R (RCCL runs one workgroup per channel; bench.py exports both before init_process_group, the
              communicator reads the variable when it is created), so the R channels always find a free CU and the GEMMs never wait
              for one they cannot get;
          "flow" (round 6): one workgroup per tile like "shared", but the ping-pong tiles of the single-GPU policy instead of the co-resident
              128x128 / 256x128 family — at world 1 it costs a fraction of what "shared" does (profiles/r06_ddp_policy_world1.json);
          "persistent": the single-GPU policy unchanged (persistent launches on all 256 CUs): the fastest when the collectives' workgroups
              co-reside with the GEMMs' or are short — nobody could measure that without a multi-GPU node, so bench.py --gpus N times it too;
          "0": leave the launcher alone."""
        mode = os.environ.get("CTMI_DDP_LAUNCH_POLICY", "flow").lower()
        if self.world_size > 1 and mode != "0" and any(p.is_cuda for p in self.module.parameters()):
            from .. import ops
            if mode == "reserve":
                want = (0, max(0, min(128, int(os.environ.get("CTMI_DDP_COMM_CUS", "16")))))
            elif mode == "persistent":
                want = (0, 0)                                            # the single-GPU policy (bench.py times it as a third candidate on the node it runs on)
            elif mode == "flow":
                want = (2, ops.get_launch_policy()[1])                   # one workgroup per tile, the single-GPU tile choice (round 6)
            else:
                want = (1, ops.get_launch_policy()[1])
            if ops.get_launch_policy() != want:
                ops.set_launch_policy(*want)

    # ---------------------------------------------------------------- construction-time broadcast (rank 0 wins)
    def _sync_module_states(self):
        tensors = [p.data for p in self.module.parameters()]
        if self.broadcast_buffers:
            tensors += [b.data for b in self.module.buffers()]
        self._broadcast_coalesced(tensors)
        # the broadcast wrote through .data (no version bump): drop compute-dtype copies cached by an earlier forward
        from .. import ops
        for p in self.module.parameters():
            ops.invalidate_compute_copies(p)

    def _broadcast_coalesced(self, tensors, chunk_bytes: int = 256 * _MiB):
        seen, groups = set(), {}
        for t in tensors:
            if t.data_ptr() in seen or t.numel() == 0:
                continue
            seen.add(t.data_ptr())
            groups.setdefault((t.dtype, t.device), []).append(t)
        for (_, _), ts in groups.items():
            batch, nbytes = [], 0
            for t in ts + [None]:
                if t is not None and (not batch or nbytes + t.numel() * t.element_size() <= chunk_bytes):
                    batch.append(t)
                    nbytes += t.numel() * t.element_size()
                    continue
                flat = torch.cat([x.reshape(-1) for x in batch])
                dist.broadcast(flat, src=dist.get_global_rank(self.process_group, 0) if hasattr(dist, "get_global_rank") else 0,
                               group=self.process_group)
                off = 0
                for x in batch:
                    x.copy_(flat[off:off + x.numel()].view_as(x))
                    off += x.numel()
                batch, nbytes = ([t], t.numel() * t.element_size()) if t is not None else ([], 0)

    # ---------------------------------------------------------------- backward-time reduction
    def _on_grad_ready(self, p: torch.nn.Parameter):
        if not self.require_backward_grad_sync:
            return
        if not self._callback_queued:
            self._callback_queued = True
            for b in self._buckets:
                b.pending = len(b.params)
                b.work = None
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize_backward)
        b, j = self._where[id(p)]
        if p is self._tied_param and self._tied_sync.active:
            self._tied_sync.active = False                          # reduced early, rows exchanged: p.grad is final already
            b.pending -= 1
            return
        if b.flat is None or b.flat.device != p.grad.device:
            b.flat = torch.empty(b.numel, dtype=torch.float32, device=p.grad.device)
        off = b.offsets[j]
        slot = b.flat[off:off + p.numel()]
        g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
        # torch-DDP order: divide by the world size, then sum.  The division rides on the copy into the bucket (one pass
        # over the gradient instead of copy + scale + copy-back); a gradient that already lives in its bucket slot (it was
        # accumulated in place into last step's view, see _finalize_backward) is scaled where it is.
        _scale_copy(g.reshape(-1), slot, 1.0 / self.world_size)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket):
        wire = b.flat
        if self.comm_dtype is not None:
            if b.comm is None or b.comm.device != b.flat.device:
                b.comm = torch.empty(b.numel, dtype=self.comm_dtype, device=b.flat.device)
            _cast(b.flat, b.comm)                                   # one pass per bucket; the collective is enqueued behind it
            wire = b.comm
        if self.stub_collectives:
            b.work = _NullWork()
        elif self._direct is not None and wire.is_cuda:
            self._direct.all_reduce(wire)
            b.work = _DirectWork(self._direct)
        else:
            b.work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
        if self._launch_events is not None and wire.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._launch_events.append(("bucket", ev))

    def _finalize_backward(self):
        self._callback_queued = False
        missing = [b.index for b in self._buckets if b.pending != 0]
        if missing:
            raise RuntimeError(f"DistributedDataParallel: buckets {missing} did not receive all of their gradients in this "
                               f"backward pass (unused parameters are not supported)")
        for b in self._buckets:
            if b.work is None:                                      # the tied parameter's bucket in a step that reduced it early
                continue
            b.work.wait()                                           # compute stream waits on the RCCL stream; host does not block
            if self.comm_dtype is not None:
                _cast(b.comm, b.flat)                               # widen the reduced wire copy back into the fp32 bucket
            for p, off in zip(b.params, b.offsets):
                # the averaged gradient IS the bucket slot from here on: no copy back (a dense, sliceable [shape] fp32
                # view — what lm_head.weight.grad[100:110, 100:110], ft_bloom_DDP.py:148, needs)
                p.grad = b.flat[off:off + p.numel()].view(p.shape)
            b.work = None

    def record_launch_events(self, on: bool = True) -> list:
        """Diagnostics / tests: from now on every collective launched by the backward hooks records a HIP event on the compute
        stream at its launch point (("bucket" | "tied", event)); returns the list they are appended to (cleared here)."""
        self._launch_events = [] if on else None
        return self._launch_events

    @contextmanager
    def no_sync(self):
        """Gradient accumulation: skip the all-reduce inside this context (same contract as torch DDP)."""
        old = self.require_backward_grad_sync
        self.require_backward_grad_sync = False
        try:
            yield
        finally:
            self.require_backward_grad_sync = old

    def _reset_step_state(self):
        """A backward pass that raised after its first gradient hook leaves the per-step bookkeeping half-way (the end-of-
        backward callback never ran): start every training forward from a clean slate."""
        self._callback_queued = False
        for b in self._buckets:
            b.pending = 0
            b.work = None
        if self._tied_sync is not None:
            self._tied_sync.active = False
            self._tied_sync.works = []

    def forward(self, *inputs, **kwargs):
        if torch.is_grad_enabled():
            self._reset_step_state()
            self._arm_launch_policy()
        return self.module(*inputs, **kwargs)

    def __del__(self):
        try:
            if getattr(self, "_holds_join", False):
                from .. import ops as _ops
                _ops.release_deferred_wgrad_join()
                self._holds_join = False
        except Exception:                                              # noqa: BLE001  (interpreter shutdown)
            pass

    def close(self) -> None:
        """Detach from the module: remove the gradient hooks, drop the bucket memory and the library communicator.  bench.py's
        configuration probe wraps the same model several times (backend x launch policy) and keeps one."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if getattr(self, "_holds_join", False):
            from .. import ops as _ops
            _ops.release_deferred_wgrad_join()
            self._holds_join = False
        for b in self._buckets:
            b.flat = b.comm = b.work = None
        if self._tied_param is not None and getattr(self._tied_param, "_ct_tied_sync", None) is self._tied_sync:
            self._tied_param._ct_tied_sync = None
        if self._direct is not None:
            self._direct.close()
            self._direct = None

    def bucket_summary(self):
        return [(b.index, len(b.params), b.numel * 4) for b in self._buckets]


def _embed_scatter(rows: torch.Tensor, ids: torch.Tensor, dtable: torch.Tensor, s: float) -> None:
    from .. import ops
    ops.embed_bwd(rows, ids, dtable, s)                             # (the CPU/gloo semantics tests patch ops.* with emulations)


def _cast(src: torch.Tensor, dst: torch.Tensor) -> None:
    from .. import ops
    ops.cast(src, dst.dtype, out=dst)


def _scale_copy(src: torch.Tensor, dst: torch.Tensor, s: float) -> None:
    from .. import ops
    ops.scale_copy(src, dst, s)
