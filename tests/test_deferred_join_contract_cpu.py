"""CPU: when may a block's parameter gradients complete on the weight-gradient side stream AFTER its autograd node has returned?
(ops.params_allow_deferred_grads — round-4 advisor / verdict item 5b.)  Only when nothing touches them before the end of the backward pass;
wrappers that install gradient hooks say so EXPLICITLY (ops.hold_deferred_wgrad_join), the block backward does not have to guess it from
torch's private hook lists — which stay a second line of defence."""
import types

import torch

from cleantransformer_amd import ops


def _params(n=3):
    return [torch.nn.Parameter(torch.randn(4, 4)) for _ in range(n)]


def _ctx():
    return types.SimpleNamespace(seen_params=set(), shared_params=False)      # the two fields of ops.MaskInfo that note_block_params uses


def test_fresh_parameters_inside_a_backward_pass_allow_it():
    with torch.no_grad():                                                     # inside autograd.Function.backward grad mode is off
        assert ops.params_allow_deferred_grads(_params())


def test_create_graph_backward_refuses():
    assert torch.is_grad_enabled()
    assert not ops.params_allow_deferred_grads(_params())                     # backward(create_graph=True) runs with grad mode on


def test_pending_accumulation_and_torch_hooks_refuse():
    with torch.no_grad():
        ps = _params()
        ps[1].grad = torch.zeros(4, 4)
        assert not ops.params_allow_deferred_grads(ps)
        ps = _params()
        h = ps[0].register_hook(lambda g: g)
        assert not ops.params_allow_deferred_grads(ps)
        h.remove()
        assert ops.params_allow_deferred_grads(ps)
        h = ps[2].register_post_accumulate_grad_hook(lambda p: None)
        assert not ops.params_allow_deferred_grads(ps)
        h.remove()
        assert ops.params_allow_deferred_grads(ps)


def test_an_explicit_hold_refuses_whatever_the_hook_lists_say():
    """the contract: a wrapper with gradient hooks torch's lists do not show (hooks on the AccumulateGrad node, as torch DDP / FSDP / apex
    register them) holds the join; with a hold active the deferred path must not be taken"""
    ps = _params()
    acc = (ps[0] * 1.0).grad_fn.next_functions[0][0]                          # the AccumulateGrad node: invisible to the attribute sniffing
    hk = acc.register_hook(lambda *a: None)
    with torch.no_grad():
        assert ops.params_allow_deferred_grads(ps)                            # <- exactly why the contract exists
        ops.hold_deferred_wgrad_join()
        try:
            assert not ops.params_allow_deferred_grads(ps)
            ops.hold_deferred_wgrad_join()
            ops.release_deferred_wgrad_join()
            assert not ops.params_allow_deferred_grads(ps)                    # holds nest
        finally:
            ops.release_deferred_wgrad_join()
        assert ops.params_allow_deferred_grads(ps)
        hk.remove()


def test_a_parameter_shared_by_two_blocks_of_one_forward_refuses():
    with torch.no_grad():
        a, b = _params(), _params()
        fwd = _ctx()
        ops.note_block_params(fwd, a)
        ops.note_block_params(fwd, b)
        assert ops.params_allow_deferred_grads(a, fwd) and ops.params_allow_deferred_grads(b, fwd)
        fwd2 = _ctx()
        ops.note_block_params(fwd2, a)
        ops.note_block_params(fwd2, [b[0], a[1]])                             # a[1] feeds two block nodes: the engine sums its two gradients
        assert not ops.params_allow_deferred_grads(b, fwd2)
        assert ops.params_allow_deferred_grads(b, fwd)                        # another forward of the same process is not affected


def test_the_data_parallel_wrapper_holds_the_join_for_its_lifetime():
    import torch.distributed as dist
    from cleantransformer_amd.trainer.ddp import DistributedDataParallel
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    try:
        m = torch.nn.Linear(8, 8)
        with torch.no_grad():
            assert ops.params_allow_deferred_grads(list(m.parameters()))
        w = DistributedDataParallel(m)
        with torch.no_grad():
            ps = _params()                                                    # not even the wrapper's own parameters: the hold is process-wide
            assert not ops.params_allow_deferred_grads(ps)
        w.close()
        with torch.no_grad():
            assert ops.params_allow_deferred_grads(ps)
        w2 = DistributedDataParallel(torch.nn.Linear(4, 4))
        with torch.no_grad():
            assert not ops.params_allow_deferred_grads(ps)
        w2.close()
        w2.close()                                                            # idempotent: one hold per wrapper
        with torch.no_grad():
            assert ops.params_allow_deferred_grads(ps)
    finally:
        dist.destroy_process_group()
