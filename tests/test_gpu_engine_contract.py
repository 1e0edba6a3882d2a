"""FusedVAEStep's contract with its optimiser and its inputs (round-3 fixes of ADVICE r2 / VERDICT r2 weak #9):
  * gradients the plan wrote into the arena survive ``zero_grad(set_to_none=True)`` / ``module.zero_grad()`` (train.py:68);
  * two engines over ONE optimiser each contract fc.0's factored update from their OWN buffers;
  * a batch of the wrong shape is refused with a VaePlayHipError instead of being broadcast into the static buffers."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params(vae):
    return {n: q.detach().clone() for n, q in vae.named_parameters()}


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("how", ["optimizer", "module"])
def test_step_after_zero_grad_set_to_none(precision, how):
    from tests.test_gpu_engine import build
    from oracle import ref_cpu as O
    C, S, z, B = 3, 32, 16, 8
    x, eps = O.synthetic_batch(B, C, S, z)
    xd, epsd = x.to(DEV), eps.to(DEV)
    out = []
    for drop in (False, True):
        vae, opt, fused, p0, L = build(C, S, z, B, precision=precision)
        before = _params(vae)
        for _ in range(2):
            if drop:
                if how == "optimizer":
                    opt.zero_grad(set_to_none=True)
                else:
                    vae.zero_grad()           # torch's module.zero_grad(): every .grad becomes None
                assert vae.encoder.conv[0].conv.weight.grad is None
            fused.step(xd, epsd)
        torch.cuda.synchronize()
        after = _params(vae)
        moved = {n: (after[n] - before[n]).abs().max().item() for n in after}
        assert all(v > 0 for v in moved.values()), [n for n, v in moved.items() if v == 0]
        out.append(after)
    for n in out[0]:        # same kernels, same inputs, same schedule: bit-identical parameters
        assert torch.equal(out[0][n], out[1][n]), n


def test_forward_backward_then_plain_optimizer_step_after_zero_grad():
    """the documented mixed use: forward_backward() (every gradient materialised) + the optimiser's own step()"""
    from tests.test_gpu_engine import build
    from oracle import ref_cpu as O
    C, S, z, B = 1, 32, 16, 4
    x, eps = O.synthetic_batch(B, C, S, z)
    outs = []
    for drop in (False, True):
        vae, opt, fused, p0, L = build(C, S, z, B)
        if drop:
            opt.zero_grad(set_to_none=True)
        fused.forward_backward(x.to(DEV), eps.to(DEV))
        assert all(p.grad is not None for p in vae.parameters())
        opt.step()
        torch.cuda.synchronize()
        outs.append(_params(vae))
    for n in outs[0]:
        assert torch.equal(outs[0][n], outs[1][n]), n


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_two_engines_share_one_optimizer(precision):
    """B = 4 and B = 8 plans over the same Adam, stepped alternately: the factored fc.0 update (VP_ADAM_OUTER=1) must equal the
    materialised-gradient path (VP_ADAM_OUTER=0), which has no per-engine state in the optimiser."""
    import vae_play_amd as V
    from vae_play_amd import engine, optim
    from oracle import ref_cpu as O
    C, S, z = 3, 32, 16
    L = O.iter_level_for(S)
    batches = {B: tuple(t.to(DEV) for t in O.synthetic_batch(B, C, S, z)) for B in (4, 8)}
    res = {}
    old = os.environ.get("VP_ADAM_OUTER")
    try:
        for mode in ("1", "0"):
            os.environ["VP_ADAM_OUTER"] = mode
            vae = V.VAE(S, z, C, init_rule=False)
            vae.load_state_dict(O.init_params(C, z, L, seed=0))
            vae.to(DEV).train()
            opt = optim.Adam(vae.parameters(), lr=1e-4)
            eng = {B: engine.FusedVAEStep(vae, opt, B, S, C, precision=precision) for B in (4, 8)}
            for B in (4, 8, 4):
                eng[B].step(*batches[B])
            torch.cuda.synchronize()
            res[mode] = (vae.encoder.fc[0].weight.detach().clone(), opt.exp_avg.clone())
    finally:
        if old is None:
            os.environ.pop("VP_ADAM_OUTER", None)
        else:
            os.environ["VP_ADAM_OUTER"] = old
    w1, w0 = res["1"][0], res["0"][0]
    d = (w1 - w0).abs()
    # a stale-factor update would move most of fc.0 the wrong way by ~lr per step; rounding flips only isolated elements
    assert (d > 2e-5).double().mean().item() <= 2e-3
    m1, m0 = res["1"][1], res["0"][1]
    rel = ((m1 - m0).double().pow(2).sum().sqrt() / m0.double().pow(2).sum().sqrt()).item()
    assert rel <= 3e-2, rel


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_wrong_input_shapes_are_refused(precision):
    from tests.test_gpu_engine import build
    from vae_play_amd._lib import VaePlayHipError
    from oracle import ref_cpu as O
    C, S, z, B = 3, 32, 16, 8
    vae, opt, fused, p0, L = build(C, S, z, B, precision=precision)
    x, eps = O.synthetic_batch(B, C, S, z)
    xd, epsd = x.to(DEV), eps.to(DEV)
    with pytest.raises(VaePlayHipError, match="x has shape"):
        fused.step(xd[:1], epsd)                     # broadcastable: used to be silently broadcast over the batch
    with pytest.raises(VaePlayHipError, match="x has shape"):
        fused.step(xd[:5], epsd[:5])                 # the short last batch of a DataLoader without drop_last
    with pytest.raises(VaePlayHipError, match="eps has shape"):
        fused.step(xd, epsd[:, :1])
    with pytest.raises(VaePlayHipError, match="x has shape"):
        fused.step(xd.reshape(B, C * S, S), epsd)
    fused.step(xd, epsd)                             # and the right shapes still run
    fused.step(xd.cpu(), epsd.double())              # other device / dtype: copied into the static buffers
    torch.cuda.synchronize()
