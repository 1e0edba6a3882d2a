"""vp_wgrad_slab_reduce_f32: the split-K weight-gradient launches' second half (slab[split][tap][Cs][Cb] -> dw[Cs][Cb][tap]).
The 16-B-load kernel the library dispatches for the split-operand layers must return the same BITS as the 4-B-load kernel
(same summation order) and both must agree with an fp64 sum; shapes = the benchmark step's layers, the tap-pair split depths,
a 3x3 layer, and shapes where the wide kernel does not apply (dispatch falls back)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("Cs,Cb,ns,nt", [(256, 256, 3, 25), (128, 64, 12, 25), (64, 32, 32, 25), (512, 256, 1, 25), (72, 64, 5, 9),
                                          (24, 200, 7, 25), (96, 64, 64, 1), (34, 124, 3, 25), (64, 32, 2, 25), (16, 8, 40, 25)])
def test_variants_are_bit_identical_and_match_fp64(Cs, Cb, ns, nt):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(Cs * 7 + Cb + ns)
    slab = (torch.randn(ns, nt, Cs, Cb, generator=g) * torch.logspace(-3, 3, ns).view(ns, 1, 1, 1)).to(DEV)
    ref = slab.double().sum(0).permute(1, 2, 0).contiguous()                 # [Cs][Cb][nt]
    outs = [ops.wgrad_slab_reduce(slab, Cs, Cb, nt, variant=v) for v in (0, 1, -1)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    scale = slab.double().abs().sum(0).permute(1, 2, 0)
    assert ((outs[0].double() - ref).abs() <= 4e-7 * scale + 1e-30).all()


def test_rejects_bad_arguments():
    from vae_play_amd import _lib, ops
    slab = torch.zeros(2, 25, 64, 64, device=DEV)
    with pytest.raises(_lib.VaePlayHipError):
        ops.wgrad_slab_reduce(slab, 64, 64, 25, variant=5)
