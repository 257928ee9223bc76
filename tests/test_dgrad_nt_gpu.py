"""Transposed weight twins for the input-gradient GEMMs (GPU; the default since round 4, `functional.DGRAD_NT`): the multi-tensor transpose kernel, the NT form of dX = dY W
against the k-major form and fp32 torch, and the bookkeeping that keeps W^T current — re-built after a re-cast, refreshed by
the fused optimizer's step."""
import pytest
import torch

from tests.test_kernels_gpu import DEV, close, nat, rnd

pytestmark = pytest.mark.gpu


def test_transpose_multi_matches_torch():
    shapes = [(768, 768), (2304, 768), (768, 3072), (64, 128), (3072, 768)]
    pairs = []
    for i, (r, c) in enumerate(shapes):
        src = rnd(r, c, seed=10 + i)
        pairs.append((src, torch.empty(c, r, dtype=torch.bfloat16, device=DEV)))
    nat().transpose_multi(pairs)
    for src, dst in pairs:
        assert torch.equal(dst, src.t().contiguous())
    with pytest.raises(nat().NativeLibraryError):
        nat().transpose_multi([(rnd(96, 64), torch.empty(64, 96, dtype=torch.bfloat16, device=DEV))])   # 96 % 64 != 0


@pytest.mark.parametrize("M,N,K", [(7296, 768, 768), (456, 3072, 768), (1000, 768, 3072), (228, 2304, 768)])
def test_dgrad_with_the_transposed_twin_equals_the_k_major_form(M, N, K):
    import mmf_amd.functional as Fn
    w = (torch.randn(N, K, device=DEV) * 0.05).requires_grad_(True)
    b = torch.randn(N, device=DEV).requires_grad_(True)
    x = rnd(M, K, seed=3).requires_grad_(True)
    g = rnd(M, N, seed=4)
    outs = {}
    for nt in (True, False):
        Fn.DGRAD_NT = nt
        try:
            Fn.shadows.clear()
            x.grad = w.grad = b.grad = None
            y = Fn.linear(x, w, b)
            y.backward(g)
            outs[nt] = (x.grad.clone(), w.grad.clone(), b.grad.clone())
            assert (Fn.shadows.transposed(Fn.shadows.get(w)) is not None) == nt
        finally:
            Fn.DGRAD_NT = False
    ref = g.float() @ w.detach().bfloat16().float()
    close(outs[True][0], ref, 2e-2, 2e-2 * float(ref.abs().max()), "dgrad (W^T twin)")
    close(outs[True][0], outs[False][0], 1e-2, 1e-2 * float(ref.abs().max()), "dgrad NT vs k-major")
    assert torch.equal(outs[True][1], outs[False][1]) and torch.equal(outs[True][2], outs[False][2])


@pytest.fixture
def twins_on():
    import mmf_amd.functional as Fn
    Fn.DGRAD_NT = True
    Fn.shadows.clear()
    yield Fn
    Fn.DGRAD_NT = False
    Fn.shadows.clear()


def test_twins_follow_recasts_and_the_fused_optimizer(twins_on):
    Fn = twins_on
    from mmf_amd.modules.optimizers import AdamW
    w = torch.nn.Parameter(torch.randn(128, 192, device=DEV) * 0.05)
    w16 = Fn.shadows.get(w)
    wt = Fn.shadows.transposed(w16)
    assert wt is not None and torch.equal(wt, w16.t())
    assert Fn.shadows.transposed(w16) is wt                       # cached
    assert Fn.shadows.transposed(w16[:64]) is None                # a slice is not the tracked buffer
    with torch.no_grad():
        w.mul_(2.0)                                               # version bump -> re-cast -> re-transpose
    w16b = Fn.shadows.get(w)
    wtb = Fn.shadows.transposed(w16b)
    assert torch.equal(w16b, w.detach().bfloat16()) and torch.equal(wtb, w16b.t())
    opt = AdamW([w], lr=1e-2, weight_decay=0.01)
    w.grad = torch.randn_like(w)
    before = w.detach().clone()
    opt.step()                                                    # in place: master, shadow and twin all move
    assert not torch.equal(w.detach(), before)
    assert torch.equal(Fn.shadows.get(w), w.detach().bfloat16())
    assert torch.equal(Fn.shadows.transposed(Fn.shadows.get(w)), w.detach().bfloat16().t())
    odd = torch.nn.Parameter(torch.randn(37, 128, device=DEV))    # 37 rows: no twin, the k-major path stays
    assert Fn.shadows.transposed(Fn.shadows.get(odd)) is None
