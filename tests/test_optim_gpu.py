"""Fused flat Nesterov SGD vs torch.optim.SGD (the reference's optimiser, instantiators.py:74-92) and the row gather."""
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Toy(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.a = torch.nn.Parameter(torch.randn(37, 5, generator=g))
        self.b = torch.nn.Parameter(torch.randn(129, generator=g))
        self.c = torch.nn.Parameter(torch.randn(4, 3, 5, generator=g))  # total 374: not a multiple of 4


@pytest.mark.parametrize("nesterov,wd,mu", [(True, 1e-4, 0.9), (False, 0.0, 0.9), (False, 1e-2, 0.0)])
def test_flat_sgd_matches_torch_sgd(nesterov, wd, mu):
    from feature_vs_text_compound_emotion_amd.data_parallel import ClipDataParallel, FlatNesterovSGD
    ref, mine = _Toy(1).cuda(), _Toy(1).cuda()
    opt_ref = torch.optim.SGD(ref.parameters(), lr=1e-3, momentum=mu, weight_decay=wd, nesterov=nesterov)
    ddp = ClipDataParallel(mine, world_size=1, broadcast=False)
    opt = FlatNesterovSGD(ddp, lr=1e-3, momentum=mu, weight_decay=wd, nesterov=nesterov)
    g = torch.Generator().manual_seed(2)
    for step in range(4):
        if step == 2:  # the reference's scheduler mutates param_groups[0]['lr'] (base/scheduler.py:167-197)
            opt_ref.param_groups[0]["lr"] = opt.param_groups[0]["lr"] = 3e-4
        opt_ref.zero_grad()
        opt.zero_grad()
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            grad = torch.randn(pr.shape, generator=g).cuda()
            pr.grad = grad.clone()
            pm.grad = grad.clone()  # what autograd leaves after zero_grad(): the optimiser gathers it into the flat bucket
        opt_ref.step()
        opt.step()
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            # same operations in the same order; torch's kernels may or may not contract a*b+c, hence 1 ulp
            assert (pr - pm).abs().max().item() <= 2.4e-7 * max(1.0, pr.abs().max().item())
    assert mine.a.data_ptr() == opt.flat_param.data_ptr()  # parameters live in the flat buffer


def test_gather_rows_with_missing_rows():
    from feature_vs_text_compound_emotion_amd import ops
    src = torch.randn(50, 768, generator=torch.Generator().manual_seed(0)).cuda()
    idx = torch.tensor([3, 3, -1, 49, 0, -1, 7], dtype=torch.int64).cuda()
    out = ops.gather_rows(src, idx)
    ref = src[idx.clamp(min=0)] * (idx >= 0).unsqueeze(1)
    assert torch.equal(out, ref)
