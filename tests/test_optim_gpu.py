"""Single-launch Adam (include/s3g_optim.h) vs torch.optim.Adam on the CPU: same state layout, same update, to fp32
round-off (the reference builds torch.optim.Adam(l, lr=0.0, eps=1e-15), scene/gaussian_model.py:177-189)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev):
    g = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (1000, 15, 3), (7,), (64, 64), (1, 32, 8, 16), (5, 1), (33,)]
    ps = []
    for s in shapes:
        t = torch.randn(s, generator=g)
        if len(s) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t.to(dev)))
    return ps


def _groups(ps):
    return [{"params": ps[:2], "lr": 1.6e-4, "name": "a"}, {"params": ps[2:5], "lr": 2.5e-3, "name": "b"},
            {"params": ps[5:], "lr": 0.05, "name": "c"}]


def test_adam_matches_torch_adam(gpu_device):
    from s3gaussian_amd.optim import Adam
    ref_p, my_p = _params("cpu"), _params(gpu_device)
    ref = torch.optim.Adam(_groups(ref_p), lr=0.0, eps=1e-15)
    mine = Adam(_groups(my_p), lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        for k, (a, b) in enumerate(zip(ref_p, my_p)):
            if k == 6 and step < 2:          # a parameter without gradient for the first steps (unused head)
                a.grad = b.grad = None
                continue
            grad = torch.randn(a.shape, generator=g) * (10.0 ** (k - 3))
            if a.dim() == 4:                 # gradient arriving in another memory format than the parameter
                grad = grad.contiguous() if step % 2 else grad.contiguous(memory_format=torch.channels_last)
            a.grad, b.grad = grad.clone(), grad.to(gpu_device)
            if k == 3:                       # gradient = view into a flat buffer at an odd offset (not 16-byte aligned)
                flat = torch.empty(grad.numel() + 1, device=gpu_device)
                flat[1:].copy_(grad.reshape(-1).to(gpu_device))
                b.grad = flat[1:].view(grad.shape)
        ref.step()
        mine.step()
    for a, b in zip(ref_p, my_p):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=1e-7)
        sa, sb = ref.state[a], mine.state[b]
        assert set(sb) == {"step", "exp_avg", "exp_avg_sq"} and float(sb["step"]) == float(sa["step"])
        # moments: fp32 round-off relative to the tensor's scale (an element of exp_avg can cancel to ~0)
        for name in ("exp_avg", "exp_avg_sq"):
            want = sa[name].numpy()
            np.testing.assert_allclose(sb[name].cpu().numpy(), want, rtol=2e-6, atol=1e-6 * float(np.abs(want).max()))
    assert my_p[4].is_contiguous(memory_format=torch.channels_last)
    sd = mine.state_dict()                   # interchangeable with torch.optim.Adam's
    fresh = torch.optim.Adam(_groups(_params(gpu_device)), lr=0.0, eps=1e-15)
    fresh.load_state_dict(sd)


def test_adam_more_than_one_launch_chunk(gpu_device):
    from s3gaussian_amd.optim import Adam, MAX_TENSORS
    n = MAX_TENSORS + 9
    ps = [torch.nn.Parameter(torch.full((5,), float(k), device=gpu_device)) for k in range(n)]
    opt = Adam(ps, lr=0.1, eps=1e-8)
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()
    for k, p in enumerate(ps):               # first Adam step moves every element by lr * sign(g)
        np.testing.assert_allclose(p.detach().cpu().numpy(), np.full(5, k - 0.1, dtype=np.float32), rtol=1e-6)


def test_adam_refuses_cpu_parameters():
    from s3gaussian_amd.optim import Adam
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="GPU"):
        Adam([p], lr=0.1).step()


def test_adam_grad_scale_equals_scaled_gradients(gpu_device):
    """grad_scale = 1/world folds the data-parallel average into the kernel: same as stepping on grad * scale."""
    from s3gaussian_amd.optim import Adam
    g = torch.Generator().manual_seed(3)
    a = torch.nn.Parameter(torch.randn(513, 3, generator=g))
    b = torch.nn.Parameter(a.detach().clone().to(gpu_device))
    ref, mine = torch.optim.Adam([a], lr=0.01, eps=1e-15), Adam([b], lr=0.01, eps=1e-15)
    mine.grad_scale = 0.125
    for _ in range(4):
        grad = torch.randn(513, 3, generator=g)
        a.grad, b.grad = grad * 0.125, grad.to(gpu_device)
        ref.step()
        mine.step()
    np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=2e-6, atol=1e-7)
