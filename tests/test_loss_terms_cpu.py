"""pipeline._WeightedTerms (the regulariser terms of train.py:404-425 combined in one autograd node) against the reference's
term-by-term expression: value and every gradient.  Pure PyTorch, runs without a GPU."""
import pytest
import torch

from s3gaussian_amd.pipeline import _WeightedTerms


@pytest.mark.parametrize("with_dx", [True, False])
@pytest.mark.parametrize("n_terms", [1, 2, 3])
def test_weighted_terms_match_the_term_by_term_sum(with_dx, n_terms):
    torch.manual_seed(n_terms)
    dx = torch.randn(257, 3, requires_grad=True) if with_dx else None
    terms = [torch.randn((), requires_grad=True) for _ in range(n_terms)]
    weights = (1.0, 0.001, 1.0)[:n_terms]
    lam_dx = 0.013
    ref = sum(t * w for t, w in zip(terms, weights))
    if with_dx:
        ref = ref + torch.mean(torch.abs(dx)) * lam_dx
    (ref * 0.7).backward()     # a non-unit upstream gradient
    want = [t.grad.clone() for t in terms] + ([dx.grad.clone()] if with_dx else [])
    for t in terms + ([dx] if with_dx else []):
        t.grad = None
    out = _WeightedTerms.apply(dx, lam_dx, weights, *terms)
    (out * 0.7).backward()
    got = [t.grad for t in terms] + ([dx.grad] if with_dx else [])
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-7)
    for g, w in zip(got, want):
        assert torch.allclose(g, w, rtol=1e-6, atol=1e-9)


def test_weights_are_uploaded_once_per_configuration():
    from s3gaussian_amd import pipeline
    pipeline._weight_cache.clear()
    a, b = torch.tensor(0.5, requires_grad=True), torch.tensor(2.0, requires_grad=True)
    for _ in range(3):
        _WeightedTerms.apply(None, 0.0, (1.0, 0.25), a, b)
    assert len(pipeline._weight_cache) == 1


def test_empty_dx_contributes_nothing():
    a = torch.tensor(0.5, requires_grad=True)
    out = _WeightedTerms.apply(torch.zeros(0, 3, requires_grad=True), 0.01, (2.0,), a)
    out.backward()
    assert float(out.detach()) == 1.0 and float(a.grad) == 2.0
