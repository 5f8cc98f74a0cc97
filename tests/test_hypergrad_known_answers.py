"""CPU: the four known-answer tests of /root/reference/test_hypergrad.ipynb (cells 3-9, 13-25, 28-30) re-expressed
as pytest, run against (a) the product operators adkf_ift_amd.hypergradient.{cauchy_hypergradient,
cauchy_hypergradient_jvp} (generic path), (b) the oracle's dense restatement; plus the golden linear-map cases whose
expected theta.grad came from the REFERENCE's own two files, and - only where /root/reference is mounted - a live
cross-check against those files."""
import importlib.util
import math
import os

import numpy as np
import pytest
import torch

from adkf_ift_amd.hypergradient import cauchy_hypergradient, cauchy_hypergradient_jvp
from oracle import gp_oracle as O
from oracle.hypergrad_oracle import dense_ift_hypergradient

CPU = torch.device("cpu")


def _oracle_variant(f_outer, f_inner, params_outer, params_inner, device, **kw):
    kw.pop("sanity_checks", None)
    return dense_ift_hypergradient(f_outer, f_inner, params_outer, params_inner, **kw)


VARIANTS = [cauchy_hypergradient, cauchy_hypergradient_jvp, _oracle_variant]


def sum_of_squares(params_outer, params_inner):
    s = 0.0
    for tup in (params_outer, params_inner):
        for p in tup:
            s = s + (p ** 2).sum()
    return s


@pytest.mark.parametrize("fn", VARIANTS)
def test_sum_of_squares(fn):  # cells 3-9: a.grad = 2a, b.grad = 2b, value = sum of squares
    torch.manual_seed(1)
    a = torch.randn(3, 4, requires_grad=True)
    b = torch.randn(5, requires_grad=True)
    val = fn(sum_of_squares, sum_of_squares, (a,), (b,), CPU)
    assert torch.allclose(val.detach(), (a ** 2).sum().detach() + (b ** 2).sum().detach())
    assert torch.allclose(a.grad, 2 * a.detach(), atol=1e-6)
    assert torch.allclose(b.grad, 2 * b.detach(), atol=1e-6)


def quadratic(params_outer, params_inner):
    a, b, c = params_outer
    (x,) = params_inner
    return a * (x ** 2) + b * x + c


@pytest.mark.parametrize("fn", VARIANTS)
def test_scalar_quadratic(fn):  # cells 13-21: (b^2/4a^2, -b/2a, 1)
    torch.manual_seed(2)
    a = torch.exp(torch.randn(1))[0].requires_grad_(True)
    b = torch.randn(1)[0].requires_grad_(True)
    c = torch.randn(1)[0].requires_grad_(True)
    x = torch.randn(1)[0].requires_grad_(True)
    with torch.no_grad():
        x.fill_(-b / 2 / a)
    fn(quadratic, quadratic, (a, b, c), (x,), CPU)
    with torch.no_grad():
        expect = (b ** 2 / 4 / a ** 2, -b / 2 / a, torch.tensor(1.0))
    for got, e in zip((a.grad, b.grad, c.grad), expect):
        assert torch.allclose(got, e, atol=1e-6)


@pytest.mark.parametrize("fn", VARIANTS)
def test_ignore_direct_grad_gives_zero_residual(fn):  # cells 22-25
    torch.manual_seed(3)
    a = torch.exp(torch.randn(1))[0].requires_grad_(True)
    b = torch.randn(1)[0].requires_grad_(True)
    c = torch.randn(1)[0].requires_grad_(True)
    x = torch.randn(1)[0].requires_grad_(True)
    with torch.no_grad():
        x.fill_(-b / 2 / a)
    fn(quadratic, quadratic, (a, b, c), (x,), CPU, ignore_direct_grad=True)
    for g in (a.grad, b.grad, c.grad):
        assert torch.allclose(g, torch.zeros(()), atol=1e-6)


def f_inner_two(params_outer, params_inner):
    a, b, c = params_outer
    x1, x2 = params_inner
    return (torch.sum(a * (x1 ** 2) + b * x1 + c) + torch.sum(a * (x2 ** 2) + b * x2 + c)) / 2


def f_outer_two(params_outer, params_inner):
    a, b, c = params_outer
    x1, x2 = params_inner
    return torch.sum(a * (x1 + x2)) / 2


@pytest.mark.parametrize("fn", VARIANTS)
def test_hundred_random_trials(fn):  # cells 28-30: a.grad ~ 0, b.grad = -0.5, c.grad ~ 0
    torch.manual_seed(4)
    for _ in range(100 if fn is not _oracle_variant else 20):
        D = 3
        a = torch.exp(torch.randn(D)).requires_grad_(True)
        b = torch.randn(D).requires_grad_(True)
        c = torch.randn(D).requires_grad_(True)
        x1 = torch.randn(D).requires_grad_(True)
        x2 = torch.randn(D).requires_grad_(True)
        with torch.no_grad():
            x1.copy_(-b / 2 / a)
            x2.copy_(-b / 2 / a)
        fn(f_outer_two, f_inner_two, (a, b, c), (x1, x2), CPU)
        assert np.allclose(a.grad.numpy(), 0.0, atol=1e-5)
        assert np.allclose(b.grad.numpy(), -0.5, atol=1e-5)
        assert np.allclose(c.grad.numpy(), 0.0, atol=1e-5)


@pytest.mark.parametrize("fn", VARIANTS)
def test_ignore_grad_correction_is_first_order(fn):
    torch.manual_seed(5)
    a = torch.randn(4, requires_grad=True)
    x = torch.randn(4, requires_grad=True)
    fo = lambda po, pi: (po[0] * pi[0]).sum() + (po[0] ** 2).sum()
    fi = lambda po, pi: ((pi[0] - po[0]) ** 2).sum()
    fn(fo, fi, (a,), (x,), CPU, ignore_grad_correction=True)
    assert torch.allclose(a.grad, x.detach() + 2 * a.detach())
    assert torch.allclose(x.grad, a.detach())


def _gp_closures(g):
    d = g["W"].shape[0]
    Xs, Xq = torch.tensor(g["X_s"]).double(), torch.tensor(g["X_q"]).double()
    ys, yq = torch.tensor(g["y_s"]).double(), torch.tensor(g["y_q"]).double()
    pri = O.Priors(*g["priors"].tolist())
    kind = int(g["kind"])
    f_in = lambda po, pi: O.f_inner(Xs @ po[0] / math.sqrt(d), ys, pi[0], pri, kind)
    f_out = lambda po, pi: O.f_outer(Xs @ po[0] / math.sqrt(d), ys, Xq @ po[0] / math.sqrt(d), yq, pi[0], kind)
    return f_out, f_in


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("fn", VARIANTS)
def test_gp_linear_map_matches_reference_fixture(golden_dir, fn, kind):
    """theta = W; expected W.grad was produced by the reference's cauchy_hypergradient AND ..._jvp (make_golden.py)."""
    g = np.load(os.path.join(golden_dir, f"linmap_N16_Nq24_d12_k{kind}.npz"))
    f_out, f_in = _gp_closures(g)
    W = torch.tensor(g["W"]).double().requires_grad_(True)
    phi = torch.tensor(g["phi"]).double().requires_grad_(True)
    val = fn(f_out, f_in, (W,), (phi,), CPU)
    assert abs(val.item() - float(g["f_out"])) <= 1e-9 * abs(float(g["f_out"]))
    for key in ("grad_W_dense", "grad_W_jvp"):
        assert np.abs(W.grad.numpy() - g[key]).max() <= 1e-7 * np.abs(g[key]).max(), key
    assert np.abs(phi.grad.numpy() - g["grad_phi"]).max() <= 1e-8 * np.abs(g["grad_phi"]).max()
    W.grad = None
    phi.grad = None
    fn(f_out, f_in, (W,), (phi,), CPU, ignore_grad_correction=True)
    assert np.abs(W.grad.numpy() - g["grad_W_first_order"]).max() <= 1e-9 * np.abs(g["grad_W_first_order"]).max()


REF = "/root/reference/fs_mol/utils"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not mounted (GPU box)")
def test_live_cross_check_against_reference_files(golden_dir):
    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    ref = load("cauchy_hypergradient").cauchy_hypergradient
    g = np.load(os.path.join(golden_dir, "linmap_N16_Nq24_d12_k1.npz"))
    f_out, f_in = _gp_closures(g)
    torch.manual_seed(0)
    res = {}
    for name, fn in (("ref", ref), ("ours", cauchy_hypergradient), ("ours_jvp", cauchy_hypergradient_jvp)):
        W = torch.tensor(g["W"]).double().requires_grad_(True)
        phi = (torch.tensor(g["phi"]).double() + 0.05).requires_grad_(True)   # off the optimum on purpose
        fn(f_out, f_in, (W,), (phi,), CPU)
        res[name] = W.grad.clone()
    scale = res["ref"].abs().max().item()
    assert (res["ref"] - res["ours"]).abs().max().item() <= 1e-6 * scale  # the reference stores H and the mixed Jacobian in float32 (torch.zeros default)
    assert (res["ref"] - res["ours_jvp"]).abs().max().item() <= 1e-6 * scale  # the reference stores H and the mixed Jacobian in float32 (torch.zeros default)
