"""Brownian-tree noise of the DPM++ 2M SDE samplers (supir_amd/modules/brownian.py; reference call sites
sgm/modules/diffusionmodules/sampling.py:491-494, 684-687).  k-diffusion / torchsde are not installable here, so what is held is what the
published algorithm guarantees and the reference relies on: one path per seed, increments consistent over arbitrary intervals and query
orders, the sign convention and RNG consumption of k-diffusion's wrapper, unit-variance normalised increments that are independent on
disjoint intervals, and the samplers constructing the class the way the reference does."""
import math

import pytest
import torch

from supir_amd.modules import brownian as B
from supir_amd.modules import sampling as S

SMIN, SMAX = 0.0292, 14.6146      # the DDPM schedule's end points the reference hands over (sampling.py:490)


def _x(n=1, c=4, h=16, w=16, device="cpu"):
    return torch.zeros(n, c, h, w, device=device)


def test_one_path_per_seed_any_query_order():
    sig = [14.6146, 9.7, 4.1, 1.3, 0.4, 0.11, 0.0292]
    a = B.BrownianTreeNoiseSampler(_x(), SMIN, SMAX, seed=1234)
    fwd = [a(torch.tensor([s0]), torch.tensor([s1])) for s0, s1 in zip(sig[:-1], sig[1:])]
    b = B.BrownianTreeNoiseSampler(_x(), SMIN, SMAX, seed=1234)
    bwd = [b(torch.tensor([s0]), torch.tensor([s1])) for s0, s1 in reversed(list(zip(sig[:-1], sig[1:])))][::-1]
    for u, v in zip(fwd, bwd):
        assert torch.equal(u, v)                                  # no draw depends on what was asked before
    c = B.BrownianTreeNoiseSampler(_x(), SMIN, SMAX, seed=1235)
    assert not torch.equal(c(sig[0], sig[1]), fwd[0])
    # a tree with a tiny node cache rebuilds evicted nodes from (entropy, path): same values
    t_small = B.BrownianTree(SMIN, SMAX, (4, 16, 16), torch.float32, "cpu", 1234, cache_size=2)
    t_big = B.BrownianTree(SMIN, SMAX, (4, 16, 16), torch.float32, "cpu", 1234, cache_size=4096)
    for t in (0.5, 7.0, 0.5, 13.2, 0.03, 7.0):
        assert torch.equal(t_small.value(t), t_big.value(t))


def test_increments_are_consistent():
    tree = B.BrownianTree(SMIN, SMAX, (4, 32, 32), torch.float32, "cpu", 7)
    a, b, c = 0.31, 2.9, 11.4
    whole, parts = tree(a, c), tree(a, b) + tree(b, c)
    assert (whole - parts).abs().max().item() <= 1e-5
    assert torch.equal(tree(SMIN, SMAX), tree._root)             # the end points are the root draw
    assert torch.equal(tree(a, a), torch.zeros(4, 32, 32))
    assert torch.equal(tree.value(-1.0), tree.value(SMIN)) and torch.equal(tree.value(99.0), tree.value(SMAX))   # clamped
    # below the tolerance the path is interpolated inside a leaf: increments stay of the order sqrt(dt) (continuity), never a jump
    t0 = 3.3333333
    assert tree(t0, t0 + 4e-7).abs().max().item() <= 6 * math.sqrt(1e-6)


def test_sign_convention_and_batched_seeds():
    x = _x(n=2)
    s = B.BrownianTreeNoiseSampler(x, SMIN, SMAX, seed=5)
    down, up = s(torch.tensor([4.0, 4.0]), torch.tensor([1.0, 1.0])), s(1.0, 4.0)
    assert down.shape == x.shape and torch.equal(down, -up)       # k-diffusion's sort(): descending sigma = negated increment
    # the raw tree before normalisation: tree(lo, hi) = W(hi) - W(lo)
    assert torch.allclose(s.tree(1.0, 4.0) / math.sqrt(3.0), up)
    # one tree per sample when the seed is a sequence (k-diffusion's `batched` branch): sample i is the single-sample tree of seed i
    sb = B.BrownianTreeNoiseSampler(x, SMIN, SMAX, seed=[11, 12])
    s11 = B.BrownianTreeNoiseSampler(x[:1], SMIN, SMAX, seed=11)
    out = sb(2.0, 0.5)
    assert out.shape == x.shape and torch.equal(out[0], s11(2.0, 0.5)[0]) and not torch.equal(out[0], out[1])
    with pytest.raises(ValueError):
        s(torch.tensor([4.0, 3.0]), torch.tensor([1.0, 1.0]))    # the samples of a batch share sigma (the reference only runs B = 1 here)
    # a sigma range handed over in descending order is the same path negated (BatchedBrownianTree.__init__'s sort)
    r = B.BrownianTreeNoiseSampler(x, SMAX, SMIN, seed=5)
    assert torch.equal(r(4.0, 1.0), -down)


def test_seed_none_takes_one_draw_from_the_global_generator():
    torch.manual_seed(99)
    expect_seed = torch.randint(0, 2 ** 63 - 1, []).item()
    after = torch.rand(3)
    torch.manual_seed(99)
    s = B.BrownianTreeNoiseSampler(_x(), SMIN, SMAX)
    assert torch.equal(torch.rand(3), after)                      # exactly one randint was consumed (k-diffusion: BatchedBrownianTree)
    assert torch.equal(s(3.0, 1.0), B.BrownianTreeNoiseSampler(_x(), SMIN, SMAX, seed=expect_seed)(3.0, 1.0))


def test_normalised_increments_are_standard_normal_and_independent():
    s = B.BrownianTreeNoiseSampler(torch.zeros(1, 4, 256, 256), SMIN, SMAX, seed=2024)
    sig = [14.6146, 6.0, 2.2, 0.7, 0.2, 0.0292]
    inc = [s(a, b).flatten().double() for a, b in zip(sig[:-1], sig[1:])]
    n = inc[0].numel()
    for v in inc:
        assert abs(v.mean().item()) <= 5 / math.sqrt(n) and abs(v.var().item() - 1.0) <= 5 * math.sqrt(2.0 / n)
        assert abs((v ** 4).mean().item() - 3.0) <= 0.15          # kurtosis of a normal
    for i in range(len(inc)):
        for j in range(i + 1, len(inc)):
            assert abs((inc[i] * inc[j]).mean().item()) <= 5 / math.sqrt(n)     # disjoint intervals: uncorrelated
    # an interval and a sub-interval of it are correlated exactly as a Wiener process prescribes: corr = sqrt(|sub| / |whole|)
    whole, sub = s(6.0, 0.7).flatten().double(), s(2.2, 0.7).flatten().double()
    assert abs((whole * sub).mean().item() - math.sqrt((2.2 - 0.7) / (6.0 - 0.7))) <= 5 / math.sqrt(n)


def test_the_samplers_construct_it_as_the_reference_does():
    """sampling.py:494 / :687: BrownianTreeNoiseSampler(x, sigmas_min, sigmas_max), seed from the global generator, queried with
    s_in * sigmas[i] pairs on the steps 0 < i < last -- two runs under the same torch seed draw the same noise, another seed another."""
    calls = []

    class Spy(B.BrownianTreeNoiseSampler):
        def __call__(self, sigma, sigma_next):
            out = super().__call__(sigma, sigma_next)
            calls.append((float(sigma[0]), float(sigma_next[0]), out.clone()))
            return out

    def run(seed, cls):
        smp = S.RestoreDPMPP2MSampler(num_steps=4, s_noise=1.003, eta=1.0, device="cpu", guider_config=S.LinearCFG(1.0, 4.0),
                                      noise_sampler_cls=cls)
        den = S.DiscreteDenoiserWithControl()
        net = lambda x, t, c, cs: 0.1 * x + c["control"].mean()  # noqa: E731
        torch.manual_seed(seed)
        x = torch.randn(1, 4, 8, 8)
        c = {"crossattn": torch.zeros(1, 2, 4), "vector": torch.zeros(1, 4), "control": torch.ones(1, 4, 8, 8)}
        return smp(lambda i, s, cc, cs: den(net, i, s, cc, cs), x, cond=c, uc=dict(c), control_scale=1.0)

    default_cls = S.RestoreDPMPP2MSampler(num_steps=4, device="cpu", guider_config=S.LinearCFG(1.0, 4.0)).noise_sampler_cls
    assert default_cls is B.BrownianTreeNoiseSampler
    a = run(3, Spy)
    first = list(calls)
    calls.clear()
    b = run(3, Spy)
    assert torch.equal(a, b) and len(first) == len(calls) == 2 and all(torch.equal(u[2], v[2]) for u, v in zip(first, calls))
    assert all(s0 > s1 > 0 for s0, s1, _ in first)                # the interior steps, descending sigma
    calls.clear()
    assert not torch.equal(run(4, Spy), a)


@pytest.mark.gpu
def test_brownian_tree_on_the_device():
    x = torch.zeros(1, 4, 128, 128, device="cuda:0")
    s = B.BrownianTreeNoiseSampler(x, SMIN, SMAX, seed=1234)
    a, b = s(torch.full((1,), 4.0, device="cuda:0"), torch.full((1,), 1.5, device="cuda:0")), s(1.5, 0.3)
    assert a.device == x.device and a.shape == x.shape and a.dtype == x.dtype
    s2 = B.BrownianTreeNoiseSampler(x, SMIN, SMAX, seed=1234)
    assert torch.equal(s2(1.5, 0.3), b) and torch.equal(s2(4.0, 1.5), a)                      # other query order, same path
    whole = s.tree(0.3, 4.0)
    parts = s.tree(0.3, 1.5) + s.tree(1.5, 4.0)
    assert (whole - parts).abs().max().item() <= 1e-5
    v = torch.cat([a.flatten(), b.flatten()]).double()
    assert abs(v.mean().item()) <= 0.03 and abs(v.var().item() - 1.0) <= 0.03


def test_the_real_k_diffusion_class_is_preferred_when_both_packages_import(monkeypatch):
    """VERDICT r05 item 7a.  A user with the reference's requirements installed (k-diffusion 0.1.1.post1 + torchsde, requirements.txt:41)
    must get the reference's exact noise stream: the DPM++ samplers then construct `k_diffusion.sampling.BrownianTreeNoiseSampler` itself
    (sampling.py:494, 687).  Neither package exists in this image, so fake modules stand in to prove the ORDER of preference: both import ->
    theirs; torchsde missing (k_diffusion present but unusable) -> the restatement; the inert stub oracle/ref_import.py plants
    (`BrownianTreeNoiseSampler = None`) -> the restatement; SUPIR_BROWNIAN=native -> the restatement; an explicit noise_sampler_cls wins."""
    import sys
    import types
    from supir_amd.modules import sampling as S
    from supir_amd.modules.brownian import BrownianTreeNoiseSampler as Ours

    class Theirs:
        def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda t: t):
            self.args = (tuple(x.shape), float(sigma_min), float(sigma_max))

        def __call__(self, sigma, sigma_next):
            raise AssertionError("not queried in this test")

    for gone in ("k_diffusion", "k_diffusion.sampling", "torchsde"):
        monkeypatch.delitem(sys.modules, gone, raising=False)
    monkeypatch.delenv("SUPIR_BROWNIAN", raising=False)
    # whether or not the real packages are installed, control what `import` finds: absent first
    monkeypatch.setitem(sys.modules, "torchsde", None)              # `import torchsde` raises ImportError
    assert S.default_noise_sampler_cls() is Ours
    kd, kds = types.ModuleType("k_diffusion"), types.ModuleType("k_diffusion.sampling")
    kds.BrownianTreeNoiseSampler = Theirs
    kd.sampling = kds
    monkeypatch.setitem(sys.modules, "k_diffusion", kd)
    monkeypatch.setitem(sys.modules, "k_diffusion.sampling", kds)
    assert S.default_noise_sampler_cls() is Ours                    # k_diffusion alone cannot run its tree
    monkeypatch.setitem(sys.modules, "torchsde", types.ModuleType("torchsde"))
    assert S.default_noise_sampler_cls() is Theirs
    smp = S.RestoreDPMPP2MSampler(num_steps=4, s_noise=1.003, eta=1.0, device="cpu", guider_config=S.LinearCFG(1.0, 4.0),
                                  discretization_config=S.LegacyDDPMDiscretization())
    assert smp.noise_sampler_cls is Theirs
    tiled = S.TiledRestoreDPMPP2MSampler(tile_size=16, tile_stride=8, num_steps=4, s_noise=1.003, eta=1.0, device="cpu",
                                         guider_config=S.LinearCFG(1.0, 4.0), discretization_config=S.LegacyDDPMDiscretization())
    assert tiled.noise_sampler_cls is Theirs
    kds.BrownianTreeNoiseSampler = None                             # the inert stub of oracle/ref_import.py
    assert S.default_noise_sampler_cls() is Ours
    kds.BrownianTreeNoiseSampler = Theirs
    monkeypatch.setenv("SUPIR_BROWNIAN", "native")
    assert S.default_noise_sampler_cls() is Ours
    monkeypatch.delenv("SUPIR_BROWNIAN")
    explicit = S.RestoreDPMPP2MSampler(num_steps=4, device="cpu", noise_sampler_cls=S.IntervalNoiseSampler, guider_config=S.LinearCFG(1.0, 4.0),
                                       discretization_config=S.LegacyDDPMDiscretization())
    assert explicit.noise_sampler_cls is S.IntervalNoiseSampler
