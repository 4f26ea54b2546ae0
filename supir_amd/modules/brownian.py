"""Brownian-tree noise for the DPM++ 2M SDE samplers (config 5; reference call sites: sgm/modules/diffusionmodules/sampling.py:491-494,
684-687 -- `BrownianTreeNoiseSampler(x, sigmas_min, sigmas_max)`, queried as `noise_sampler(s_in * sigmas[i], s_in * sigmas[i + 1])`).

The class lives in a third-party package that is not in the reference tree and not installable here: k-diffusion 0.1.1.post1
(requirements.txt:41) `BrownianTreeNoiseSampler` -> `BatchedBrownianTree` -> `torchsde.BrownianTree`, i.e. torchsde's `BrownianInterval`
with `halfway_tree=True, tol=1e-6, pool_size=24`.  This module restates the PUBLISHED algorithm those implement -- the virtual Brownian
tree of Li, Wong, Chen, Duvenaud, "Scalable Gradients for Stochastic Differential Equations" (AISTATS 2020), section 4 / algorithm 3 --
behind k-diffusion's interface:

  * the Wiener increment over the whole interval [t0, t1] is one normal draw; every node [a, b] of the dyadic tree below it draws its
    midpoint by the Brownian bridge   W(a, m) = W(a, b) / 2 + sqrt((b - a) / 4) z,   W(m, b) = W(a, b) - W(a, m);
  * the draws come from a SPLITTABLE seed: a node's generator is a function of (entropy, path from the root) only
    (`numpy.random.SeedSequence` with the path as spawn key -- the primitive torchsde uses), so no draw depends on the order of the
    queries and any node can be rebuilt after it fell out of the cache;
  * a query descends to nodes narrower than `tol` and interpolates linearly inside the leaf (error variance <= tol / 4).

So the capability is the reference's: one Brownian path per seed, increments over arbitrary (overlapping, repeated, re-ordered) intervals
consistent with it -- W(a, c) = W(a, b) + W(b, c) --, normalised increments N(0, 1), and with `seed=None` ONE draw from torch's global
generator (what `seed_everything` seeds), as in k-diffusion.  What is NOT pinned is bit equality with torchsde's stream: its exact
spawn-key bookkeeping cannot be checked without the package ("parity unpinned" for config 5's noise stream; the
solver arithmetic is pinned with an injected sampler).
"""
import collections
import math

import numpy as np
import torch


class BrownianTree:
    """One Brownian path W on [t0, t1] with values of a given shape; `tree(ta, tb)` = W(tb) - W(ta)."""

    def __init__(self, t0, t1, shape, dtype, device, entropy, tol=1e-6, pool_size=24, cache_size=64):
        t0, t1 = float(t0), float(t1)
        if not t0 < t1:
            raise ValueError(f"BrownianTree needs t0 < t1, got {t0}, {t1}")
        self.t0, self.t1, self.tol = t0, t1, float(tol)
        self.shape, self.dtype, self.device = tuple(shape), dtype, torch.device(device)
        self.entropy, self.pool_size = int(entropy), int(pool_size)
        self._half = collections.OrderedDict()     # path -> W(a, m) of that node, LRU
        self._at = collections.OrderedDict()       # t -> W(t) - W(t0), LRU (consecutive sampler steps share an end point)
        self._cache_size = int(cache_size)
        self._root = self._normal((3,)) * math.sqrt(t1 - t0)

    def _normal(self, key):
        """Standard normal of the tree's shape whose generator is a function of (entropy, key) only."""
        ss = np.random.SeedSequence(entropy=self.entropy, spawn_key=tuple(key), pool_size=self.pool_size)
        seed = int(ss.generate_state(1, dtype=np.uint64)[0]) & (2 ** 63 - 1)
        gen = torch.Generator(device=self.device).manual_seed(seed)
        return torch.randn(self.shape, dtype=self.dtype, device=self.device, generator=gen)

    def _left_half(self, path, a, b, w_ab):
        """W(a, m) of the node `path` = [a, b] with increment w_ab: the Brownian bridge at the midpoint."""
        w = self._half.get(path)
        if w is None:
            w = self._normal(path + (2,)).mul_(0.5 * math.sqrt(b - a)).add_(w_ab, alpha=0.5)
            self._half[path] = w
            if len(self._half) > self._cache_size:
                self._half.popitem(last=False)
        else:
            self._half.move_to_end(path)
        return w

    def value(self, t):
        """W(t) - W(t0); t is clamped to [t0, t1]."""
        t = min(max(float(t), self.t0), self.t1)
        hit = self._at.get(t)
        if hit is not None:
            self._at.move_to_end(t)
            return hit
        a, b, w_ab, path = self.t0, self.t1, self._root, ()
        acc = torch.zeros(self.shape, dtype=self.dtype, device=self.device)
        while True:
            if t <= a:
                out = acc
                break
            if t >= b:
                out = acc.add_(w_ab)
                break
            if b - a <= self.tol:
                out = acc.add_(w_ab, alpha=(t - a) / (b - a))
                break
            m = 0.5 * (a + b)
            w_am = self._left_half(path, a, b, w_ab)
            if t < m:
                b, w_ab, path = m, w_am, path + (0,)
            else:
                acc.add_(w_am)                     # acc is private to this query (cached tensors are never written)
                a, w_ab, path = m, w_ab - w_am, path + (1,)
        self._at[t] = out
        if len(self._at) > 8:
            self._at.popitem(last=False)
        return out

    def __call__(self, ta, tb):
        return self.value(tb) - self.value(ta)


def _scalar(t):
    """A time handed over as python number, 0-dim tensor or [B] tensor whose entries agree (the reference passes s_in * sigma)."""
    if torch.is_tensor(t):
        vals = t.detach().reshape(-1).tolist()        # one device -> host copy (the reference's torchsde call does the same)
        if any(v != vals[0] for v in vals[1:]):
            raise ValueError("BrownianTreeNoiseSampler: the samples of a batch must share sigma")
        return float(vals[0])
    return float(t)


class BatchedBrownianTree:
    """k-diffusion's wrapper: one tree for the whole batch (seed: int or None) or one per sample (seed: sequence of B ints);
    the sign convention of its `sort`: tree(t0, t1) with t0 > t1 is the negated increment."""

    def __init__(self, x, t0, t1, seed=None, **kwargs):
        t0, t1 = _scalar(t0), _scalar(t1)
        self.sign = 1.0
        if t0 > t1:
            t0, t1, self.sign = t1, t0, -1.0
        if seed is None:
            seed = torch.randint(0, 2 ** 63 - 1, []).item()      # the one draw k-diffusion takes from the global generator
        self.batched = True
        try:
            seeds = [int(s) for s in seed]
            assert len(seeds) == x.shape[0]
            shape = x.shape[1:]
        except TypeError:
            seeds, shape, self.batched = [int(seed)], x.shape, False
        self.trees = [BrownianTree(t0, t1, shape, x.dtype, x.device, s, **kwargs) for s in seeds]

    def __call__(self, t0, t1):
        t0, t1 = _scalar(t0), _scalar(t1)
        sign = 1.0
        if t0 > t1:
            t0, t1, sign = t1, t0, -1.0
        w = torch.stack([tree(t0, t1) for tree in self.trees]) * (self.sign * sign)
        return w if self.batched else w[0]


class BrownianTreeNoiseSampler:
    """k_diffusion.sampling.BrownianTreeNoiseSampler: `sampler(sigma, sigma_next)` = the path's increment between the two (transformed)
    noise levels, normalised to unit variance.  Constructor and call signatures are k-diffusion's, so the class drops into
    RestoreDPMPP2MSampler / TiledRestoreDPMPP2MSampler where the reference constructs it (sampling.py:494, 687)."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda x: x):
        self.transform = transform
        t0, t1 = _scalar(self.transform(torch.as_tensor(sigma_min))), _scalar(self.transform(torch.as_tensor(sigma_max)))
        self.tree = BatchedBrownianTree(x, t0, t1, seed)

    def __call__(self, sigma, sigma_next):
        t0, t1 = _scalar(self.transform(torch.as_tensor(sigma))), _scalar(self.transform(torch.as_tensor(sigma_next)))
        return self.tree(t0, t1) / math.sqrt(abs(t1 - t0))
