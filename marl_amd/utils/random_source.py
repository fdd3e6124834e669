"""Where the learners get their random numbers.

The reference draws from numpy's and torch's global generators in the middle of its updates: `np.random.choice` for
expert samples (gail_agent.py:90,139; dgail_agent.py:158,214,428,522), `th.randint` / `th.randn_like` inside
`DiffusionDiscriminator.compute_loss` (dgail_network.py:90-91) and the DataLoader's shuffle (dgail_agent.py:290).  Here
every such draw goes through one object, on the device and without host round trips; the parity tests swap in a source
that replays what the reference drew (tests/test_learners_*.py), which makes an update a function of its inputs.
"""
import torch as th


class RandomSource:
    def __init__(self, device, generator=None):
        self.device = th.device(device)
        self.generator = generator

    def choice_without_replacement(self, size, n):
        """n distinct indices below `size`, uniformly, in random order (np.random.choice(size, n, replace=False))"""
        if n * 4 <= size:
            # a uniformly random n-subset in random order = the n largest of `size` i.i.d. keys: two small kernels
            # instead of the sort behind randperm (a dozen launches, once per minibatch step of the learners)
            return th.rand(size, device=self.device, generator=self.generator).topk(n).indices
        return th.randperm(size, device=self.device, generator=self.generator)[:n]

    def choice_without_replacement_many(self, size, n, count):
        """`count` consecutive choice_without_replacement(size, n) draws as one [count, n] tensor (one batched top-k:
        the minibatch steps of an epoch take their expert samples from it without a launch sequence per step)"""
        if n * 4 <= size:
            return th.rand(count, size, device=self.device, generator=self.generator).topk(n, dim=1).indices
        return th.stack([self.choice_without_replacement(size, n) for _ in range(count)])

    def choice_with_replacement(self, size, n):
        """np.random.choice(size, n, replace=True)"""
        return th.randint(0, size, (int(n),), device=self.device, generator=self.generator)

    def permutation(self, n):
        """the DataLoader's shuffle: a permutation of range(n)"""
        return th.randperm(n, device=self.device, generator=self.generator)

    def diffusion_draws(self, rows, dim):
        """(t, noise) of one compute_loss call, in the reference's order (dgail_network.py:90-91)"""
        t = th.randint(0, 2, (rows,), device=self.device, generator=self.generator)
        noise = th.randn(rows, dim, device=self.device, generator=self.generator)
        return t, noise


class ReplaySource(RandomSource):
    """Replays recorded draws in call order (parity tests): `draws` is a list of (kind, value) with kind in
    {"choice", "perm", "t", "noise"}; every call checks that the kind and size it asks for is the next one recorded."""

    def __init__(self, device, draws):
        super().__init__(device)
        self.draws = list(draws)
        self.pos = 0

    def _next(self, kind, n=None):
        assert self.pos < len(self.draws), "replay exhausted at a %s draw" % kind
        k, v = self.draws[self.pos]
        self.pos += 1
        assert k == kind, "draw %d is a %s, the learner asked for a %s" % (self.pos - 1, k, kind)
        v = th.as_tensor(v)
        assert n is None or v.shape[0] == n, "draw %d has %d entries, the learner asked for %d" % (self.pos - 1, v.shape[0], n)
        return v.to(self.device)

    def choice_without_replacement(self, size, n):
        return self._next("choice", n).long()

    def choice_without_replacement_many(self, size, n, count):
        return th.stack([self._next("choice", n).long() for _ in range(count)])

    def choice_with_replacement(self, size, n):
        return self._next("choice", n).long()

    def permutation(self, n):
        return self._next("perm", n).long()

    def diffusion_draws(self, rows, dim):
        return self._next("t", rows).long(), self._next("noise", rows).float()

    def exhausted(self):
        return self.pos == len(self.draws)
