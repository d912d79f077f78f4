"""Minimal stand-ins for the bits of gym==0.11.0 the env surface needs
(`spaces.Discrete`, `spaces.Box`, `utils.seeding.np_random`, `Env`).  The real
gym is used when it is importable; it is not installed in the build image."""
import hashlib
import struct

import numpy as np

try:                                        # pragma: no cover - gym is absent in this image
    import gym as _gym
    from gym import spaces as _spaces
    HAVE_GYM = True
except Exception:                           # noqa: BLE001
    _gym, _spaces, HAVE_GYM = None, None, False


def hash_seed_digits(seed):
    """gym.utils.seeding: sha512(str(seed))[:8] -> little-endian uint32 digits."""
    seed = int(seed) % 2 ** 64
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    digest += b"\0" * (4 - len(digest) % 4)
    words = struct.unpack("{}I".format(len(digest) // 4), digest)
    big = sum(w << (32 * i) for i, w in enumerate(words))
    digits = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        digits.append(mod)
    return digits


def np_random(seed=None):
    """gym.utils.seeding.np_random: -> (RandomState, seed)."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    if seed is None:
        seed = int.from_bytes(np.random.bytes(8), "little")
    seed = int(seed) % 2 ** 64
    rng = np.random.RandomState()
    rng.seed(hash_seed_digits(seed))
    return rng, seed


if HAVE_GYM:                                # pragma: no cover
    Env, Discrete, Box = _gym.Env, _spaces.Discrete, _spaces.Box
else:
    class Env(object):
        metadata = {"render.modes": []}
        reward_range = (-float("inf"), float("inf"))
        spec = None
        action_space = None
        observation_space = None

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass

    class _Space(object):
        def __init__(self, shape, dtype):
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)
            self.np_random = np.random.RandomState()

        def seed(self, seed):
            self.np_random.seed(seed)

    class Discrete(_Space):
        def __init__(self, n):
            super(Discrete, self).__init__((), np.int64)
            self.n = n

        def sample(self):
            return self.np_random.randint(self.n)

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class Box(_Space):
        def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
            if shape is None:
                low, high = np.asarray(low), np.asarray(high)
                shape = low.shape
            else:
                low, high = np.full(shape, low), np.full(shape, high)
            super(Box, self).__init__(shape, dtype)
            self.low, self.high = low.astype(dtype), high.astype(dtype)

        def sample(self):
            return self.np_random.uniform(low=self.low, high=self.high, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and (x >= self.low).all() and (x <= self.high).all()

        def __repr__(self):
            return "Box" + str(self.shape)
