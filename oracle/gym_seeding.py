"""Restatement of gym==0.11.0 ``gym.utils.seeding.np_random`` (third-party,
pinned at /root/reference/environment.yml:101; called from
/root/reference/environments/srl_env.py:77).

Published algorithm: seed -> sha512(str(seed)) -> first 8 bytes -> little
endian uint32 digits -> ``numpy.random.RandomState().seed(digits)``
(MT19937 ``init_by_array``).  TEST INFRASTRUCTURE ONLY (see oracle/__init__).
"""
import hashlib
import struct

import numpy as np


def hash_seed_digits(seed):
    """Key (list of uint32) handed to RandomState.seed for ``env.seed(seed)``."""
    if seed is None or not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError("Seed must be a non-negative integer")
    seed = int(seed) % 2 ** 64
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()[:8]
    digest += b"\0" * (4 - len(digest) % 4)   # gym pads a full word when aligned
    words = struct.unpack("{}I".format(len(digest) // 4), digest)
    big = sum(w << (32 * i) for i, w in enumerate(words))
    digits = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        digits.append(mod)
    return digits


def np_random(seed):
    """-> (RandomState, seed) exactly like gym.utils.seeding.np_random."""
    rng = np.random.RandomState()
    rng.seed(hash_seed_digits(seed))
    return rng, int(seed) % 2 ** 64
