"""Synthetic-sequence generator shared by tests and bench.py.

SURVEY.md section 8(d), config 3/4: `std::mt19937 g(12345)`; seq1[i] = "ACGU"[g() & 3]
for i < n, then seq2 likewise from the same generator.  A small pure-Python
MT19937 (32-bit, init_genrand seeding == std::mt19937(seed)) keeps the stream
identical to the C++ one without depending on numpy internals.
"""


class MT19937:
    def __init__(self, seed):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
            v = mt[(i + 397) % 624] ^ (y >> 1)
            if y & 1:
                v ^= 0x9908B0DF
            mt[i] = v
        self.idx = 0

    def __call__(self):
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def random_pair(n1, n2=None, seed=12345):
    """The (seq1, seq2) pair of SURVEY.md section 8(d): one generator, seq1 drawn first."""
    if n2 is None:
        n2 = n1
    g = MT19937(seed)
    s1 = "".join("ACGU"[g() & 3] for _ in range(n1))
    s2 = "".join("ACGU"[g() & 3] for _ in range(n2))
    return s1, s2


def random_pairs(count, n1, n2=None, seed=12345):
    """`count` consecutive pairs from one generator (batch workloads)."""
    if n2 is None:
        n2 = n1
    g = MT19937(seed)
    out = []
    for _ in range(count):
        s1 = "".join("ACGU"[g() & 3] for _ in range(n1))
        s2 = "".join("ACGU"[g() & 3] for _ in range(n2))
        out.append((s1, s2))
    return out
