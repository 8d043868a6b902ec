"""Minimal stand-ins for the gym spaces the reference exposes (maenv:362, 398-427).

gym is not a dependency of this package; these carry the same attributes callers read
(`Discrete.n`, `Box.low/high/shape`, `Dict.spaces`) and sample()/contains() for convenience.
"""
import numpy as np


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high = low, high
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low, self.high, self.shape)


class Dict(dict):
    def __init__(self, spaces):
        super().__init__(spaces)
        self.spaces = dict(spaces)
