"""Minimal Box / Dict spaces (gym is not installed here); the real gym.spaces are used when importable."""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box, Dict  # type: ignore
except Exception:  # noqa: BLE001
    class Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return 'Box(%s, %s)' % (self.low, self.high)

    class Dict:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, k):
            return self.spaces[k]

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}
