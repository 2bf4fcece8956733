import numpy as np


class HashTable:
    """TEST-ONLY stand-in: values of a key come back in insertion order."""

    def __init__(self, keys, values, mod=None, **kw):
        self._keys = np.asarray(keys)
        self._values = np.asarray(values) if not np.isscalar(values) else np.full(len(self._keys), values)
        self.dtype = self._values.dtype
        self._map = {}
        for i, k in enumerate(self._keys.tolist()):
            self._map.setdefault(k, []).append(i)

    def __getitem__(self, keys):
        if np.isscalar(keys):
            return self._values[self._map.get(int(keys), [])]
        idx = [j for k in np.asarray(keys).tolist() for j in self._map.get(k, [])]
        return self._values[idx]

    def __contains__(self, key):
        return int(key) in self._map


class Counter(HashTable):
    def __init__(self, keys, values=0, mod=None, value_dtype=None, **kw):
        keys = np.asarray(keys)
        v = np.full(len(keys), values, dtype=value_dtype or np.int64) if np.isscalar(values) else values
        super().__init__(keys, v)

    def count(self, samples):
        for s in np.asarray(samples).tolist():
            if s in self._map:
                self._values[self._map[s][0]] += 1
