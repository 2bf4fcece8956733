"""Growable NumPy-backed list, the reference's `NpList` interface (nplist.py:4-69): append / extend / truncate
(`set_n_elements`) / copy / comparison, contents as a NumPy view.

The device path does not need it (output slots come from a count pass + prefix sums); it exists because it is part
of the reference's importable surface and `DenseKmerFinder` exposes such lists.  Storage is a power-of-two buffer
that is re-allocated only when a write does not fit."""
import numpy as np


def _capacity_for(n):
    cap = 64
    while cap < n:
        cap *= 2
    return cap


class NpList:
    def __init__(self, dtype=None):
        self._dtype = dtype
        self._buf = None if dtype is None else np.empty(64, dtype=dtype)
        self._size = 0

    # the reference's attribute names, read by its callers
    @property
    def _n_elements(self):
        return self._size

    @property
    def _data(self):
        return self._buf if self._buf is not None else np.empty(0)

    def _room_for(self, extra, like):
        if self._buf is None:                               # dtype from the first value, as the reference infers it
            if self._dtype is None:
                self._dtype = np.asarray(like).dtype
            self._buf = np.empty(_capacity_for(extra), dtype=self._dtype)
        need = self._size + extra
        if need > len(self._buf):
            bigger = np.empty(_capacity_for(need), dtype=self._buf.dtype)
            bigger[:self._size] = self._buf[:self._size]
            self._buf = bigger

    def append(self, element):
        self._room_for(1, element)
        self._buf[self._size] = element
        self._size += 1

    def extend(self, elements):
        values = np.asarray(elements)
        if values.size == 0:
            return
        self._room_for(values.size, values.ravel()[0])
        self._buf[self._size:self._size + values.size] = values
        self._size += values.size

    def set_n_elements(self, n):
        """Truncate (or re-expose) to n elements, like the reference: the buffer is not touched."""
        self._size = int(n)

    def get_nparray(self):
        if self._buf is None:
            return np.empty(0) if self._dtype is None else np.empty(0, dtype=self._dtype)
        return self._buf[:self._size]

    def __getitem__(self, item):
        return self.get_nparray()[item]

    def __len__(self):
        return self._size

    def copy(self):
        twin = NpList(dtype=self._dtype)
        twin.extend(self.get_nparray())
        return twin

    def __eq__(self, other):
        mine, theirs = self.get_nparray(), other.get_nparray()
        return len(mine) == len(theirs) and bool(np.all(mine == theirs))

    def __str__(self):
        return str(self.get_nparray())

    def __repr__(self):
        return "NpList(%s)" % self
