"""Growable NumPy-backed list with the reference's semantics (nplist.py:4-69).

The device path does not need it (output slots come from a count pass + prefix sums); it is kept
because it is part of the reference's importable surface."""
import numpy as np


class NpList:
    def __init__(self, dtype=None):
        self._dtype = dtype
        self._data = np.empty(0, dtype=dtype) if dtype is not None else np.empty(0)
        self._n_elements = 0

    def _reserve(self, n):
        if n <= len(self._data):
            return
        grown = np.zeros(n, dtype=self._data.dtype)
        grown[:self._n_elements] = self._data[:self._n_elements]
        self._data = grown

    def append(self, element):
        if len(self._data) == 0:
            if self._dtype is None:
                self._dtype = type(element)
            self._data = np.zeros(100, dtype=self._dtype)
        if self._n_elements == len(self._data):
            self._reserve(int(len(self._data) * 1.5))
        self._data[self._n_elements] = element
        self._n_elements += 1

    def extend(self, elements):
        m = len(elements)
        if self._n_elements + m >= len(self._data):
            self._reserve((self._n_elements + m) * 2)
        self._data[self._n_elements:self._n_elements + m] = elements
        self._n_elements += m

    def get_nparray(self):
        return self._data[:self._n_elements]

    def __getitem__(self, item):
        return self.get_nparray()[item]

    def set_n_elements(self, n):
        self._n_elements = n

    def copy(self):
        new = NpList(dtype=self._dtype)
        new.extend(self.get_nparray())
        return new

    def __eq__(self, other):
        return bool(np.all(self.get_nparray() == other.get_nparray()))

    def __len__(self):
        return self._n_elements

    def __str__(self):
        return str(self.get_nparray())

    def __repr__(self):
        return "NpList(" + str(self) + ")"
