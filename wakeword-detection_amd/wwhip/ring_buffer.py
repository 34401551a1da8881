"""Fixed-capacity circular buffer with the observable behaviour of the reference's
``RingBuffer`` (``spokestack/ring_buffer.py:9-130``): one spare slot tells full from empty,
``rewind`` treats the buffer as full, ``read`` hands out a 1-item view, ``read_all`` rewinds
and returns everything oldest-first, overflow/underflow raise ``IndexError``.

This is host bookkeeping for callers that want the reference's buffer object; the HIP path
keeps its rings in HBM (``csrc/streams.hip``).  Unlike the reference the storage is
addressed with two slices instead of one Python-level read per item, and block writes are
available (``write_block``), so the 320-calls-per-20-ms loop of the reference is not needed.
"""
from __future__ import annotations

from typing import Sequence, Union

import numpy as np


class RingBuffer:
    def __init__(self, shape: Sequence[int], dtype=np.float32) -> None:
        # The reference bumps shape[0] in the caller's list (ring_buffer.py:15); callers never
        # reuse that list, and mutating an argument is a hazard, so the list is left alone here.
        dims = list(shape)
        if not dims or dims[0] < 0:
            raise ValueError("shape must start with a non-negative capacity")
        self._dtype = dtype
        self._slots = dims[0] + 1
        self._store = np.empty([self._slots] + dims[1:], dtype=dtype)
        self._head = 0  # next slot to read
        self._tail = 0  # next slot to write

    # ---- state -----------------------------------------------------------------------------
    @property
    def capacity(self) -> int:
        return self._slots - 1

    @property
    def is_empty(self) -> bool:
        return self._head == self._tail

    @property
    def is_full(self) -> bool:
        return (self._tail + 1) % self._slots == self._head

    def __len__(self) -> int:
        return (self._tail - self._head) % self._slots

    # ---- cursor moves (all return self, like the reference) ----------------------------------
    def rewind(self) -> "RingBuffer":
        self._head = (self._tail + 1) % self._slots
        return self

    def seek(self, steps: int) -> "RingBuffer":
        self._head = (self._head + steps) % self._slots
        return self

    def reset(self) -> "RingBuffer":
        self._tail = self._head
        return self

    def fill(self, value: Union[int, float]) -> "RingBuffer":
        self._store.fill(value)
        return self.rewind()

    # ---- data ----------------------------------------------------------------------------------
    def write(self, item) -> None:
        if self.is_full:
            raise IndexError("Buffer is full")
        self._store[self._tail] = item
        self._tail = (self._tail + 1) % self._slots

    def write_block(self, items: np.ndarray) -> None:
        """Append ``len(items)`` items at once (not in the reference)."""
        n = len(items)
        if n > self.capacity - len(self):
            raise IndexError("Buffer is full")
        first = min(n, self._slots - self._tail)
        self._store[self._tail:self._tail + first] = items[:first]
        if n > first:
            self._store[: n - first] = items[first:]
        self._tail = (self._tail + n) % self._slots

    def read(self) -> np.ndarray:
        if self.is_empty:
            raise IndexError("Buffer is empty")
        view = self._store[self._head:self._head + 1]
        self._head = (self._head + 1) % self._slots
        return view

    def read_all(self) -> np.ndarray:
        self.rewind()
        if self.is_empty:  # zero-capacity ring: the reference fails inside np.concatenate
            raise ValueError("need at least one array to concatenate")
        n = len(self)
        first = min(n, self._slots - self._head)
        out = np.concatenate((self._store[self._head:self._head + first], self._store[: n - first]))
        self._head = self._tail
        return out.astype(self._dtype)
