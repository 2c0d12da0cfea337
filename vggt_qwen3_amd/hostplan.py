"""Host-side facts about a batch's integer inputs (where the <image> tokens sit, which rows carry a label, how many key tiles are
attended): each costs a device -> host read, i.e. a wait for everything queued before it. They are pure functions of tensors the
training loop passes in unchanged for a whole accumulation window or benchmark, so they are remembered per tensor OBJECT: an entry
is valid only while the very same torch.Tensor is alive (weak reference) and has not been written to (`_version`)."""
import weakref
from typing import Callable, Sequence

import torch


class HostPlan:
    def __init__(self, capacity: int = 32):
        self._d = {}
        self._cap = capacity

    def get(self, what, tensors: Sequence[torch.Tensor], fn: Callable):
        key = (what,) + tuple(id(t) for t in tensors)
        e = self._d.get(key)
        if e is not None:
            refs, vers, val = e
            if all(r() is t and v == t._version for r, v, t in zip(refs, vers, tensors)):
                return val
        val = fn()
        if len(self._d) >= self._cap:
            self._d.clear()
        try:
            self._d[key] = ([weakref.ref(t) for t in tensors], [t._version for t in tensors], val)
        except TypeError:      # an object without weak-reference support: simply do not remember
            pass
        return val


PLAN = HostPlan()
