"""Attribute-freeze mixin (mirrors /root/reference/bev/frozen_class.py:1-10).

After `_freeze()` assigning to a name the instance does not yet have raises
`TypeError("<obj> is a frozen class")`; existing attributes stay writable.  As in the
reference, `self.__dict__.update(...)` bypasses the guard (calib.py:52, bev.py:27).
"""


class FrozenClass(object):
    _FrozenClass__isfrozen = False

    def __setattr__(self, name, value):
        if self._FrozenClass__isfrozen and not hasattr(self, name):
            raise TypeError("%r is a frozen class" % self)
        super().__setattr__(name, value)

    def _freeze(self):
        super().__setattr__("_FrozenClass__isfrozen", True)
