"""Small helpers the managers share (reference: utils/misc.py:67-92)."""
import re


class Singleton(type):
    """One instance per class; later constructor calls (with or without arguments) return it.
    The layer code reaches the managers this way: `QMI()`, `self.sm()` (utils/misc.py:67-73)."""
    _instances = {}

    def __call__(cls, *args, **kwargs):
        if cls not in cls._instances:
            cls._instances[cls] = super(Singleton, cls).__call__(*args, **kwargs)
        return cls._instances[cls]

    @classmethod
    def reset(mcs, cls=None):
        """Forget one (or every) singleton instance - tests and repeated experiments."""
        if cls is None:
            mcs._instances.clear()
        else:
            mcs._instances.pop(cls, None)


def sorted_nicely(items):
    """Human ordering: conv2 before conv10 (utils/misc.py:79-92)."""
    def key(s):
        return [int(c) if c.isdigit() else c for c in re.split('([0-9]+)', s)]
    return sorted(items, key=key)
