"""``clock()`` -- the wall timer the reference exports at top level (utils/timer.py:4-6, ``__init__.py:26``) and its tests
print timings with (tests/test_forward_equation.py:28-33)."""
from timeit import default_timer


def clock():
    """Seconds on the highest-resolution monotonic wall clock (the reference's docstring says UTC; it returns this)."""
    return default_timer()
