"""Host logic of bench.py that needs no GPU: the settle phase (an untimed run-in of a leg before its warmups) and the byte counts
the roofline figures are built from."""
import time
import types

import bench


class _FakeCuda:
    def synchronize(self):
        pass


class _FakeTensor:
    def __init__(self, v):
        self.v = v

    def item(self):
        return self.v


def _fake_torch():
    t = types.SimpleNamespace()
    t.cuda = _FakeCuda()
    t.int64 = "int64"
    t.tensor = lambda v, dtype=None, device=None: _FakeTensor(v[0])
    return t


def test_settle_runs_the_leg_for_about_the_requested_time_and_not_at_all_when_off():
    calls = []

    def leg():
        calls.append(time.perf_counter())
        time.sleep(0.002)
    assert bench.settle(leg, _fake_torch(), None, 1, 0.0) == 0 and not calls        # off: the leg is not touched
    n = bench.settle(leg, _fake_torch(), None, 1, 40.0)                               # ~40 ms of a ~2 ms leg
    assert 5 <= n <= 40 and len(calls) == n + 2                                       # (+ the two launches that size the loop)
    assert n == int(min(20000, max(1, n)))                                            # capped and at least one


def test_settle_agrees_on_the_launch_count_across_ranks():
    seen = {}

    class Dist:
        class ReduceOp:
            MAX = "max"

        @staticmethod
        def all_reduce(t, op=None):
            seen["n"] = t.v
            t.v = 7                                     # another rank measured a slower leg: everybody runs ITS count

    calls = []
    n = bench.settle(lambda: calls.append(1), _fake_torch(), Dist, 2, 1.0)
    assert n == 7 and len(calls) == 7 + 2 and seen["n"] >= 1


def test_algorithmic_bytes():
    # Ns samples x 8 corners x the element size + origin, direction and the result
    assert bench.algorithmic_bytes_per_ray(257, 8) == 257 * 8 * 8 + 56
    assert bench.fermat_bytes_per_ray(257, 2, "cubic") == 256 * 2 * 4 * 8 * 64 + 257 * 8 * 8 + 56
    assert bench.fermat_bytes_per_ray(129, 4, "linear") == 128 * 4 * 4 * 8 * 8 + 129 * 8 * 8 + 56
