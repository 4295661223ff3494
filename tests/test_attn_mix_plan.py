"""CPU test of the mixed-size attention launch's planner (host arithmetic in librf_flux, reached through the read-only
rf_debug_attn_mix_plan -- no GPU): per head S / 16 q-tiles = 16 a + 12 b with b % 4 == 0 (256- and 192-query workgroups), the plan
never loses to the plain grid in its own cost model, and the BASELINE shapes get the plans the measurements
in profiles/r03_kb_attn_mixsize.log found best."""
import ctypes as C

import pytest

from reflectionflow_amd import _lib as L


def plan(S, heads, cus=256):
    out = (C.c_int32 * 4)()
    assert L.load().rf_debug_attn_mix_plan(S, heads, cus, out) == 0
    return tuple(out)


@pytest.mark.parametrize("S", [256, 1024, 2048, 4608, 5632, 8192, 17920])
@pytest.mark.parametrize("heads", [3, 8, 24, 40])
def test_plan_invariants(S, heads):
    a, b, span_milli, rounds = plan(S, heads)
    assert a >= 0 and b >= 0 and b % 4 == 0 and 16 * a + 12 * b == S // 16            # every query row exactly once
    assert span_milli <= 1000 * rounds + 1                                             # never worse than the plain grid in the model
    if b == 0:
        assert a == S // 256                                                           # the plain grid
    if heads * (S // 256) <= 256 and b > 0:
        assert heads * (a + b) <= 256 or span_milli < 1000                             # one round either way: smaller workgroups on more CUs


def test_baseline_shapes():
    assert plan(4608, 24)[:2] == (9, 12)          # cfg2: 216 + 288 workgroups, 1.9 rounds instead of 2
    assert plan(5632, 24)[:2] == (4, 24)          # cfg4: 2.7 instead of 3
    a, b, span, rounds = plan(17920, 24)
    assert span > 0.96 * 1000 * rounds            # cfg5: < 4 % predicted -> AUTO keeps the plain grid (measured 1-4 % slower)


def test_bad_arguments_fail_loudly():
    out = (C.c_int32 * 4)()
    assert L.load().rf_debug_attn_mix_plan(300, 24, 256, out) < 0          # S % 256 != 0
    assert L.load().rf_debug_attn_mix_plan(4608, 0, 256, out) < 0
