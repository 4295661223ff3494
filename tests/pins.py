"""Test helper: pin the GEMM schedule of the launches a test makes.

librf_flux.so has no process-global kernel switch (round 3): a launch's schedule travels in its descriptor
(`rf_gemm_desc.schedule`, `rf_attn_desc.kernel`).  `ops.gemm_schedule()` / `ops.attn_kernel()` set the default that the
Python wrappers put into the descriptors they build; `GemmPins` keeps the (tile, stream-K mode) vocabulary the tests were
written in and maps it onto `rf_gemm_schedule`:
    tile 128 -> TILE128, 257 -> PLAIN256 (plain-loop twin), 256 / 0 with sk 1 -> STREAMK, sk 2 -> PERSISTENT,
    256 (or 0 with sk 0) -> TILE256 (one tile per workgroup, never stream-K), 0 with sk -1 -> AUTO.
"""
from reflectionflow_amd import _lib as L
from reflectionflow_amd import ops


class GemmPins:
    def __init__(self):
        self._tile, self._sk = 0, -1
        self._tok = None

    def _apply(self):
        t, sk = self._tile, self._sk
        if t == 128:
            s = L.RF_SCHED_TILE128
        elif t == 257:
            s = L.RF_SCHED_PLAIN256
        elif sk == 1:
            s = L.RF_SCHED_STREAMK
        elif sk == 2:
            s = L.RF_SCHED_PERSISTENT
        elif t == 256 or sk == 0:
            s = L.RF_SCHED_TILE256
        else:
            s = L.RF_SCHED_AUTO
        ops._SCHED.set(s)

    def tile(self, t):
        assert t in (0, 128, 256, 257), t
        self._tile = t
        self._apply()

    def sk(self, mode):
        assert mode in (-1, 0, 1, 2), mode
        self._sk = mode
        self._apply()

    def reset(self):
        self._tile, self._sk = 0, -1
        ops._SCHED.set(L.RF_SCHED_AUTO)

    @staticmethod
    def last_path():
        return L.load().rf_debug_last_gemm_path()


pins = GemmPins()
