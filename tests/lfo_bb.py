"""BabyBearRingNTT binding of the CPU oracle (oracle/liblfo_bb.so): the source of tests/lfo.py executed with
_RING = "babybear".  TEST INFRASTRUCTURE ONLY (same rule as lfo.py)."""
import os as _os
import sys as _sys

_RING = "babybear"
_src = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lfo.py")
exec(compile(open(_src).read(), _src, "exec"), globals())
