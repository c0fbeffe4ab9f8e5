"""Parity tolerances live in the package (dex_tts_amd/tolerances.py) so that __graft_entry__.smoke() does not depend on the
tests package being importable; the tests import them from here."""
from dex_tts_amd.tolerances import *          # noqa: F401,F403
from dex_tts_amd.tolerances import LOWP, LOWP_AT, lowp_bounds  # noqa: F401
