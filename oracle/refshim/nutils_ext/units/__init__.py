'''TEST INFRASTRUCTURE (build container only): stand-in for the third-party package `nutils-units` (pinned ">=0.2" by examples/cahnhilliard.py:3, absent from
/root/reference), which examples/cahnhilliard.py:10 imports as `nutils.units.typing`.  nutils-units is the spin-off of the reference's own `nutils.SI`
module (src/nutils/SI.py: dimension metaclass, Quantity, unit parser, nutils_dispatch integration); this stand-in re-exports that module.  The directory
above is appended to `nutils.__path__` by the test runner (tests/seam_hook_run.py), nothing is written into the reference tree.'''
from nutils.SI import *  # noqa: F401,F403
