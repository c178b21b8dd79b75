'''see nutils_ext/units/__init__.py: the dimension types of the reference's nutils.SI under the name examples/cahnhilliard.py:10 imports them from'''
from nutils.SI import *  # noqa: F401,F403
from nutils.SI import Length, Time, Density, Tension, Energy, Pressure, Velocity  # noqa: F401
