"""`network` package of the drop-in (see nu_nerf_amd/compat/__init__.py): this directory provides renderer_zerothick and renderer;
every other `network.*` module is looked up in the `network/` directories further down sys.path -- the user's NU-NeRF checkout."""
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
for _p in list(sys.path):
    _d = os.path.abspath(os.path.join(_p or os.getcwd(), 'network'))
    if os.path.isdir(_d) and _d != _here and _d not in __path__:
        __path__.append(_d)
