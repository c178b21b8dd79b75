'''Mesh generators producing hot-path inputs (cf. /root/reference/src/nutils/mesh.py).'''

import numpy

from . import function, topology


def rectilinear(richshape, periodic=()):
    '''mesh.rectilinear (mesh.py:34-60): each entry of `richshape` is an int (n unit
    elements: geom = index + xi) or a vertex array.  Returns (domain, geom).  Uniformly
    spaced vertices give the rectilinear geometry the structured fast paths recognise,
    anything else one axis-aligned box per element.  periodic: axes along which the bases wrap around.'''
    axes = [numpy.arange(int(v) + 1, dtype=float) if numpy.ndim(v) == 0 else numpy.asarray(v, dtype=float) for v in richshape]
    if any(v.ndim != 1 or len(v) < 2 for v in axes):
        raise ValueError('every axis needs at least two vertices')
    shape = [len(v) - 1 for v in axes]
    domain = topology.StructuredTopology(shape, periodic)
    steps = [numpy.diff(v) for v in axes]
    if all(numpy.allclose(h, h[0], rtol=1e-14, atol=0) for h in steps):
        scale = [1. if numpy.ndim(r) == 0 else (v[-1] - v[0]) / (len(v) - 1) for r, v in zip(richshape, axes)]
        return domain, function.RectilinearGeometry(domain, [v[0] for v in axes], scale)
    return domain, function.GradedGeometry(domain, axes)


def unitsquare(nelems, etype='square'):
    '''mesh.unitsquare (mesh.py:686-) for etype square.'''
    if etype != 'square':
        raise NotImplementedError('only square elements are on the accelerated path')
    return rectilinear([numpy.linspace(0, 1, nelems + 1)] * 2)
