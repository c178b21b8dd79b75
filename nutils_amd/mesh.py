'''Mesh generators producing hot-path inputs (cf. /root/reference/src/nutils/mesh.py).'''

import numpy

from . import function, topology


def rectilinear(richshape, periodic=()):
    '''mesh.rectilinear (mesh.py:34-60): each entry of `richshape` is an int (n unit
    elements: geom = index + xi) or a uniformly spaced vertex array.  Returns
    (domain, geom).'''
    if periodic:
        raise NotImplementedError('periodic rectilinear meshes are outside the accelerated path')
    shape, offset, scale = [], [], []
    for v in richshape:
        if numpy.ndim(v) == 0:
            shape.append(int(v)), offset.append(0.), scale.append(1.)
        else:
            v = numpy.asarray(v, dtype=float)
            h = numpy.diff(v)
            if len(v) < 2 or not numpy.allclose(h, h[0], rtol=1e-14, atol=0):
                raise NotImplementedError('non-uniform vertex spacing: use an isoparametric geometry')
            shape.append(len(v) - 1), offset.append(v[0]), scale.append((v[-1] - v[0]) / (len(v) - 1))
    domain = topology.StructuredTopology(shape)
    return domain, function.RectilinearGeometry(domain, offset, scale)


def unitsquare(nelems, etype='square'):
    '''mesh.unitsquare (mesh.py:686-) for etype square.'''
    if etype != 'square':
        raise NotImplementedError('only square elements are on the accelerated path')
    return rectilinear([numpy.linspace(0, 1, nelems + 1)] * 2)
