'''Quadrature tables (inputs of the hot path): Gauss-Legendre points on the unit
interval / tensor products thereof, in the reference's enumeration
(/root/reference/src/nutils/points.py:343-355 ``gauss1``: degree//2+1 points per
axis; :144-163 ``TensorPoints``: first coordinate slowest).'''

import numpy


class Points:
    def __init__(self, coords, weights):
        self.coords = numpy.ascontiguousarray(coords, dtype=float)
        self.weights = numpy.ascontiguousarray(weights, dtype=float)
        self.coords.setflags(write=False)
        self.weights.setflags(write=False)
        self.npoints, self.ndims = self.coords.shape


def gauss1(degree):
    n = degree // 2 + 1
    x, w = numpy.polynomial.legendre.leggauss(n)  # on [-1, 1], ascending
    return (x + 1.) * .5, w * .5


def gauss(degree, ndims):
    x, w = gauss1(degree)
    grids = numpy.meshgrid(*[x] * ndims, indexing='ij')
    coords = numpy.stack(grids, axis=-1).reshape(-1, ndims)
    weights = w
    for _ in range(ndims - 1):
        weights = numpy.multiply.outer(weights, w)
    return Points(coords, numpy.ravel(weights))


def bezier(n, ndims):
    '''Equidistant points including the element boundary (points.py ``bezier``), used by
    Sample.eval for post-processing; all weights are absent (None) in the reference, here
    a uniform dummy so the sample can still be passed to the eval kernel.'''
    x = numpy.linspace(0., 1., n)
    grids = numpy.meshgrid(*[x] * ndims, indexing='ij')
    coords = numpy.stack(grids, axis=-1).reshape(-1, ndims)
    return Points(coords, numpy.full(len(coords), numpy.nan))
