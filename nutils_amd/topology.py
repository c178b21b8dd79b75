'''Topologies as producers of hot-path inputs.  Only what the assembly path needs
from /root/reference/src/nutils/topology.py: ``basis`` / ``field`` / ``sample`` /
``integral`` / ``integrate`` on structured (rectilinear) meshes
(topology.py:365-440, 1566-1575, 2209-2364) and on explicit element lists with
ragged bases (hierarchical refinements, topology.py:2926-3074, enter through
:class:`ElementList`).  Mesh generation, refinement logic, boundaries, trimming
stay with the reference (out of scope).'''

import numpy

from . import basis as _basis, function, points as _points, sample as _sample


class Topology:
    ndims: int
    nelems: int

    def __len__(self):
        return self.nelems

    def sample(self, ischeme, degree):
        '''topology.py:1566-1575: same points on every element.'''
        key = ischeme, degree
        cache = self.__dict__.setdefault('_samples', {})
        if key not in cache:
            if ischeme == 'gauss':
                pts = _points.gauss(degree, self.ndims)
            elif ischeme == 'bezier':
                pts = _points.bezier(degree, self.ndims)
            else:
                raise NotImplementedError(f'point scheme {ischeme!r}')
            cache[key] = _sample.Sample(self, pts)
        return cache[key]

    def integral(self, func, degree, ischeme='gauss'):
        '''topology.py:433-440.'''
        return self.sample(ischeme, degree).integral(func)

    def integrate(self, funcs, degree, arguments=None, ischeme='gauss', **kwargs):
        return self.sample(ischeme, degree).integrate(funcs, arguments, **kwargs)

    def field(self, name, *, btype='std', degree=1, shape=()):
        '''topology.py ``field`` -> function.field (function.py:2598-2627).'''
        return function.field(name, self.basis(btype, degree=degree), shape)


class StructuredTopology(Topology):
    '''mesh.rectilinear topology: prod(shape) elements, element index with the last
    axis fastest (transformseq.py:526-620).'''

    def __init__(self, shape):
        self.shape = tuple(int(n) for n in shape)
        self.ndims = len(self.shape)
        self.nelems = int(numpy.prod(self.shape))
        self._bases = {}

    def basis(self, btype, degree=1):
        key = btype, int(degree)
        if key not in self._bases:
            self._bases[key] = _basis.StructuredBasis(self.shape, btype, degree)
        return self._bases[key]


class ElementList(Topology):
    '''Unstructured list of axis-aligned box elements with externally supplied
    (possibly ragged) bases -- the form in which hierarchical / imported bases reach
    the assembly path.'''

    def __init__(self, origin, size):
        self.geom = function.BoxGeometry(origin, size)
        self.ndims = self.geom.ndims
        self.nelems = len(self.geom.origin)

    def plain_basis(self, coefficients, dofs, ndofs):
        return _basis.PlainBasis(coefficients, dofs, ndofs, self.ndims)
