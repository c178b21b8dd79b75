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

    def __init__(self, shape, periodic=()):
        self.shape = tuple(int(n) for n in shape)
        self.ndims = len(self.shape)
        self.nelems = int(numpy.prod(self.shape))
        self.periodic = tuple(sorted(int(i) for i in periodic))  # axes whose bases wrap around (mesh.rectilinear(periodic=...), mesh.py:34-60)
        if any(not 0 <= i < self.ndims for i in self.periodic):
            raise ValueError('periodic axis out of range')
        self._bases = {}

    def basis(self, btype, degree=1):
        key = btype, int(degree)
        if key not in self._bases:
            self._bases[key] = _basis.StructuredBasis(self.shape, btype, degree, self.periodic)
        return self._bases[key]

    @property
    def boundary(self):
        if not hasattr(self, '_boundary'):
            self._boundary = _Boundary(self)
        return self._boundary


class BoundaryTopology(Topology):
    '''One side of a structured topology (``domain.boundary['left']`` etc.; names as in the
    reference: left/right = first axis, bottom/top = second, front/back = third).  Bases and
    fields are those of the parent (their traces); samples are Gauss points on the face,
    embedded in the parent element's reference coordinates.'''

    def __init__(self, parent, axis, side):
        self.parent, self.axis, self.side = parent, axis, side
        self.ndims = parent.ndims
        idx = numpy.arange(parent.nelems).reshape(parent.shape)
        self.elements = numpy.ascontiguousarray(numpy.take(idx, 0 if side == 0 else parent.shape[axis] - 1, axis=axis).ravel(), dtype=numpy.int32)
        self.nelems = len(self.elements)

    def basis(self, btype, degree=1):
        return self.parent.basis(btype, degree)

    def sample(self, ischeme, degree):
        key = ischeme, degree
        cache = self.__dict__.setdefault('_samples', {})
        if key not in cache:
            if ischeme != 'gauss':
                raise NotImplementedError(f'point scheme {ischeme!r} on a boundary')
            nd = self.ndims
            face = _points.gauss(degree, nd - 1) if nd > 1 else _points.Points(numpy.zeros((1, 0)), numpy.ones(1))
            coords = numpy.insert(face.coords, self.axis, float(self.side), axis=1)
            cache[key] = _sample.Sample(self.parent, _points.Points(coords, face.weights), elist=self.elements, bnd_axis=self.axis)
        return cache[key]


_BNAMES = ('left', 'right'), ('bottom', 'top'), ('front', 'back')


class _Boundary:
    def __init__(self, parent):
        self.parent = parent
        self._cache = {}

    def sides(self):
        '''(a periodic axis has no boundary: topology.py StructuredTopology.boundary skips it)'''
        return [self[n] for axis, names in enumerate(_BNAMES[:self.parent.ndims]) if axis not in self.parent.periodic for n in names]

    def integral(self, func, degree):
        '''Integral over the whole boundary (domain.boundary.integral): sum over the sides.'''
        total = None
        for side in self.sides():
            term = side.integral(func, degree)
            total = term if total is None else total + term
        return total

    def __getitem__(self, name):
        if name not in self._cache:
            for axis, names in enumerate(_BNAMES[:self.parent.ndims]):
                if name in names and axis not in self.parent.periodic:
                    self._cache[name] = BoundaryTopology(self.parent, axis, names.index(name))
                    break
            else:
                raise KeyError(name)
        return self._cache[name]


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
