'''nutils_amd -- MI355X-native element-integration and sparse-assembly backend
behind the Nutils interfaces of that path (Sample.integrate / Sample.eval,
function.as_csr / eval, matrix.assemble_csr).  Hot path = hand-written HIP
(libnutils_hip.so, C ABI in include/nutils_hip.h); host layer = Python, like
the reference.  See DESIGN.md.'''

__version__ = '0.1'
