"""MI355X-native backend for the MultiGridBarrier.jl inner Newton hot path.

Host-side mirror of the reference interface for this path: `fem1d/fem2d/fem3d/
fem2d_P1/fem2d_P2/spectral1d/spectral2d`, `subdivide`, `amg`, `assemble`, `mgb_solve` with a
`device=HIPDevice` keyword (reference: src/MultiGridBarrier.jl exports, src/device.jl).
Setup (meshes, hierarchies, grids) is NumPy on the CPU exactly as in the reference;
everything from `mgb_solve` down runs in the HIP library `libmgbhip.so` through its
C ABI (`include/mgbhip.h`).  There is no CPU solve path in this package: a missing
library raises.
"""
from .blockmatrices import BlockDiag, BlockColumn
from .multigrid import Geometry, MultiGrid, AMG, prepare_amg, amg_helper
from .fem2d_p1 import fem2d_P1, FEM2D_P1
from .fem2d_p2 import fem2d_P2, FEM2D_P2
from .tensorfem import fem1d, fem2d, fem3d, TensorFEM, tensor_dofmap
from .spectral import spectral1d, spectral2d, SPECTRAL1D, SPECTRAL2D
from .amg_prolongators import amg_ruge_stuben, amg_smoothed_aggregation
from .convex import Convex, Piece, convex_Euclidian_power, convex_linear, convex_piecewise, intersect
from .parabolic import parabolic_solve, ParabolicSOL
from .problem import (MGBProblem, assemble, amg, subdivide, geometric_mg, find_boundary, default_f, default_g,
                      default_D, default_idx)

try:  # the device layer needs the built shared library; importing the setup layer does not
    from .device import (Device, HIPDevice, CPUDevice, native_to_device, device_to_native,
                         default_device, default_device_set, mgb_cleanup, library_path)
    from .solve import mgb_solve, MGBSOL, MGBConvergenceFailure
except ImportError as _e:  # pragma: no cover - only while the device layer is being built
    _device_import_error = _e
