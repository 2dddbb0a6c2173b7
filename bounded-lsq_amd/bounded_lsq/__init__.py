"""bounded_lsq on MI355X — the trust-region step path of nmayorov/bounded-lsq
on hand-written gfx950 HIP kernels, behind the reference's own API names
(``least_squares``, ``trf``, ``dogbox`` and the bound helpers the reference
package exports, bounded_lsq/__init__.py:3-13).

Importing the package does not touch the GPU; the first solver/plan does, and
raises if libblsq_hip.so is not built or no device is visible (no CPU path).
"""
from ._hip_step import (TrfStepSolver, DogboxStepSolver, SCALE_GIVEN,  # noqa: F401
                        SCALE_JAC_INIT, SCALE_JAC_UPDATE)
from ._drivers import trf, dogbox  # noqa: F401
from ._frontend import least_squares  # noqa: F401
from ._batch import least_squares_batch  # noqa: F401
from ._outer import OuterDriver  # noqa: F401
from ._hostmath import (active_mask as find_active_constraints,  # noqa: F401
                        prepare_bounds, cl_optimality as CL_optimality,
                        shift_into_interior as make_strictly_feasible)

__all__ = ['dogbox', 'trf', 'find_active_constraints', 'CL_optimality', 'prepare_bounds',
           'make_strictly_feasible', 'least_squares', 'least_squares_batch', 'TrfStepSolver', 'DogboxStepSolver',
           'OuterDriver']
