"""bounded_lsq on MI355X — the trust-region step path of nmayorov/bounded-lsq
on hand-written gfx950 HIP kernels, behind the reference's own API names.

Importing the package does not touch the GPU; the first solver/plan does, and
raises if libblsq_hip.so is not built or no device is visible (no CPU path).
"""
from ._hip_step import (TrfStepSolver, DogboxStepSolver, SCALE_GIVEN,  # noqa: F401
                        SCALE_JAC_INIT, SCALE_JAC_UPDATE)

__all__ = ["TrfStepSolver", "DogboxStepSolver"]
