"""onephase.jl_amd -- MI355X-native KKT linear-system path of the one-phase IPM.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/okkt.h),
the ctypes binding, and the host-side mirror of the reference's linear-solver / KKT-solver
plug-in interface.  The directory name contains a dot, so import it through the root-level
shim:  `import onephase_jl_amd`.
"""
from . import _lib  # noqa: F401
from .linear_system_solvers import (  # noqa: F401
    OkktError,
    abstract_linear_system_solver,
    finalize_b,
    inertia_status,
    initialize_b,
    linear_solver_HIP,
)
