"""MI355X-native batched rigid-body simulation step: drop-in for jaxsim's ``js.model.step()``.

Host side is plain Python + NumPy + ctypes over a C-ABI library of hand-written HIP
kernels (``jaxsim_amd/csrc``, ``include/jaxsim_amd.h``).  No PyTorch, no JAX.
"""

from .model import (  # noqa: F401
    ActuationParams,
    FlatTerrain,
    HeightFieldTerrain,
    IntegratorType,
    JaxSimModel,
    PlaneTerrain,
    RelaxedRigidContacts,
    RelaxedRigidContactsParams,
    RigidContacts,
    RigidContactsParams,
    SoftContacts,
    SoftContactsParams,
    VelRepr,
)
from . import robots  # noqa: F401
