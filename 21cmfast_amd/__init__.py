"""21cmfast_amd -- MI355X-native backend for 21cmFAST's 3-D box hot path.

InitialConditions -> PerturbedField -> IonizedBox as hand-written CDNA4 (gfx950) HIP
kernels behind the reference's own C ABI (``include/c21cm_abi.h``).  This Python
package is the thin host-side mirror of py21cmfast's wrapper layer for that path:
struct definitions and defaults (``structs``), numpy / torch front-ends of the explicit-scalar
entry points (``grid_api``), the sharded R loop (``distributed``), synthetic benchmark inputs
(``workloads``) and the library loader (``_lib``).

The directory name starts with a digit, so import it with
``importlib.import_module("21cmfast_amd")``.
"""

from . import structs  # noqa: F401
from ._lib import BackendError, LIB_PATH, check, last_error, load  # noqa: F401

__version__ = "0.1.0"
