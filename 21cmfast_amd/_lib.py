"""Loader for ``lib21cmfast_hip.so`` -- the C-ABI library holding the HIP kernels.

There is NO fallback: if the shared object is missing or fails to load, importing the
compute entry points raises ``ImportError`` (build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C 21cmfast_amd/csrc``).

PyTorch is imported first on purpose.  torch ships its own ``libamdhip64.so`` /
``librocfft.so`` with the same SONAMEs our library links against; loading torch first
makes the dynamic loader resolve our NEEDED entries to those already-loaded objects, so
the process has a single HIP runtime and torch device pointers are valid in our kernels.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from . import structs as S

_HERE = Path(__file__).resolve().parent
# C21CM_LIB: another build of the same library (kernel A/B experiments, tools/build_variant.sh);
# the default is the in-tree build next to this file.
LIB_PATH = Path(os.environ["C21CM_LIB"]) if os.environ.get("C21CM_LIB") else _HERE / "lib21cmfast_hip.so"

_lib = None


def _declare(lib):
    vp, i32, f64, f32 = C.c_void_p, C.c_int, C.c_double, C.c_float
    P = C.POINTER
    lib.c21cm_version.restype = C.c_char_p
    lib.c21cm_last_error.restype = C.c_char_p
    lib.c21cm_device_synchronize.restype = i32
    lib.c21cm_release_device_cache.restype = None

    lib.Broadcast_struct_global_all.restype = None
    lib.Broadcast_struct_global_all.argtypes = [
        P(S.SimulationOptions), P(S.MatterOptions), P(S.CosmoParams), P(S.AstroParams),
        P(S.AstroOptions), P(S.CosmoTables),
    ]
    lib.test_filter.restype = i32
    lib.test_filter.argtypes = [vp, f64, f64, f64, i32, vp]

    lib.c21cm_fft_r2c.restype = i32
    lib.c21cm_fft_r2c.argtypes = [vp, i32, i32, i32, vp]
    lib.c21cm_fft_c2r.restype = i32
    lib.c21cm_fft_c2r.argtypes = [vp, i32, i32, i32, vp]
    lib.c21cm_filter_grid.restype = i32
    lib.c21cm_filter_grid.argtypes = [vp, vp, i32, i32, i32, f64, f64, i32, f64, f64, vp]

    lib.c21cm_ionize_grids.restype = i32
    lib.c21cm_ionize_grids.argtypes = [
        P(S.IonizeSpec), P(S.PerturbedFieldStruct), P(S.IonizedBoxStruct), P(S.TsBoxStruct),
        P(S.HaloBoxStruct), P(S.IonizedBoxStruct), P(S.IonizeReport), vp,
    ]
    lib.c21cm_mturn_grids.restype = i32
    lib.c21cm_mturn_grids.argtypes = [P(S.MturnSpec), vp, vp, vp, vp, vp, vp, P(f64), P(f64), vp]
    lib.c21cm_ionize_shard_radii.restype = i32
    lib.c21cm_ionize_shard_radii.argtypes = [
        P(S.IonizeSpec), i32, i32, P(S.PerturbedFieldStruct), P(S.IonizedBoxStruct),
        P(S.TsBoxStruct), P(S.HaloBoxStruct), vp, P(S.IonizeReport), vp,
    ]
    lib.c21cm_ionize_shard_finish.restype = i32
    lib.c21cm_ionize_shard_finish.argtypes = [
        P(S.IonizeSpec), vp, P(S.PerturbedFieldStruct), P(S.IonizedBoxStruct), P(S.TsBoxStruct),
        P(S.HaloBoxStruct), P(S.IonizedBoxStruct), P(S.IonizeReport), vp,
    ]
    for name, argt in (
        ("ComputeInitialConditions", [C.c_ulonglong, P(S.InitialConditionsStruct)]),
        ("ComputePerturbedField", [f32, P(S.InitialConditionsStruct), P(S.PerturbedFieldStruct)]),
        ("ComputeIonizedBox", [
            f32, f32, P(S.PerturbedFieldStruct), P(S.PerturbedFieldStruct), P(S.IonizedBoxStruct),
            P(S.TsBoxStruct), P(S.HaloBoxStruct), P(S.InitialConditionsStruct),
            P(S.IonizedBoxStruct)]),
        ("c21cm_perturb_grids", [P(S.PerturbSpec), P(S.InitialConditionsStruct),
                                 P(S.PerturbedFieldStruct), vp]),
        ("c21cm_ics_grids", [P(S.IcsSpec), P(S.InitialConditionsStruct), vp]),
        ("UpdateXraySourceBox", [P(S.HaloBoxStruct), f64, f64, i32, f64, P(S.XraySourceBoxStruct)]),
        ("c21cm_fill_Rbox_grids", [P(S.RboxSpec), vp, vp, vp, vp, vp, vp]),
        ("c21cm_annular_filter_grids", [P(S.AnnularSpec), vp, vp, vp, vp, vp]),
        ("c21cm_ts_grids", [P(S.TsSpec), vp, P(S.TsBoxStruct), P(S.XraySourceBoxStruct), vp,
                            P(S.TsBoxStruct), P(S.TsReport), vp]),
        ("c21cm_ts_first_grids", [P(S.TsFirstSpec), vp, P(S.TsBoxStruct), vp]),
    ):
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = i32
            fn.argtypes = argt
    for name, argt in (("compute_mu_for_multiple_scattering", [f64]),
                       ("compute_eta_for_multiple_scattering", [f64]),
                       ("hyper_2F3", [f64, f64, f64])):
        if hasattr(lib, name):
            getattr(lib, name).restype = f64
            getattr(lib, name).argtypes = argt
    if hasattr(lib, "init_ps"):
        lib.init_ps.restype = None  # void init_ps(void), include/c21cm_abi.h
        lib.init_ps.argtypes = []
        lib.free_ps.restype = None
        lib.free_ps.argtypes = []
    return lib


def load(require_gpu: bool = False):
    """Return the ctypes handle of lib21cmfast_hip.so (loaded once per process)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `make -C 21cmfast_amd/csrc`). There is no CPU fallback."
            )
        try:
            import torch  # noqa: F401  (load torch's HIP runtime first, see module docstring)
        except Exception:  # pragma: no cover - torch is optional for C-only hosts
            pass
        try:
            _lib = _declare(C.CDLL(str(LIB_PATH), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else C.DEFAULT_MODE))
        except OSError as exc:
            raise ImportError(f"could not load {LIB_PATH}: {exc}") from exc
    if require_gpu:
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError(
                "21cmfast_amd needs an MI355X (no HIP device is visible); there is no CPU path"
            )
    return _lib


def last_error() -> str:
    return load().c21cm_last_error().decode()


class BackendError(RuntimeError):
    """A Compute*/c21cm_* call returned a non-zero status."""

    # reference: src/py21cmfast/wrapper/exceptions.py:84-117 maps the same codes
    NAMES = {
        1: "IOError", 2: "GSLError", 3: "ValueError", 4: "PhotonConsError",
        5: "TableGenerationError", 6: "TableEvaluationError", 7: "InfinityorNaNError",
        8: "MassDepZetaError", 9: "MemoryAllocError",
    }

    def __init__(self, code: int, where: str):
        self.code = code
        super().__init__(f"{where} failed with status {code} ({self.NAMES.get(code, '?')}): "
                         f"{last_error()}")


def check(code: int, where: str):
    if code != 0:
        raise BackendError(code, where)
