"""Python front-end of the explicit-scalar entry points (``include/c21cm_grid.h``).

Inputs may be numpy arrays (host memory -- the library stages them, which is the
py21cmfast/CFFI situation) or torch CUDA tensors (HBM -- used in place).  Outputs are
created in the same kind of memory as ``density``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import structs as S
from ._lib import check, load


def _is_torch(a) -> bool:
    return a is not None and type(a).__module__.startswith("torch")


def _fptr(a):
    if a is None:
        return None
    if _is_torch(a):
        assert a.is_contiguous() and a.dtype.is_floating_point and a.element_size() == 4
        return C.cast(a.data_ptr(), S.c_float_p)
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(S.c_float_p)


def _vptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _new_like(ref, fill: float, dtype=None):
    if _is_torch(ref):
        import torch

        return torch.full(ref.shape, fill, dtype=dtype or torch.float32, device=ref.device)
    return np.full(ref.shape, fill, dtype or np.float32)


def _stack_like(ref, n: int):
    """zeros of shape (n, *ref.shape), float32, where ``ref`` lives"""
    if _is_torch(ref):
        import torch

        return torch.zeros((n,) + tuple(ref.shape), dtype=torch.float32, device=ref.device)
    return np.zeros((n,) + tuple(ref.shape), np.float32)


def _stream(stream):
    if stream is not None:
        return C.c_void_p(stream)
    try:
        import torch

        if torch.cuda.is_available():
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    except Exception:
        pass
    return C.c_void_p(0)


def fft_r2c(padded, nx, ny, nz, stream=None):
    check(load().c21cm_fft_r2c(_vptr(padded), nx, ny, nz, _stream(stream)), "c21cm_fft_r2c")


def fft_c2r(padded, nx, ny, nz, stream=None):
    check(load().c21cm_fft_c2r(_vptr(padded), nx, ny, nz, _stream(stream)), "c21cm_fft_c2r")


def filter_grid(box, box_len, filter_type, R, R_param=0.0, box_len_z=None, stream=None):
    """r2c -> /N -> W(kR) -> c2r of one box (reference: filtering.c:397-445)."""
    nx, ny, nz = box.shape
    out = _new_like(box, 0.0)
    check(
        load().c21cm_filter_grid(_vptr(box), _vptr(out), nx, ny, nz, box_len,
                                 box_len if box_len_z is None else box_len_z, filter_type, R,
                                 R_param, _stream(stream)),
        "c21cm_filter_grid",
    )
    return out


class IonizeBuffers:
    """Owns the output arrays of one IonizedBox (what ``IonizedBox.new`` allocates in
    py21cmfast, reference: src/py21cmfast/wrapper/outputs.py:1475-1545)."""

    def __init__(self, density, need_nion: bool = False, minimize_memory: bool = False,
                 recomb_model: int = 0, mini_radii: int = 0):
        self.neutral_fraction = _new_like(density, 1.0)  # initialised to ones (:1524-1527)
        self.z_reion = _new_like(density, 0.0)
        self.kinetic_temperature = None if minimize_memory else _new_like(density, 0.0)
        self.unnormalised_nion = _new_like(density, 0.0) if need_nion else None
        self.unnormalised_nion_mini = None
        if mini_radii:  # USE_MINI_HALOS: one f_coll grid per filter radius (:1538-1543)
            stack = _stack_like(density, mini_radii)
            self.unnormalised_nion = stack
            self.unnormalised_nion_mini = _stack_like(density, mini_radii)
        # recombination models (:1526-1537): Gamma_12 and the mean free path are always part of
        # the reference's IonizedBox; here they are only allocated when something writes them
        self.ionisation_rate_G12 = self.mean_free_path = self.cumulative_recombinations = None
        if recomb_model:
            self.ionisation_rate_G12 = _new_like(density, 0.0)
            if not minimize_memory:
                self.mean_free_path = _new_like(density, 0.0)
            if recomb_model == 2:
                self.cumulative_recombinations = _new_like(density, 0.0)
            else:  # homogeneous: one number, shape (1, 1, 1)
                self.cumulative_recombinations = _new_like(density[:1, :1, :1], 0.0)

    def reset(self):
        """Back to the state of a freshly allocated IonizedBox.  z_reion needs no reset: every
        ComputeIonizedBox path overwrites it (IonisationBox.c:1372-1378 fills it with -1)."""
        self.neutral_fraction[...] = 1.0
        if self.kinetic_temperature is not None:
            self.kinetic_temperature[...] = 0.0
        for a in (self.ionisation_rate_G12, self.mean_free_path, self.cumulative_recombinations):
            if a is not None:
                a[...] = 0.0

    def struct(self) -> S.IonizedBoxStruct:
        return S.IonizedBoxStruct(
            neutral_fraction=_fptr(self.neutral_fraction), z_reion=_fptr(self.z_reion),
            kinetic_temperature=_fptr(self.kinetic_temperature),
            unnormalised_nion=_fptr(self.unnormalised_nion),
            unnormalised_nion_mini=_fptr(self.unnormalised_nion_mini),
            ionisation_rate_G12=_fptr(self.ionisation_rate_G12),
            mean_free_path=_fptr(self.mean_free_path),
            cumulative_recombinations=_fptr(self.cumulative_recombinations),
        )


def _input_structs(density, n_ion, xe, Tneutral, prev_z_reion, prev_nrec=None, whalo_sfr=None):
    pf = S.PerturbedFieldStruct(density=_fptr(density))
    prev = S.IonizedBoxStruct(z_reion=_fptr(prev_z_reion),
                              cumulative_recombinations=_fptr(prev_nrec))
    ts = S.TsBoxStruct(xray_ionised_fraction=_fptr(xe), kinetic_temp_neutral=_fptr(Tneutral))
    hb = S.HaloBoxStruct(n_ion=_fptr(n_ion), whalo_sfr=_fptr(whalo_sfr))
    return pf, prev, ts, hb


def ionize_grids(spec: S.IonizeSpec, density, n_ion=None, xe=None, Tneutral=None,
                 prev_z_reion=None, buffers: IonizeBuffers | None = None, stream=None,
                 prev_nrec=None, whalo_sfr=None, mini=None):
    """One ComputeIonizedBox grid pass on the MI355X.  Returns (buffers, box_struct, report).
    ``prev_nrec`` (the previous box's cumulative_recombinations) and ``whalo_sfr`` (HaloBox) are
    the extra inputs of the recombination models.  ``mini`` (USE_MINI_HALOS): dict with
    prev_density, log10_mturn_acg, log10_mturn_mcg [N] and prev_nion, prev_nion_mini
    [n_radii, N] (the previous box's per-radius f_coll history), numpy or CUDA tensors."""
    if buffers is None:
        buffers = IonizeBuffers(density, need_nion=spec.fcoll_mode != 0,
                                minimize_memory=bool(spec.minimize_memory),
                                recomb_model=spec.recomb_model,
                                mini_radii=spec.n_radii if mini is not None else 0)
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion, prev_nrec,
                                      whalo_sfr)
    if mini is not None:
        spec.prev_density = _fptr(mini["prev_density"])
        spec.log10_mturn_acg = _fptr(mini["log10_mturn_acg"])
        spec.log10_mturn_mcg = _fptr(mini["log10_mturn_mcg"])
        prev.unnormalised_nion = _fptr(mini["prev_nion"])
        prev.unnormalised_nion_mini = _fptr(mini["prev_nion_mini"])
    box = buffers.struct()
    rep = S.IonizeReport()
    check(
        load().c21cm_ionize_grids(C.byref(spec), C.byref(pf), C.byref(prev), C.byref(ts),
                                  C.byref(hb), C.byref(box), C.byref(rep), _stream(stream)),
        "c21cm_ionize_grids",
    )
    return buffers, box, rep


def mturn_grids(spec: S.MturnSpec, prev_G12, prev_z_reion, J_21_LW, vcb=None, stream=None):
    """calculate_mcrit_boxes on the device: (log10 M_turn,a, log10 M_turn,m, <a>, <m>); the
    grids are allocated like ``J_21_LW`` (numpy array or CUDA tensor)."""
    a, m = _new_like(J_21_LW, 0.0), _new_like(J_21_LW, 0.0)
    ave_a, ave_m = C.c_double(), C.c_double()
    check(
        load().c21cm_mturn_grids(C.byref(spec), _vptr(prev_G12), _vptr(prev_z_reion),
                                 _vptr(J_21_LW), _vptr(vcb), _vptr(a), _vptr(m), C.byref(ave_a),
                                 C.byref(ave_m), _stream(stream)),
        "c21cm_mturn_grids",
    )
    return a, m, ave_a.value, ave_m.value


def ionize_shard_radii(spec, rank, world, first_cross, density, n_ion=None, xe=None,
                       Tneutral=None, prev_z_reion=None, stream=None, want_report=True):
    """Shard phase: this rank's radii -> ``first_cross`` (uint8 CUDA tensor).
    ``want_report=False`` skips the per-radius f_coll means and with them the host
    synchronisation at the end of the phase (the call only enqueues work)."""
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion)
    rep = S.IonizeReport() if want_report else None
    check(
        load().c21cm_ionize_shard_radii(C.byref(spec), rank, world, C.byref(pf), C.byref(prev),
                                        C.byref(ts), C.byref(hb),
                                        C.c_void_p(first_cross.data_ptr()),
                                        C.byref(rep) if want_report else None,
                                        _stream(stream)),
        "c21cm_ionize_shard_radii",
    )
    return rep


def ionize_shard_set_means(means):
    """Hand the rank-summed per-radius f_coll grid means to the next finish step."""
    m = np.ascontiguousarray(means, np.float64)
    lib = load()
    lib.c21cm_ionize_shard_set_means.restype = C.c_int
    lib.c21cm_ionize_shard_set_means.argtypes = [C.c_void_p, C.c_int]
    check(lib.c21cm_ionize_shard_set_means(m.ctypes.data_as(C.c_void_p), int(m.size)),
          "c21cm_ionize_shard_set_means")


def ionize_shard_finish(spec, first_cross, density, n_ion=None, xe=None, Tneutral=None,
                        prev_z_reion=None, buffers: IonizeBuffers | None = None, stream=None):
    """Finish phase on the owning rank: apply the reduced mask, radius 0, post-loop."""
    if buffers is None:
        buffers = IonizeBuffers(density, need_nion=spec.fcoll_mode != 0,
                                minimize_memory=bool(spec.minimize_memory))
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion)
    box = buffers.struct()
    rep = S.IonizeReport()
    check(
        load().c21cm_ionize_shard_finish(C.byref(spec), C.c_void_p(first_cross.data_ptr()),
                                         C.byref(pf), C.byref(prev), C.byref(ts), C.byref(hb),
                                         C.byref(box), C.byref(rep), _stream(stream)),
        "c21cm_ionize_shard_finish",
    )
    return buffers, box, rep


class ShardSlabState(C.Structure):
    """c21cm_shard_slab_state (include/c21cm_grid.h): what the exchange callback of the slab finish sees."""
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("n_chunks", C.c_int), ("chunk_begin", C.c_int),
                ("chunk_end", C.c_int), ("chunk_cells", C.c_size_t), ("cell_begin", C.c_size_t),
                ("cell_end", C.c_size_t), ("ntot", C.c_size_t), ("partials_stars", C.c_void_p),
                ("partials_xh", C.c_void_p), ("flag", C.c_void_p), ("out", C.c_void_p * 3)]


SLAB_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ShardSlabState), C.c_int, C.c_void_p)


def shard_slab_supported(spec) -> bool:
    """Does the finish phase of this model run by cell slabs (c21cm_ionize_shard_slab_supported)?"""
    lib = load()
    lib.c21cm_ionize_shard_slab_supported.restype = C.c_int
    return bool(lib.c21cm_ionize_shard_slab_supported(C.byref(spec)))


def shard_slab(spec, rank: int, world: int) -> dict:
    """Chunks and cells of `rank`'s slab of the final sweep (c21cm_ionize_shard_slab)."""
    lib = load()
    lib.c21cm_ionize_shard_slab.restype = C.c_int
    cb, ce, nch = C.c_int(), C.c_int(), C.c_int()
    c0, c1, cc = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(lib.c21cm_ionize_shard_slab(C.byref(spec), C.c_int(rank), C.c_int(world), C.byref(cb),
                                      C.byref(ce), C.byref(c0), C.byref(c1), C.byref(nch), C.byref(cc)),
          "c21cm_ionize_shard_slab")
    return {"chunk_begin": cb.value, "chunk_end": ce.value, "cell_begin": c0.value,
            "cell_end": c1.value, "n_chunks": nch.value, "chunk_cells": cc.value}


class _DevView:
    """A raw device address as a CUDA array (torch.as_tensor reads __cuda_array_interface__)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                         "version": 2}


def device_view(ptr: int, n: int, kind: str):
    """torch view (no copy) of n elements at a device address; kind: 'f8', 'f4', 'i4'."""
    import torch

    return torch.as_tensor(_DevView(ptr, n, "<" + kind), device="cuda")


def ionize_shard_finish_slab(spec, first_cross, rank, world, density, n_ion=None, xe=None,
                             Tneutral=None, prev_z_reion=None, buffers: IonizeBuffers | None = None,
                             exchange=None, outputs_gathered=False, stream=None):
    """Finish phase of `rank` by cell slabs (c21cm_ionize_shard_finish_slab): the final sweep over the
    rank's chunks from the combined first crossings of that slab, `exchange(state, local_status)`
    (a Python callable: all-gathers the chunk sums -- and the output slabs if the caller wants whole
    boxes -- and returns the agreed status), then the fixed-order reduce and the post-loop on every
    rank.  Returns (buffers, box_struct, report); the outputs are valid on the rank's slab
    (everywhere with ``outputs_gathered``)."""
    if buffers is None:
        buffers = IonizeBuffers(density, need_nion=spec.fcoll_mode != 0,
                                minimize_memory=bool(spec.minimize_memory))
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion)
    box = buffers.struct()
    rep = S.IonizeReport()
    lib = load()
    lib.c21cm_ionize_shard_finish_slab.restype = C.c_int
    err = []

    def _cb(user, state, local_status, strm):
        try:
            return int(exchange(state.contents, local_status) or 0)
        except Exception as e:  # an exception must not unwind through the C frames
            err.append(e)
            return 3

    cb = SLAB_EXCHANGE_FN(_cb) if exchange is not None else C.cast(None, SLAB_EXCHANGE_FN)
    st = lib.c21cm_ionize_shard_finish_slab(
        C.byref(spec), C.c_void_p(first_cross.data_ptr()), C.c_int(rank), C.c_int(world), C.byref(pf),
        C.byref(prev), C.byref(ts), C.byref(hb), C.byref(box), C.byref(rep), cb, None,
        C.c_int(1 if outputs_gathered else 0), _stream(stream))
    if err:
        raise err[0]
    check(st, "c21cm_ionize_shard_finish_slab")
    return buffers, box, rep


def shard_pack_mask_bits(first_cross, stream=None):
    """uint8 first crossings -> one bit per cell (uint32 words; c21cm_shard_pack_mask_bits)."""
    import torch

    n = first_cross.numel()
    bits = torch.empty((n + 31) // 32, dtype=torch.int32, device=first_cross.device)
    lib = load()
    lib.c21cm_shard_pack_mask_bits.restype = C.c_int
    check(lib.c21cm_shard_pack_mask_bits(C.c_void_p(first_cross.data_ptr()), C.c_void_p(bits.data_ptr()),
                                         C.c_size_t(n), _stream(stream)), "c21cm_shard_pack_mask_bits")
    return bits


def shard_or_unpack_mask_bits(pieces, first_cross_slab, stream=None):
    """OR of the packed pieces (2-D int32 [world][words]) into the uint8 cells of a slab
    (c21cm_shard_or_unpack_mask_bits)."""
    lib = load()
    lib.c21cm_shard_or_unpack_mask_bits.restype = C.c_int
    world, words = pieces.shape
    check(lib.c21cm_shard_or_unpack_mask_bits(C.c_void_p(pieces.data_ptr()), C.c_size_t(words),
                                              C.c_int(world), C.c_void_p(first_cross_slab.data_ptr()),
                                              C.c_size_t(first_cross_slab.numel()), _stream(stream)),
          "c21cm_shard_or_unpack_mask_bits")
    return first_cross_slab


def ionize_shard_radii_keys(spec, rank, world, cross_keys, density, n_ion=None, xe=None,
                            Tneutral=None, prev_z_reion=None, prev_nrec=None, whalo_sfr=None,
                            stream=None):
    """Shard phase of a recombination model: this rank's radii -> ``cross_keys`` (int64 / uint64
    CUDA tensor of N entries: bits(mean free path) << 32 | bits(Gamma_12), 0 = not crossed)."""
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion, prev_nrec,
                                      whalo_sfr)
    lib = load()
    lib.c21cm_ionize_shard_radii_keys.restype = C.c_int
    check(lib.c21cm_ionize_shard_radii_keys(C.byref(spec), C.c_int(rank), C.c_int(world),
                                            C.byref(pf), C.byref(prev), C.byref(ts), C.byref(hb),
                                            C.c_void_p(cross_keys.data_ptr()), None,
                                            _stream(stream)),
          "c21cm_ionize_shard_radii_keys")


def ionize_shard_finish_keys(spec, cross_keys, density, n_ion=None, xe=None, Tneutral=None,
                             prev_z_reion=None, prev_nrec=None, whalo_sfr=None,
                             buffers: IonizeBuffers | None = None, stream=None):
    """Finish phase of a recombination model on the owning rank (reduced keys in)."""
    if buffers is None:
        buffers = IonizeBuffers(density, need_nion=spec.fcoll_mode != 0,
                                minimize_memory=bool(spec.minimize_memory),
                                recomb_model=spec.recomb_model)
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion, prev_nrec,
                                      whalo_sfr)
    box = buffers.struct()
    rep = S.IonizeReport()
    lib = load()
    lib.c21cm_ionize_shard_finish_keys.restype = C.c_int
    check(lib.c21cm_ionize_shard_finish_keys(C.byref(spec), C.c_void_p(cross_keys.data_ptr()),
                                             C.byref(pf), C.byref(prev), C.byref(ts), C.byref(hb),
                                             C.byref(box), C.byref(rep), _stream(stream)),
          "c21cm_ionize_shard_finish_keys")
    return buffers, box, rep


def ionize_last_loop_flags() -> int:
    """Which R loop the last ionisation call set up (c21cm_ionize_last_loop_flags): 1 fused, 2 fused
    recombination loop, 4 third spectrum in the barrier kernel, 8 ... which is the filtered N_rec, 16 a
    fourth spectrum (x_e and filtered N_rec), 32 two radii per sweep."""
    lib = load()
    lib.c21cm_ionize_last_loop_flags.restype = C.c_int
    return int(lib.c21cm_ionize_last_loop_flags())


def shard_rc_supported(spec) -> bool:
    """Does a recombination spec shard through the fused loop (first-crossing index + Gamma_12,
    5 bytes per cell) rather than through the 64-bit keys?  c21cm_ionize_shard_rc_supported."""
    return bool(load().c21cm_ionize_shard_rc_supported(C.byref(spec)))


def ionize_shard_radii_rc(spec, rank, world, first_cross, cross_g12, density, n_ion=None,
                          prev_z_reion=None, prev_nrec=None, whalo_sfr=None, xe=None, Tneutral=None,
                          stream=None):
    """Shard phase of the fused recombination loop: this rank's radii -> ``first_cross`` (uint8
    CUDA tensor, radius index of the first crossing, 0 = none) and ``cross_g12`` (float32,
    Gamma_12 at that crossing).  ``xe`` / ``Tneutral``: the TsBox grids of a spin-temperature run."""
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion, prev_nrec, whalo_sfr)
    lib = load()
    lib.c21cm_ionize_shard_radii_rc.restype = C.c_int
    check(lib.c21cm_ionize_shard_radii_rc(C.byref(spec), C.c_int(rank), C.c_int(world), C.byref(pf),
                                          C.byref(prev), C.byref(ts), C.byref(hb),
                                          C.c_void_p(first_cross.data_ptr()),
                                          C.c_void_p(cross_g12.data_ptr()), None, _stream(stream)),
          "c21cm_ionize_shard_radii_rc")


def ionize_shard_finish_rc(spec, first_cross, cross_g12, density, n_ion=None, prev_z_reion=None,
                           prev_nrec=None, whalo_sfr=None, buffers: IonizeBuffers | None = None,
                           xe=None, Tneutral=None, stream=None):
    """Finish phase of the fused recombination loop on the owning rank (combined grids in)."""
    if buffers is None:
        buffers = IonizeBuffers(density, need_nion=spec.fcoll_mode != 0,
                                minimize_memory=bool(spec.minimize_memory),
                                recomb_model=spec.recomb_model)
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion, prev_nrec, whalo_sfr)
    box = buffers.struct()
    rep = S.IonizeReport()
    lib = load()
    lib.c21cm_ionize_shard_finish_rc.restype = C.c_int
    check(lib.c21cm_ionize_shard_finish_rc(C.byref(spec), C.c_void_p(first_cross.data_ptr()),
                                           C.c_void_p(cross_g12.data_ptr()), C.byref(pf),
                                           C.byref(prev), C.byref(ts), C.byref(hb), C.byref(box),
                                           C.byref(rep), _stream(stream)),
          "c21cm_ionize_shard_finish_rc")
    return buffers, box, rep


def combine_cross_g12(mask, g12, peer_mask, peer_g12, stream=None):
    """c21cm_shard_combine_cross_g12: own slab (uint8 / float32 CUDA tensors, in place) against
    the peers' slabs ``peer_mask`` [n_peers, stride] / ``peer_g12`` [n_peers, stride]."""
    lib = load()
    lib.c21cm_shard_combine_cross_g12.restype = C.c_int
    n_peers, stride = peer_mask.shape
    check(lib.c21cm_shard_combine_cross_g12(C.c_void_p(mask.data_ptr()), C.c_void_p(g12.data_ptr()),
                                            C.c_void_p(peer_mask.data_ptr()),
                                            C.c_void_p(peer_g12.data_ptr()), C.c_int(n_peers),
                                            C.c_size_t(stride), C.c_size_t(mask.numel()),
                                            _stream(stream)),
          "c21cm_shard_combine_cross_g12")


PLACEMENT_OUTCOMES = ("placed against the partner by timed launches", "off / not applicable",
                      "other tenants on the device", "another process was walking", "no faster candidate in the budget",
                      "an earlier walk for this size found nothing", "tenancy unknown and the device is busy",
                      "the walk's time budget ran out")


def placement_report():
    """What the last placement decision of a work spectrum was and what it cost (csrc/host/placement.c:
    c21cm_placement_report); None before any."""
    lib = load()
    out = (C.c_double * 8)()
    lib.c21cm_placement_report.restype = C.c_int
    if lib.c21cm_placement_report(out) != 0:
        return None
    return {"outcome": PLACEMENT_OUTCOMES[int(out[0])], "held_GB": round(out[1], 2), "probes": int(out[2]),
            "chosen_ms": round(out[3], 4), "first_candidate_ms": round(out[4], 4), "decision_wall_ms": round(out[5], 2),
            "tenants": int(out[6]), "walks": int(out[7])}


def placement_set(mode: int):
    """Opt into (1; 2 = also on a shared device) or out of (0) the placement walk of the work spectra; -1: the
    environment's C21CM_WS_PLACE (unset: off).  csrc/host/placement.c."""
    lib = load()
    lib.c21cm_placement_set.restype = C.c_int
    check(lib.c21cm_placement_set(C.c_int(mode)), "c21cm_placement_set")


def shard_init_from_torch(group=None):
    """Bootstrap the library's own RCCL communicator from an initialised ``torch.distributed``
    job: rank 0 draws the unique id in C (c21cm_shard_unique_id), torch broadcasts its 128 bytes,
    every rank calls c21cm_shard_init.  Afterwards the exchange of the sharded R loop happens
    inside the C library (c21cm_ionize_sharded, and ComputeIonizedBox itself)."""
    import torch
    import torch.distributed as dist

    lib = load(require_gpu=True)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        check(lib.c21cm_shard_unique_id(buf), "c21cm_shard_unique_id")
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor(list(buf), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0, group=group)
    raw = bytes(t.cpu().tolist())
    check(lib.c21cm_shard_init(C.c_int(rank), C.c_int(world), raw), "c21cm_shard_init")
    return rank, world


def shard_init_single():
    """A one-rank communicator (plumbing tests on a single GPU)."""
    lib = load(require_gpu=True)
    buf = (C.c_ubyte * 128)()
    check(lib.c21cm_shard_unique_id(buf), "c21cm_shard_unique_id")
    check(lib.c21cm_shard_init(0, 1, bytes(buf)), "c21cm_shard_init")


def shard_emulate(rank: int, world: int, mailbox):
    """Test hook (c21cm_shard_emulate): run the ranks of a world > 1 job one after the other in
    this process against the device ``mailbox`` (uint8 CUDA tensor, zeroed per round)."""
    lib = load(require_gpu=True)
    lib.c21cm_shard_emulate.restype = C.c_int
    check(lib.c21cm_shard_emulate(C.c_int(rank), C.c_int(world), C.c_void_p(mailbox.data_ptr()),
                                  C.c_size_t(mailbox.numel())), "c21cm_shard_emulate")


def shard_finalize():
    load().c21cm_shard_finalize()


def ionize_sharded(spec: S.IonizeSpec, density, n_ion=None, xe=None, Tneutral=None,
                   prev_z_reion=None, buffers: IonizeBuffers | None = None, stream=None,
                   prev_nrec=None, whalo_sfr=None, broadcast=False):
    """One ComputeIonizedBox pass with the R loop sharded over the ranks of the library's RCCL
    communicator (c21cm_ionize_sharded), all inside the C library: shard phase, exchange, finish.
    Where the finish phase runs by cell slabs (``shard_slab_supported``) every rank finishes its slab
    and holds the complete scalars; otherwise the owner rank finishes.  Returns (buffers, box_struct,
    report); ``broadcast``: True / 1 whole boxes on every rank, 2 the whole neutral-fraction box on every
    rank (slab finish, device arrays), False / 0 a rank's slab / the owner's box, None the library's
    default (c21cm_shard_output_mode)."""
    if buffers is None:
        buffers = IonizeBuffers(density, need_nion=spec.fcoll_mode != 0,
                                minimize_memory=bool(spec.minimize_memory),
                                recomb_model=spec.recomb_model)
    pf, prev, ts, hb = _input_structs(density, n_ion, xe, Tneutral, prev_z_reion, prev_nrec,
                                      whalo_sfr)
    box = buffers.struct()
    rep = S.IonizeReport()
    lib = load()
    lib.c21cm_ionize_sharded.restype = C.c_int
    check(lib.c21cm_ionize_sharded(C.byref(spec), C.byref(pf), C.byref(prev), C.byref(ts),
                                   C.byref(hb), C.byref(box), C.byref(rep),
                                   C.c_int(-1 if broadcast is None else int(broadcast)),
                                   _stream(stream)),
          "c21cm_ionize_sharded")
    return buffers, box, rep


IC_FIELDS = ("lowres_density", "lowres_vx", "lowres_vy", "lowres_vz", "lowres_vx_2LPT",
             "lowres_vy_2LPT", "lowres_vz_2LPT", "hires_density", "hires_vx", "hires_vy",
             "hires_vz", "hires_vx_2LPT", "hires_vy_2LPT", "hires_vz_2LPT", "lowres_vcb")


def ics_struct(ics: dict) -> S.InitialConditionsStruct:
    """InitialConditions struct over a dict of numpy arrays / torch tensors (missing -> NULL)."""
    return S.InitialConditionsStruct(**{k: _fptr(ics.get(k)) for k in IC_FIELDS})


def perturb_grids(spec: S.PerturbSpec, ics: dict, stream=None) -> dict:
    """ComputePerturbedField grid algorithm on the MI355X.

    Outputs live where ``ics['hires_density']`` (or lowres_density for LINEAR) lives.
    Shapes as py21cmfast's PerturbedField.new (reference: wrapper/outputs.py:689-719).
    """
    ref = ics.get("hires_density")
    if ref is None:
        ref = ics["lowres_density"]
    lo = (spec.hii_dim, spec.hii_dim, spec.hii_dim_z)

    def new():
        if _is_torch(ref):
            import torch

            return torch.zeros(lo, dtype=torch.float32, device=ref.device)
        return np.zeros(lo, np.float32)

    out = {"density": new(), "velocity_z": new()}
    if spec.keep_3d_velocities:
        out["velocity_x"] = new()
        out["velocity_y"] = new()
    pf = S.PerturbedFieldStruct(**{k: _fptr(v) for k, v in out.items()})
    icss = ics_struct(ics)
    check(load().c21cm_perturb_grids(C.byref(spec), C.byref(icss), C.byref(pf), _stream(stream)),
          "c21cm_perturb_grids")
    return out


def new_ics_arrays(spec: S.IcsSpec, device=None) -> dict:
    """Zeroed arrays as ``InitialConditions.new`` allocates them
    (reference: src/py21cmfast/wrapper/outputs.py:534-581)."""
    lo = (spec.hii_dim, spec.hii_dim, spec.hii_dim_z)
    hi = (spec.dim, spec.dim, spec.dim_z)

    def new(shape):
        if device is not None:
            import torch

            return torch.zeros(shape, dtype=torch.float32, device=device)
        return np.zeros(shape, np.float32)

    ics = {"hires_density": new(hi), "lowres_density": new(lo)}
    shape = hi if spec.perturb_on_high_res else lo
    pre = "hires" if spec.perturb_on_high_res else "lowres"
    for ax in "xyz":
        ics[f"{pre}_v{ax}"] = new(shape)
        if spec.perturb_algorithm == 2:
            ics[f"{pre}_v{ax}_2LPT"] = new(shape)
    return ics


def ics_grids(spec: S.IcsSpec, ics: dict | None = None, device=None, stream=None) -> dict:
    """ComputeInitialConditions grid algorithm on the MI355X; fills and returns ``ics``."""
    if ics is None:
        ics = new_ics_arrays(spec, device)
    icss = ics_struct(ics)
    check(load().c21cm_ics_grids(C.byref(spec), C.byref(icss), _stream(stream)), "c21cm_ics_grids")
    return ics


def brightness_grids(spec: S.BrightnessSpec, density, neutral_fraction, spin_temperature=None,
                     stream=None) -> dict:
    """ComputeBrightnessTemp sweep on the MI355X (reference:
    src/py21cmfast/src/BrightnessTemperatureBox.c:22-105).  Outputs live where ``density`` lives.
    Returns dict(brightness_temp[, tau_21], mean)."""
    out = {"brightness_temp": _new_like(density, 0.0)}
    if spec.use_ts_fluct:
        out["tau_21"] = _new_like(density, 0.0)
    mean = C.c_double()
    check(load().c21cm_brightness_grids(C.byref(spec), _vptr(density), _vptr(neutral_fraction),
                                        _vptr(spin_temperature), _vptr(out["brightness_temp"]),
                                        _vptr(out.get("tau_21")), C.byref(mean), _stream(stream)),
          "c21cm_brightness_grids")
    out["mean"] = mean.value
    return out


def halobox_grids(spec: S.HaloBoxSpec, ics: dict, with_whalo=False, with_xray=False,
                  stream=None) -> dict:
    """ComputeHaloBox's grids on the MI355X (reference: src/py21cmfast/src/HaloBox.c:302-436,518-560,
    map_mass.c:214-476): the integrated branch and, when ``spec.halos`` / ``spec.halo_consts`` point at a
    catalogue (structs.halo_catalog, structs.HaloConsts), the halos deposited first
    (``spec.skip_integral``: halos only).  Outputs live where the source density lives.
    Returns dict(n_ion, halo_sfr[, whalo_sfr][, halo_xray][, halo_sfr_mini])."""
    ref = ics["hires_density" if spec.perturb_on_high_res else "lowres_density"]
    lo = (spec.hii_dim, spec.hii_dim, spec.hii_dim_z)

    def new():
        if _is_torch(ref):
            import torch

            return torch.zeros(lo, dtype=torch.float32, device=ref.device)
        return np.zeros(lo, np.float32)

    out = {"n_ion": new(), "halo_sfr": new()}
    if with_whalo:
        out["whalo_sfr"] = new()
    if with_xray:  # needs spec.ln_xray_table (USE_TS_FLUCT: the X-ray emissivity grid)
        out["halo_xray"] = new()
    if spec.use_mini_halos:  # the molecularly cooled star formation (HaloBox.c:271-277)
        out["halo_sfr_mini"] = new()
    hb = S.HaloBoxStruct(**{k: _fptr(v) for k, v in out.items()})
    icss = ics_struct(ics)
    check(load().c21cm_halobox_grids(C.byref(spec), C.byref(icss), C.byref(hb), _stream(stream)),
          "c21cm_halobox_grids")
    return out


def grid_minmax(values, stream=None):
    mm = (C.c_double * 2)()
    n = values.numel() if _is_torch(values) else values.size
    check(load().c21cm_grid_minmax(_vptr(values), C.c_size_t(n), mm, _stream(stream)),
          "c21cm_grid_minmax")
    return mm[0], mm[1]


def halobox_turnovers(spec: S.MturnSpec, m_turn, below_z_heat_max, n_threads, prev_G12,
                      prev_z_reion, J_21_LW, vcb=None, like=None, stream=None):
    """get_log10_turnovers (HaloBox.c:465-516) on the MI355X: (log10 M_turn,a, log10 M_turn,m,
    (<a>, <m>)); grids allocated like ``like`` (default: ``J_21_LW``)."""
    ref = J_21_LW if like is None else like
    a, m = _new_like(ref, 0.0), _new_like(ref, 0.0)
    ave = (C.c_double * 2)()
    lib = load()
    lib.c21cm_halobox_turnovers.restype = C.c_int
    lib.c21cm_halobox_turnovers.argtypes = [C.POINTER(S.MturnSpec), C.c_double, C.c_int, C.c_int] + [
        C.c_void_p] * 6 + [C.POINTER(C.c_double), C.c_void_p]
    check(lib.c21cm_halobox_turnovers(C.byref(spec), float(m_turn), int(below_z_heat_max),
                                      int(n_threads), _vptr(prev_G12), _vptr(prev_z_reion),
                                      _vptr(J_21_LW), _vptr(vcb), _vptr(a), _vptr(m), ave,
                                      _stream(stream)), "c21cm_halobox_turnovers")
    return a, m, (ave[0], ave[1])


def fill_Rbox_grids(spec: S.RboxSpec, field, stream=None) -> dict:
    """prepare_filter_boxes + fill_Rbox_table on the MI355X (reference:
    src/py21cmfast/src/SpinTemperatureBox.c:502-520,560-636).  Returns dict(result [n_R, ...],
    min, average, max [n_R]); result lives where ``field`` lives."""
    n_R = spec.n_R
    shape = (n_R,) + tuple(field.shape)
    if _is_torch(field):
        import torch

        result = torch.zeros(shape, dtype=torch.float32, device=field.device)
    else:
        result = np.zeros(shape, np.float32)
    mn, av, mx = ((C.c_double * n_R)() for _ in range(3))
    check(load().c21cm_fill_Rbox_grids(C.byref(spec), _vptr(field), _vptr(result), mn, av, mx,
                                       _stream(stream)), "c21cm_fill_Rbox_grids")
    return {"result": result, "min": np.array(mn[:]), "average": np.array(av[:]),
            "max": np.array(mx[:])}


def annular_filter_grids(spec: S.AnnularSpec, inputs, stream=None) -> dict:
    """one_annular_filter for ``spec.n_grids`` grids of one shell on the MI355X (reference:
    src/py21cmfast/src/SpinTemperatureBox.c:642-742).  Returns dict(outputs [list], u_avg, f_avg)."""
    n = spec.n_grids
    assert len(inputs) == n
    outputs = [_new_like(a, 0.0) for a in inputs]
    in_p = (C.c_void_p * n)(*[_vptr(a).value if _is_torch(a) else a.ctypes.data for a in inputs])
    out_p = (C.c_void_p * n)(*[_vptr(a).value if _is_torch(a) else a.ctypes.data for a in outputs])
    u, f = (C.c_double * n)(), (C.c_double * n)()
    check(load().c21cm_annular_filter_grids(C.byref(spec), in_p, out_p, u, f, _stream(stream)),
          "c21cm_annular_filter_grids")
    return {"outputs": outputs, "u_avg": np.array(u[:]), "f_avg": np.array(f[:])}


TS_FIELDS = ("spin_temperature", "kinetic_temp_neutral", "xray_ionised_fraction")


def ts_grids(spec: S.TsSpec, density, previous: dict, source: dict | None = None,
             filtered_density=None, stream=None) -> dict:
    """The per-cell part of ComputeTsBox on the MI355X (reference:
    src/py21cmfast/src/SpinTemperatureBox.c:1387-1946 from the first cell loop on): the R loop
    over the source grids (``source``: filtered_sfr / filtered_xray [n_step, ...], Lagrangian
    models) or over the filtered densities (``filtered_density`` [n_step, ...] with the spec's
    SFRD tables), then the x_e / T_k update and T_s of every cell.  ``previous``: the three boxes
    of the previous snapshot.  Outputs live where ``density`` lives; ``report`` holds box means.
    With ``spec.use_mini_halos`` (2-D tables and ``spec.filtered_log10_mcrit`` set by the caller)
    the result also holds ``J_21_LW``."""
    out = {k: _new_like(density, 0.0) for k in TS_FIELDS}
    prev = S.TsBoxStruct(**{k: _fptr(previous[k]) for k in TS_FIELDS})
    box = S.TsBoxStruct(**{k: _fptr(out[k]) for k in TS_FIELDS})
    if spec.use_mini_halos:
        out["J_21_LW"] = _new_like(density, 0.0)
        box.J_21_LW = _fptr(out["J_21_LW"])
    src = S.XraySourceBoxStruct(**{k: _fptr(v) for k, v in (source or {}).items()})
    rep = S.TsReport()
    check(load().c21cm_ts_grids(C.byref(spec), _vptr(density), C.byref(prev), C.byref(src),
                                _vptr(filtered_density), C.byref(box), C.byref(rep),
                                _stream(stream)), "c21cm_ts_grids")
    out["report"] = rep
    return out


def ts_mcrit_grid(spec: S.MturnSpec, m_turn: float, J_21_LW, vcb=None, stream=None):
    """log10 of the Lyman-Werner turnover mass per cell from the previous TsBox's J_21_LW
    (prepare_filter_boxes, SpinTemperatureBox.c:535-565); allocated like ``J_21_LW``."""
    out = _new_like(J_21_LW, 0.0)
    lib = load()
    lib.c21cm_ts_mcrit_grid.restype = C.c_int
    lib.c21cm_ts_mcrit_grid.argtypes = [C.POINTER(S.MturnSpec), C.c_double, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    check(lib.c21cm_ts_mcrit_grid(C.byref(spec), float(m_turn), _vptr(J_21_LW), _vptr(vcb),
                                  _vptr(out), _stream(stream)), "c21cm_ts_mcrit_grid")
    return out


def ts_first_grids(spec: S.TsFirstSpec, density, stream=None) -> dict:
    """init_first_Ts (SpinTemperatureBox.c:892-927) on the MI355X."""
    out = {k: _new_like(density, 0.0) for k in TS_FIELDS}
    box = S.TsBoxStruct(**{k: _fptr(out[k]) for k in TS_FIELDS})
    check(load().c21cm_ts_first_grids(C.byref(spec), _vptr(density), C.byref(box), _stream(stream)),
          "c21cm_ts_first_grids")
    return out
