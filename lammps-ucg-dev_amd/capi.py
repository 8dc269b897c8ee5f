"""ctypes binding of libucg_hip.so (the C ABI in include/ucg_hip.h).

This is plumbing for tests, bench.py and smoke(): it adds nothing to the path.  There is
no CPU fallback -- if the HIP library is missing or no GPU is visible, calls fail loudly.
The classes mirror the reference's plugin interface for this path (same style names,
same command arguments, same hooks):

    Pair("table_ucgld" | "table_ucg_bethe" | "table_ucg_bethe_density")
        .settings(args) .coeff(args) .init(T) .compute(eflag, vflag)      -- LAMMPS Pair
    Context.fix_nve_ucgld_* / fix_ucgld_langevin_* / fix_ucgstate_*      -- LAMMPS Fix hooks
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libucg_hip.so")
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_ll_p = C.POINTER(C.c_longlong)

STYLE_IDS = {"table_ucgld": 0, "table_ucg_bethe": 1, "table_ucg_bethe_density": 2}
ORIENT_BIT = 29
NEIGHMASK = 0x1FFFFFFF

# every symbol include/ucg_hip.h declares (checked by the CPU test-suite against the .so)
SYMBOLS = [
    "ucg_abi_version", "ucg_ctx_create", "ucg_ctx_destroy", "ucg_last_error", "ucg_ctx_set_stream",
    "ucg_ctx_synchronize", "ucg_ctx_set_units", "ucg_ctx_set_option", "ucg_selftest_div", "ucg_selftest_stream",
    "ucg_pair_create", "ucg_pair_create_host", "ucg_pair_last_error", "ucg_pair_destroy", "ucg_pair_settings", "ucg_pair_coeff", "ucg_pair_init",
    "ucg_pair_cut", "ucg_pair_cutforce", "ucg_pair_gather_slots", "ucg_pair_sum_fixed", "ucg_pair_single", "ucg_pair_table_count",
    "ucg_pair_table_params", "ucg_pair_table_array", "ucg_pair_tabindex", "ucg_pair_compute",
    "ucg_pair_check_errors",
    "ucg_atoms_upload", "ucg_atoms_upload_comm", "ucg_atoms_upload_owned", "ucg_atoms_download", "ucg_atoms_counts", "ucg_ghosts_upload", "ucg_force_clear",
    "ucg_neigh_upload_full", "ucg_domain_set", "ucg_neigh_rebuild", "ucg_halo_forward", "ucg_neigh_download",
    "ucg_ghosts_download",
    "ucg_fix_nve_initial", "ucg_fix_nve_final",
    "ucg_fix_nve_wall_hard_set", "ucg_fix_nve_wall_hard_initial", "ucg_fix_nve_wall_hard_final",
    "ucg_fix_nve_wall_hard_post_force",
    "ucg_pair_compute_part", "ucg_pair_density_phase", "ucg_pair_density_buffer", "ucg_halo_aux_pack", "ucg_halo_aux_unpack",
    "ucg_atoms_upload_molecule", "ucg_atoms_download_molecule", "ucg_fix_cluster_switch_create",
    "ucg_fix_cluster_switch_check_cluster", "ucg_fix_cluster_switch_attempt_switch", "ucg_fix_cluster_switch_maxmol",
    "ucg_fix_cluster_switch_array", "ucg_fix_cluster_switch_vector", "ucg_halo_molmask_pack", "ucg_halo_molmask_unpack",
    "ucg_fix_cluster_switch_scalars", "ucg_fix_cluster_switch_set_scalars", "ucg_fix_cluster_switch_set_array",
    "ucg_fix_cluster_switch_sweep", "ucg_fix_cluster_switch_finalize", "ucg_fix_cluster_switch_attempt_local",
    "ucg_fix_cluster_switch_attempt_apply", "ucg_fix_cluster_switch_due", "ucg_fix_cluster_switch_advance",
    "ucg_md_set_timestep",
    "ucg_fix_langevin_create", "ucg_fix_langevin_init", "ucg_fix_langevin_init_from_ucgml",
    "ucg_fix_langevin_post_force", "ucg_fix_langevin_end_of_step", "ucg_fix_langevin_t_target",
    "ucg_fix_ucgstate_create", "ucg_fix_ucgstate_post_force",
    "ucg_decomp_set", "ucg_record_bytes", "ucg_exchange_count", "ucg_exchange_pack", "ucg_exchange_unpack",
    "ucg_border_count", "ucg_border_pack", "ucg_border_unpack", "ucg_halo_pack", "ucg_halo_unpack", "ucg_decide_local",
    "ucg_ranmars_fill",
    "ucg_md_attach", "ucg_md_post_fused", "ucg_md_pair_post", "ucg_md_setup", "ucg_md_run", "ucg_md_info", "ucg_md_thermo",
    "ucg_profile_enable", "ucg_profile_read",
    "ucg_comm_attach", "ucg_comm_rccl_unique_id", "ucg_comm_attach_rccl", "ucg_comm_detach", "ucg_comm_info",
    "ucg_comm_allreduce_f64", "ucg_pair_density_aux_download", "ucg_pair_density_aux_upload",
    "ucg_device_count", "ucg_selftest_div_core", "ucg_selftest_sqrt_core", "ucg_comm_attach_host", "ucg_comm_transport", "ucg_fix_langevin_reset_target", "ucg_fix_langevin_reset_dt",
    "ucg_fix_langevin_set_bias", "ucg_md_run_until", "ucg_md_set_window", "ucg_atoms_download_mask",
    "ucg_ghosts_upload_images", "ucg_host_bind", "ucg_host_modified", "ucg_host_sync", "ucg_host_status", "ucg_verlet_hooks_run",
]

# communicator callbacks of a decomposed run (include/ucg_hip.h: ucg_comm_ops)
CB_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p)
CB_ALLTOALL_LL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))
CB_ALLREDUCE_LL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_longlong), C.c_int, C.c_int)
CB_ALLREDUCE_F64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)


class CommOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("alltoallv", CB_ALLTOALLV),
                ("alltoall_ll", CB_ALLTOALL_LL), ("allreduce_ll", CB_ALLREDUCE_LL), ("allreduce_f64", CB_ALLREDUCE_F64)]


class RcclId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


class UcgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[ucg_hip error {code}] {msg}")
        self.code = code
        self.msg = msg


def lib():
    """Load libucg_hip.so (built in-tree by __graft_entry__.build()).  No fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # UCG_HIP_LIBRARY: another build of the same sources (tools/asan_cpu.sh: host code under AddressSanitizer / UBSan)
    path = os.environ.get("UCG_HIP_LIBRARY", LIB_PATH)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the UCG hot path.")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ucg_abi_version.restype = C.c_int
    L.ucg_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ucg_ctx_destroy.argtypes = [vp]
    L.ucg_ctx_destroy.restype = None
    L.ucg_last_error.argtypes = [vp]
    L.ucg_last_error.restype = C.c_char_p
    L.ucg_ctx_set_stream.argtypes = [vp, vp]
    L.ucg_ctx_synchronize.argtypes = [vp]
    L.ucg_ctx_set_units.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, c_double_p]
    L.ucg_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.ucg_selftest_div.argtypes = [vp, C.c_double, C.c_longlong, C.c_int, c_ll_p]
    L.ucg_selftest_stream.argtypes = [vp, C.c_longlong, C.c_int, C.c_int]
    L.ucg_pair_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.ucg_pair_create_host.argtypes = [C.c_int, C.c_double, C.POINTER(vp)]
    L.ucg_pair_last_error.argtypes = [vp]
    L.ucg_pair_last_error.restype = C.c_char_p
    L.ucg_pair_destroy.argtypes = [vp]
    L.ucg_pair_destroy.restype = None
    L.ucg_pair_settings.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p)]
    L.ucg_pair_coeff.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_char_p)]
    L.ucg_pair_init.argtypes = [vp, C.c_int, C.c_double]
    L.ucg_pair_cut.argtypes = [vp, C.c_int, C.c_int]
    L.ucg_pair_cut.restype = C.c_double
    L.ucg_pair_gather_slots.argtypes = [vp]
    L.ucg_pair_sum_fixed.argtypes = [vp]
    L.ucg_pair_cutforce.argtypes = [vp]
    L.ucg_pair_cutforce.restype = C.c_double
    L.ucg_pair_single.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, c_double_p, c_double_p]
    L.ucg_pair_table_count.argtypes = [vp]
    L.ucg_pair_table_params.argtypes = [vp, C.c_int, c_double_p]
    L.ucg_pair_table_array.argtypes = [vp, C.c_int, C.c_char_p, c_double_p, C.c_int]
    L.ucg_pair_tabindex.argtypes = [vp, c_int_p, C.c_int]
    L.ucg_pair_compute.argtypes = [vp, C.c_int, C.c_int, c_double_p, c_double_p]
    L.ucg_pair_check_errors.argtypes = [vp]
    L.ucg_atoms_upload.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p,
                                   c_int_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.ucg_atoms_upload_comm.argtypes = [vp, c_double_p, c_int_p, c_double_p, c_double_p]
    L.ucg_atoms_upload_owned.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_int_p, c_int_p, c_double_p,
                                         c_double_p, c_double_p, c_double_p, c_double_p]
    L.ucg_atoms_download.argtypes = [vp, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p,
                                     c_int_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
    L.ucg_atoms_counts.argtypes = [vp, c_int_p, c_int_p]
    L.ucg_ghosts_upload.argtypes = [vp, c_int_p, C.c_int]
    L.ucg_force_clear.argtypes = [vp]
    L.ucg_neigh_upload_full.argtypes = [vp, C.c_int, c_int_p, c_ll_p, c_int_p]
    L.ucg_domain_set.argtypes = [vp, c_double_p, c_double_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    L.ucg_neigh_rebuild.argtypes = [vp]
    L.ucg_halo_forward.argtypes = [vp]
    L.ucg_neigh_download.argtypes = [vp, c_int_p, c_int_p, c_ll_p, c_int_p, C.c_longlong, c_ll_p]
    L.ucg_ghosts_download.argtypes = [vp, c_int_p, c_int_p, C.c_int]
    L.ucg_fix_nve_initial.argtypes = [vp, C.c_int]
    L.ucg_fix_nve_final.argtypes = [vp, C.c_int]
    L.ucg_fix_nve_wall_hard_set.argtypes = [vp, C.c_int, C.c_double]
    L.ucg_fix_nve_wall_hard_initial.argtypes = [vp, C.c_int]
    L.ucg_fix_nve_wall_hard_final.argtypes = [vp, C.c_int]
    L.ucg_fix_nve_wall_hard_post_force.argtypes = [vp, C.c_int]
    L.ucg_atoms_upload_molecule.argtypes = [vp, c_int_p]
    L.ucg_pair_compute_part.argtypes = [vp, C.c_int]
    L.ucg_pair_density_phase.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]
    L.ucg_pair_density_buffer.argtypes = [vp, C.c_int]
    L.ucg_pair_density_buffer.restype = C.c_void_p
    L.ucg_halo_aux_pack.argtypes = [vp, C.c_void_p, C.c_void_p]
    L.ucg_halo_aux_unpack.argtypes = [vp, C.c_void_p, C.c_void_p]
    L.ucg_atoms_download_molecule.argtypes = [vp, c_int_p]
    L.ucg_fix_cluster_switch_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_char_p,
                                                C.c_char_p]
    L.ucg_fix_cluster_switch_check_cluster.argtypes = [vp]
    L.ucg_fix_cluster_switch_attempt_switch.argtypes = [vp]
    L.ucg_fix_cluster_switch_maxmol.argtypes = [vp]
    L.ucg_fix_cluster_switch_array.argtypes = [vp, C.c_int, c_int_p]
    L.ucg_fix_cluster_switch_vector.argtypes = [vp, c_double_p]
    L.ucg_halo_molmask_pack.argtypes = [vp, C.c_void_p]
    L.ucg_halo_molmask_unpack.argtypes = [vp, C.c_void_p]
    L.ucg_fix_cluster_switch_scalars.argtypes = [vp, c_ll_p]
    L.ucg_fix_cluster_switch_set_scalars.argtypes = [vp, C.c_longlong, C.c_longlong, C.c_longlong]
    L.ucg_fix_cluster_switch_set_array.argtypes = [vp, C.c_int, c_int_p]
    L.ucg_fix_cluster_switch_sweep.argtypes = [vp, C.c_int, c_int_p]
    L.ucg_fix_cluster_switch_finalize.argtypes = [vp]
    L.ucg_fix_cluster_switch_attempt_local.argtypes = [vp]
    L.ucg_fix_cluster_switch_attempt_apply.argtypes = [vp]
    L.ucg_fix_cluster_switch_due.argtypes = [vp, c_int_p, c_int_p]
    L.ucg_fix_cluster_switch_advance.argtypes = [vp]
    L.ucg_md_set_timestep.argtypes = [vp, C.c_longlong]
    L.ucg_fix_langevin_create.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    L.ucg_fix_langevin_init.argtypes = [vp, C.c_int, c_double_p, c_double_p]
    L.ucg_fix_langevin_init_from_ucgml.argtypes = [vp, C.c_int, c_double_p]
    L.ucg_fix_langevin_post_force.argtypes = [vp, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong]
    L.ucg_fix_langevin_end_of_step.argtypes = [vp, C.c_int, c_double_p]
    L.ucg_fix_langevin_t_target.argtypes = [vp]
    L.ucg_fix_langevin_t_target.restype = C.c_double
    L.ucg_fix_ucgstate_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
    L.ucg_fix_ucgstate_post_force.argtypes = [vp]
    L.ucg_decomp_set.argtypes = [vp, c_int_p, C.c_int]
    L.ucg_record_bytes.argtypes = [c_int_p, c_int_p]
    L.ucg_exchange_count.argtypes = [vp, c_ll_p]
    L.ucg_exchange_pack.argtypes = [vp, vp]
    L.ucg_exchange_unpack.argtypes = [vp, vp, C.c_longlong]
    L.ucg_border_count.argtypes = [vp, c_ll_p]
    L.ucg_border_pack.argtypes = [vp, vp]
    L.ucg_border_unpack.argtypes = [vp, vp, C.c_longlong]
    L.ucg_halo_pack.argtypes = [vp, vp]
    L.ucg_halo_unpack.argtypes = [vp, vp]
    L.ucg_decide_local.argtypes = [vp, c_int_p, c_int_p]
    L.ucg_ranmars_fill.argtypes = [vp, C.c_int, C.c_longlong, C.c_int, c_double_p]
    L.ucg_md_attach.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.ucg_md_pair_post.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong]
    L.ucg_md_post_fused.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong]
    L.ucg_md_setup.argtypes = [vp, C.c_longlong]
    L.ucg_md_run.argtypes = [vp, C.c_longlong, C.c_int]
    L.ucg_md_info.argtypes = [vp, c_ll_p]
    L.ucg_md_thermo.argtypes = [vp, c_double_p]
    L.ucg_profile_enable.argtypes = [vp, C.c_int]
    L.ucg_profile_read.argtypes = [vp, c_ll_p, c_double_p, C.c_int]
    L.ucg_comm_attach.argtypes = [vp, C.POINTER(CommOps)]
    L.ucg_comm_rccl_unique_id.argtypes = [C.POINTER(RcclId)]
    L.ucg_comm_attach_rccl.argtypes = [vp, C.POINTER(RcclId), C.c_int, C.c_int]
    L.ucg_comm_detach.argtypes = [vp]
    L.ucg_comm_info.argtypes = [vp, c_int_p, c_int_p, c_int_p, c_ll_p]
    L.ucg_comm_allreduce_f64.argtypes = [vp, c_double_p, C.c_int, C.c_int]
    L.ucg_pair_density_aux_download.argtypes = [vp, C.c_int, c_double_p, C.c_int, C.c_int]
    L.ucg_pair_density_aux_upload.argtypes = [vp, C.c_int, c_double_p, C.c_int, C.c_int]
    L.ucg_ghosts_upload_images.argtypes = [vp, c_int_p, c_int_p, C.c_int]
    L.ucg_host_bind.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p,
                                c_double_p, c_double_p]
    L.ucg_host_modified.argtypes = [vp, C.c_int]
    L.ucg_host_sync.argtypes = [vp, C.c_int]
    L.ucg_host_status.argtypes = [vp, c_int_p, c_int_p, c_ll_p]
    L.ucg_verlet_hooks_run.argtypes = [vp, vp, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ll_p]
    L.ucg_device_count.argtypes = []
    L.ucg_selftest_div_core.argtypes = [vp, C.c_longlong, C.c_int, c_ll_p]
    L.ucg_selftest_sqrt_core.argtypes = [vp, C.c_longlong, C.c_int, c_ll_p]
    L.ucg_comm_attach_host.argtypes = [vp, C.POINTER(CommOps)]
    L.ucg_comm_transport.argtypes = [vp, c_int_p]
    L.ucg_fix_langevin_reset_target.argtypes = [vp, C.c_double]
    L.ucg_fix_langevin_reset_dt.argtypes = [vp, C.c_int, c_double_p]
    L.ucg_fix_langevin_set_bias.argtypes = [vp, C.c_int]
    L.ucg_md_run_until.argtypes = [vp, C.c_longlong, C.c_int]
    L.ucg_md_set_window.argtypes = [vp, C.c_longlong, C.c_longlong]
    L.ucg_atoms_download_mask.argtypes = [vp, c_int_p]
    _LIB = L
    return L


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


def _argv(args):
    return (C.c_char_p * len(args))(*[str(a).encode() for a in args])


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class Context:
    """One GPU context (one per rank), plus the fix hooks that act on its resident beads."""

    def __init__(self, device: int = -1, boltz=1.0, ftm2v=1.0, mvv2e=1.0, dt=0.002, special_lj=(1.0, 1.0, 1.0, 1.0)):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.ucg_ctx_create(device, C.byref(h))
        if rc:
            raise UcgError(rc, "ucg_ctx_create failed: no usable HIP device (the UCG hot path has no CPU fallback)")
        self.h = h
        self.set_units(boltz, ftm2v, mvv2e, dt, special_lj)

    def close(self):
        if getattr(self, "h", None):
            self.L.ucg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def chk(self, rc):
        if rc:
            raise UcgError(rc, self.L.ucg_last_error(self.h).decode())

    def set_units(self, boltz, ftm2v, mvv2e, dt, special_lj=(1.0, 1.0, 1.0, 1.0)):
        s = _f64(special_lj)
        self.chk(self.L.ucg_ctx_set_units(self.h, boltz, ftm2v, mvv2e, dt, _dp(s)))

    def set_option(self, name, value):
        self.chk(self.L.ucg_ctx_set_option(self.h, name.encode(), int(value)))

    def selftest_div(self, b, seed, n):
        m = C.c_longlong(0)
        self.chk(self.L.ucg_selftest_div(self.h, float(b), int(seed), int(n), C.byref(m)))
        return m.value

    def selftest_div_core(self, seed, n):
        bad = C.c_longlong(0)
        self.chk(self.L.ucg_selftest_div_core(self.h, int(seed), int(n), C.byref(bad)))
        return bad.value

    def selftest_sqrt_core(self, seed, n):
        bad = C.c_longlong(0)
        self.chk(self.L.ucg_selftest_sqrt_core(self.h, int(seed), int(n), C.byref(bad)))
        return bad.value

    def selftest_stream(self, nbytes, wide, repeats=1):
        self.chk(self.L.ucg_selftest_stream(self.h, int(nbytes), int(wide), int(repeats)))

    def set_stream(self, stream_ptr):
        self.chk(self.L.ucg_ctx_set_stream(self.h, C.c_void_p(stream_ptr or 0)))

    def synchronize(self):
        self.chk(self.L.ucg_ctx_synchronize(self.h))

    # ---- atoms
    def atoms_upload(self, nlocal, nghost, ntypes, x, v, type, tag, mask, ucgstate, ucgl, ucgvl, ucgml, ucgp, mass):
        keep = [_f64(x), _f64(v), _i32(type), _i32(tag), _i32(mask), _i32(ucgstate), _f64(ucgl), _f64(ucgvl),
                _f64(ucgml), _f64(ucgp), _f64(mass)]
        self.chk(self.L.ucg_atoms_upload(self.h, nlocal, nghost, ntypes, _dp(keep[0]), _dp(keep[1]), _ip(keep[2]),
                                         _ip(keep[3]), _ip(keep[4]), _ip(keep[5]), _dp(keep[6]), _dp(keep[7]),
                                         _dp(keep[8]), _dp(keep[9]), _dp(keep[10])))

    def upload_beads(self, beads):
        """all beads owned, no ghosts (the device builder makes the periodic images)"""
        self.atoms_upload(beads.n, 0, beads.ntypes, beads.x, beads.v, beads.type, beads.tag, beads.mask,
                          beads.ucgstate, beads.ucgl, beads.ucgvl, beads.ucgml, beads.ucgp, beads.mass)
        if getattr(beads, "molecule", None) is not None:
            self.upload_molecule(beads.molecule)

    def atoms_upload_comm(self, x, ucgstate, ucgl, ucgp):
        k = [_f64(x), _i32(ucgstate), _f64(ucgl), _f64(ucgp)]
        self.chk(self.L.ucg_atoms_upload_comm(self.h, _dp(k[0]), _ip(k[1]), _dp(k[2]), _dp(k[3])))

    def counts(self):
        a, b = C.c_int(0), C.c_int(0)
        self.chk(self.L.ucg_atoms_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def atoms_download(self, with_ghosts=False):
        nl, ng = self.counts()
        nall = nl + (ng if with_ghosts else 0)
        out = dict(
            x=np.zeros((nall, 3)), v=np.zeros((nl, 3)), f=np.zeros((nl, 3)), type=np.zeros(nall, np.int32),
            tag=np.zeros(nall, np.int32), ucgstate=np.zeros(nall, np.int32), num_ucgstates=np.zeros(nl, np.int32),
            ucgl=np.zeros(nall), ucgvl=np.zeros(nl), ucgml=np.zeros(nl), ucgp=np.zeros(nall), ucgforce=np.zeros(nl),
            scores=np.zeros((nl, 2)))
        self.chk(self.L.ucg_atoms_download(
            self.h, 1 if with_ghosts else 0, _dp(out["x"]), _dp(out["v"]), _dp(out["f"]), _ip(out["type"]),
            _ip(out["tag"]), _ip(out["ucgstate"]), _ip(out["num_ucgstates"]), _dp(out["ucgl"]), _dp(out["ucgvl"]),
            _dp(out["ucgml"]), _dp(out["ucgp"]), _dp(out["ucgforce"]), _dp(out["scores"])))
        out["nlocal"], out["nghost"] = nl, ng
        return out

    def ghosts_upload(self, src):
        s = _i32(src)
        self.chk(self.L.ucg_ghosts_upload(self.h, _ip(s), len(s)))

    def ghosts_upload_images(self, src, shift3):
        a, b = _i32(src), _i32(np.ascontiguousarray(shift3).reshape(-1))
        self.chk(self.L.ucg_ghosts_upload_images(self.h, _ip(a), _ip(b), len(a)))

    def force_clear(self):
        self.chk(self.L.ucg_force_clear(self.h))

    # ---- neighbour lists / domain
    def neigh_upload_full(self, numneigh, first, neigh):
        nn, fi, ne = _i32(numneigh), np.ascontiguousarray(first, dtype=np.int64), _i32(neigh)
        if ne.size == 0:
            ne = np.zeros(1, np.int32)
        self.chk(self.L.ucg_neigh_upload_full(self.h, len(nn), _ip(nn), fi.ctypes.data_as(c_ll_p), _ip(ne)))

    def domain_set(self, boxlo, boxhi, cutforce, skin, every=1, delay=0, check=1):
        lo, hi = _f64(boxlo), _f64(boxhi)
        self.chk(self.L.ucg_domain_set(self.h, _dp(lo), _dp(hi), cutforce, skin, every, delay, check))

    def neigh_rebuild(self):
        self.chk(self.L.ucg_neigh_rebuild(self.h))

    def halo_forward(self):
        self.chk(self.L.ucg_halo_forward(self.h))

    def neigh_download(self):
        inum, total = C.c_int(0), C.c_longlong(0)
        self.chk(self.L.ucg_neigh_download(self.h, C.byref(inum), None, None, None, 0, C.byref(total)))
        n, tot = inum.value, total.value
        nn = np.zeros(max(n, 1), np.int32)
        fi = np.zeros(max(n, 1), np.int64)
        ne = np.zeros(max(tot, 1), np.int32)
        self.chk(self.L.ucg_neigh_download(self.h, C.byref(inum), _ip(nn), fi.ctypes.data_as(c_ll_p), _ip(ne), tot,
                                           C.byref(total)))
        return np.arange(n, dtype=np.int32), nn[:n], fi[:n], ne[:tot]

    def ghosts_download(self):
        nl, ng = self.counts()
        src = np.zeros(max(ng, 1), np.int32)
        sh = np.zeros((max(ng, 1), 3), np.int32)
        self.chk(self.L.ucg_ghosts_download(self.h, _ip(src), _ip(sh), ng))
        return src[:ng], sh[:ng]

    def atoms_upload_owned(self, **fields):
        """refresh owned-bead fields from host arrays (drop-in fix hooks): x, v, f, ucgstate, num_ucgstates,
        ucgl, ucgvl, ucgp, ucgforce, scores"""
        order = ["x", "v", "f", "ucgstate", "num_ucgstates", "ucgl", "ucgvl", "ucgp", "ucgforce", "scores"]
        keep, args = [], []
        for k in order:
            a = fields.pop(k, None)
            if a is None:
                args.append(None)
            elif k in ("ucgstate", "num_ucgstates"):
                a = _i32(a); keep.append(a); args.append(_ip(a))
            else:
                a = _f64(a); keep.append(a); args.append(_dp(a))
        if fields:
            raise TypeError(f"unknown fields {sorted(fields)}")
        self.chk(self.L.ucg_atoms_upload_owned(self.h, *args))

    # ---- fix nve/ucgld
    def fix_nve_ucgld_initial_integrate(self, groupbit=1):
        self.chk(self.L.ucg_fix_nve_initial(self.h, groupbit))

    def fix_nve_ucgld_final_integrate(self, groupbit=1):
        self.chk(self.L.ucg_fix_nve_final(self.h, groupbit))

    # ---- fix cluster_switch
    def upload_molecule(self, molecule):
        m = _i32(molecule)
        self.chk(self.L.ucg_atoms_upload_molecule(self.h, _ip(m)))

    def download_molecule(self):
        nl, _ = self.counts()
        out = np.zeros(nl, dtype=np.int32)
        self.chk(self.L.ucg_atoms_download_molecule(self.h, _ip(out)))
        return out

    def fix_cluster_switch(self, mol_seed, mol_offset, cutoff, seed, switch_freq, rate_file, contact_file, groupbit=1):
        self.chk(self.L.ucg_fix_cluster_switch_create(self.h, groupbit, int(mol_seed), int(mol_offset), float(cutoff),
                                                      int(seed), int(switch_freq), rate_file.encode(),
                                                      contact_file.encode()))

    def fix_cluster_switch_check_cluster(self):
        self.chk(self.L.ucg_fix_cluster_switch_check_cluster(self.h))

    def fix_cluster_switch_attempt_switch(self):
        self.chk(self.L.ucg_fix_cluster_switch_attempt_switch(self.h))

    def fix_cluster_switch_arrays(self):
        n = self.L.ucg_fix_cluster_switch_maxmol(self.h) + 1
        out = {}
        for w, k in enumerate(("mol_cluster", "mol_state", "mol_restrict", "mol_accept")):
            a = np.zeros(max(n, 1), dtype=np.int32)
            self.chk(self.L.ucg_fix_cluster_switch_array(self.h, w, _ip(a)))
            out[k] = a[:n]
        return out

    # decomposed runs (multi.RankSim drives these; see include/ucg_hip.h)
    def halo_molmask_pack(self, sendbuf):
        self.chk(self.L.ucg_halo_molmask_pack(self.h, sendbuf))

    def halo_molmask_unpack(self, recvbuf):
        self.chk(self.L.ucg_halo_molmask_unpack(self.h, recvbuf))

    def cs_scalars(self):
        out = np.zeros(3, dtype=np.int64)
        self.chk(self.L.ucg_fix_cluster_switch_scalars(self.h, out.ctypes.data_as(c_ll_p)))
        return out

    def cs_set_scalars(self, maxmol, nspm, nmolatoms):
        self.chk(self.L.ucg_fix_cluster_switch_set_scalars(self.h, int(maxmol), int(nspm), int(nmolatoms)))

    def cs_array(self, which):
        n = self.L.ucg_fix_cluster_switch_maxmol(self.h) + 1
        a = np.zeros(max(n, 1), dtype=np.int32)
        self.chk(self.L.ucg_fix_cluster_switch_array(self.h, which, _ip(a)))
        return a[:n]

    def cs_set_array(self, which, arr):
        a = _i32(arr)
        self.chk(self.L.ucg_fix_cluster_switch_set_array(self.h, which, _ip(a)))

    def cs_sweep(self, begin):
        ch = C.c_int(0)
        self.chk(self.L.ucg_fix_cluster_switch_sweep(self.h, int(begin), C.byref(ch)))
        return ch.value

    def cs_finalize(self):
        self.chk(self.L.ucg_fix_cluster_switch_finalize(self.h))

    def cs_attempt_local(self):
        self.chk(self.L.ucg_fix_cluster_switch_attempt_local(self.h))

    def cs_attempt_apply(self):
        self.chk(self.L.ucg_fix_cluster_switch_attempt_apply(self.h))

    def cs_due(self):
        f, sw = C.c_int(0), C.c_int(0)
        self.chk(self.L.ucg_fix_cluster_switch_due(self.h, C.byref(f), C.byref(sw)))
        return bool(f.value), bool(sw.value)

    def cs_advance(self):
        self.chk(self.L.ucg_fix_cluster_switch_advance(self.h))

    def md_set_timestep(self, n):
        self.chk(self.L.ucg_md_set_timestep(self.h, int(n)))

    def fix_cluster_switch_vector(self):
        out = np.zeros(7)
        self.chk(self.L.ucg_fix_cluster_switch_vector(self.h, _dp(out)))
        return out

    # ---- fix nve/ucgld/wall/hard [bias_potential [barrier]]
    def fix_nve_ucgld_wall_hard(self, bias_potential=False, barrier=0.1):
        self.chk(self.L.ucg_fix_nve_wall_hard_set(self.h, int(bool(bias_potential)), float(barrier)))

    def fix_nve_ucgld_wall_hard_initial_integrate(self, groupbit=1):
        self.chk(self.L.ucg_fix_nve_wall_hard_initial(self.h, groupbit))

    def fix_nve_ucgld_wall_hard_final_integrate(self, groupbit=1):
        self.chk(self.L.ucg_fix_nve_wall_hard_final(self.h, groupbit))

    def fix_nve_ucgld_wall_hard_post_force(self, groupbit=1):
        self.chk(self.L.ucg_fix_nve_wall_hard_post_force(self.h, groupbit))

    # ---- fix ucgld/langevin
    def fix_ucgld_langevin(self, t_start, t_stop, damp, seed, me=0):
        self.chk(self.L.ucg_fix_langevin_create(self.h, t_start, t_stop, damp, int(seed), me))

    def fix_ucgld_langevin_init(self, ntypes, ucgml_by_type_index):
        ml = _f64(ucgml_by_type_index)
        self.chk(self.L.ucg_fix_langevin_init_from_ucgml(self.h, ntypes, _dp(ml)))

    def fix_ucgld_langevin_post_force(self, ntimestep, beginstep, endstep, groupbit=1):
        self.chk(self.L.ucg_fix_langevin_post_force(self.h, groupbit, ntimestep, beginstep, endstep))

    def fix_ucgld_langevin_end_of_step(self, groupbit=1):
        t = C.c_double(0)
        self.chk(self.L.ucg_fix_langevin_end_of_step(self.h, groupbit, C.byref(t)))
        return t.value

    def fix_ucgld_langevin_t_target(self):
        return self.L.ucg_fix_langevin_t_target(self.h)

    def fix_ucgld_langevin_reset_target(self, t_new):
        self.chk(self.L.ucg_fix_langevin_reset_target(self.h, float(t_new)))

    def fix_ucgld_langevin_reset_dt(self, ntypes, mass_by_type=None):
        m = None if mass_by_type is None else _f64(mass_by_type)
        self.chk(self.L.ucg_fix_langevin_reset_dt(self.h, int(ntypes), None if m is None else _dp(m)))

    def fix_ucgld_langevin_set_bias(self, bias):
        self.chk(self.L.ucg_fix_langevin_set_bias(self.h, int(bool(bias))))

    # ---- fix ucgstate
    def fix_ucgstate(self, mode=None, seed=0, rate=0.01, me=0):
        """mode: None (plain) | "ld" | "mc" -- `fix ID grp ucgstate [ld | mc seed rate]`"""
        self.chk(self.L.ucg_fix_ucgstate_create(self.h, 1 if mode == "ld" else 0, 1 if mode == "mc" else 0, int(seed),
                                                float(rate), me))

    def fix_ucgstate_post_force(self):
        self.chk(self.L.ucg_fix_ucgstate_post_force(self.h))

    # ---- multi-rank support (count / pack / unpack; the caller moves the buffers)
    def decomp_set(self, procgrid, me):
        g = _i32(procgrid)
        self.chk(self.L.ucg_decomp_set(self.h, _ip(g), int(me)))
        self._world = int(g[0] * g[1] * g[2])

    def halo_aux_pack(self, field_dev, sendbuf):
        self.chk(self.L.ucg_halo_aux_pack(self.h, field_dev, sendbuf))

    def halo_aux_unpack(self, field_dev, recvbuf):
        self.chk(self.L.ucg_halo_aux_unpack(self.h, field_dev, recvbuf))

    def record_bytes(self):
        a, b = C.c_int(0), C.c_int(0)
        self.L.ucg_record_bytes(C.byref(a), C.byref(b))
        return a.value, b.value

    def exchange_count(self):
        out = np.zeros(self._world, np.int64)
        self.chk(self.L.ucg_exchange_count(self.h, out.ctypes.data_as(c_ll_p)))
        return out

    def exchange_pack(self, ptr):
        self.chk(self.L.ucg_exchange_pack(self.h, C.c_void_p(ptr)))

    def exchange_unpack(self, ptr, nrecv):
        self.chk(self.L.ucg_exchange_unpack(self.h, C.c_void_p(ptr), int(nrecv)))

    def border_count(self):
        out = np.zeros(self._world, np.int64)
        self.chk(self.L.ucg_border_count(self.h, out.ctypes.data_as(c_ll_p)))
        return out

    def border_pack(self, ptr):
        self.chk(self.L.ucg_border_pack(self.h, C.c_void_p(ptr)))

    def border_unpack(self, ptr, nrecv):
        self.chk(self.L.ucg_border_unpack(self.h, C.c_void_p(ptr), int(nrecv)))

    def halo_pack(self, ptr):
        self.chk(self.L.ucg_halo_pack(self.h, C.c_void_p(ptr)))

    def halo_unpack(self, ptr):
        self.chk(self.L.ucg_halo_unpack(self.h, C.c_void_p(ptr)))

    def decide_local(self):
        due, flag = C.c_int(0), C.c_int(0)
        self.chk(self.L.ucg_decide_local(self.h, C.byref(due), C.byref(flag)))
        return due.value, flag.value

    # ---- RanMars
    def ranmars_fill(self, seed, skip, n):
        out = np.zeros(max(n, 1))
        self.chk(self.L.ucg_ranmars_fill(self.h, int(seed), int(skip), int(n), _dp(out)))
        return out[:n]

    # ---- resident driver
    def md_attach(self, pair, nve=True, langevin=False, ucgstate=False):
        """nve: False | True (fix nve/ucgld) | "wall" (fix nve/ucgld/wall/hard, see fix_nve_ucgld_wall_hard)"""
        kind = 2 if nve == "wall" else int(bool(nve))
        self.chk(self.L.ucg_md_attach(self.h, pair.h, kind, int(langevin), int(ucgstate)))

    def md_pair_post(self, pair, langevin, ucgstate, nve, ntimestep, beginstep, endstep, groupbit=1):
        """pair force + the fused per-bead hooks (next initial_integrate included) in one launch; False where the
        two-call form has to be used (UCG_ERR_UNSUPPORTED)"""
        rc = self.L.ucg_md_pair_post(self.h, pair.h, int(langevin), int(ucgstate), int(nve), groupbit, int(ntimestep),
                                     int(beginstep), int(endstep))
        if rc == 6:  # UCG_ERR_UNSUPPORTED
            return False
        self.chk(rc)
        return True

    def md_post_fused(self, langevin, ucgstate, nve, fuse_next, ntimestep, beginstep, endstep, groupbit=1):
        self.chk(self.L.ucg_md_post_fused(self.h, int(langevin), int(ucgstate), int(nve), int(fuse_next), groupbit,
                                          ntimestep, beginstep, endstep))

    def md_setup(self, nsteps):
        self.chk(self.L.ucg_md_setup(self.h, nsteps))

    def md_run(self, nsteps, thermo_every=0):
        self.chk(self.L.ucg_md_run(self.h, nsteps, thermo_every))

    def md_run_until(self, nsteps, ev_on_last=False):
        self.chk(self.L.ucg_md_run_until(self.h, int(nsteps), int(bool(ev_on_last))))

    def md_set_window(self, beginstep, endstep):
        self.chk(self.L.ucg_md_set_window(self.h, int(beginstep), int(endstep)))

    def download_mask(self):
        out = np.zeros(max(self.counts()[0], 1), np.int32)
        self.chk(self.L.ucg_atoms_download_mask(self.h, _ip(out)))
        return out[:self.counts()[0]]

    def md_info(self):
        out = np.zeros(16, np.int64)
        self.chk(self.L.ucg_md_info(self.h, out.ctypes.data_as(c_ll_p)))
        keys = ["ntimestep", "nrebuild", "nlocal", "nghost", "list_entries", "pair_error_steps", "maxrow", "pitch",
                "nbx", "nby", "nbz"]
        return dict(zip(keys, [int(v) for v in out[:len(keys)]]))

    def md_thermo(self):
        out = np.zeros(9)
        self.chk(self.L.ucg_md_thermo(self.h, _dp(out)))
        return dict(eng_vdwl=out[0], virial=out[1:7].copy(), lambda_temp=out[7], state1=out[8])

    # ---- communicator of a decomposed run (the rank-level step loop then runs inside ucg_md_setup / ucg_md_run)
    def comm_attach(self, rank, world, alltoallv, alltoall_ll, allreduce_ll, allreduce_f64):
        """caller-provided transport: Python callables wrapped as the C callbacks of ucg_comm_ops"""
        self._comm_cbs = (CB_ALLTOALLV(alltoallv), CB_ALLTOALL_LL(alltoall_ll), CB_ALLREDUCE_LL(allreduce_ll),
                          CB_ALLREDUCE_F64(allreduce_f64))  # keep them alive as long as the context
        self._comm_ops = CommOps(None, int(rank), int(world), *self._comm_cbs)
        self.chk(self.L.ucg_comm_attach(self.h, C.byref(self._comm_ops)))

    @staticmethod
    def rccl_unique_id():
        rid = RcclId()
        if lib().ucg_comm_rccl_unique_id(C.byref(rid)):
            raise UcgError(8, "ncclGetUniqueId failed (librccl not loadable?)")
        return bytes(rid)

    def comm_attach_rccl(self, id_bytes, rank, world):
        rid = RcclId.from_buffer_copy(id_bytes)
        self.chk(self.L.ucg_comm_attach_rccl(self.h, C.byref(rid), int(rank), int(world)))

    def comm_detach(self):
        self.chk(self.L.ucg_comm_detach(self.h))

    def comm_info(self):
        r, w, k, n = C.c_int(0), C.c_int(0), C.c_int(0), C.c_longlong(0)
        self.chk(self.L.ucg_comm_info(self.h, C.byref(r), C.byref(w), C.byref(k), C.byref(n)))
        return dict(rank=r.value, world=w.value, rccl=bool(k.value), nrebuild=n.value)

    def comm_transport(self):
        """what the attached communicator really is (asked of RCCL itself: ncclCommCount / ncclCommCuDevice)"""
        out = np.zeros(4, np.int32)
        self.chk(self.L.ucg_comm_transport(self.h, _ip(out)))
        return dict(rccl=bool(out[0]), rccl_nranks=int(out[1]), rccl_device=int(out[2]), host_staged=bool(out[3]))

    def comm_allreduce_sum(self, values):
        a = _f64(values).copy()
        self.chk(self.L.ucg_comm_allreduce_f64(self.h, _dp(a), len(a), 0))
        return a

    # ---- host mirrors of a drop-in caller (lazy synchronisation: ucg_host_bind / _modified / _sync)
    F_X, F_V, F_F, F_STATE, F_NSTATES, F_UCGL, F_UCGVL, F_UCGP, F_UCGFORCE, F_SCORES, F_ALL = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1023

    def host_bind(self, arrays):
        """arrays: dict with x v f (n,3), ucgstate num_ucgstates (int32), ucgl ucgvl ucgp ucgforce, scores (n,2) of the owned
        atoms -- C-contiguous numpy arrays the caller keeps alive (LAMMPS' atom arrays); None unbinds"""
        if arrays is None:
            self._mirror = None
            z = None
            self.chk(self.L.ucg_host_bind(self.h, z, z, z, z, z, z, z, z, z, z))
            return
        for k in ("x", "v", "f", "ucgl", "ucgvl", "ucgp", "ucgforce", "scores"):
            a = arrays[k]
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], k
        for k in ("ucgstate", "num_ucgstates"):
            assert arrays[k].dtype == np.int32 and arrays[k].flags["C_CONTIGUOUS"], k
        self._mirror = arrays
        a = arrays
        self.chk(self.L.ucg_host_bind(self.h, _dp(a["x"]), _dp(a["v"]), _dp(a["f"]), _ip(a["ucgstate"]), _ip(a["num_ucgstates"]),
                                      _dp(a["ucgl"]), _dp(a["ucgvl"]), _dp(a["ucgp"]), _dp(a["ucgforce"]), _dp(a["scores"])))

    def host_modified(self, mask):
        self.chk(self.L.ucg_host_modified(self.h, int(mask)))

    def host_sync(self, mask=1023):
        self.chk(self.L.ucg_host_sync(self.h, int(mask)))

    def host_status(self):
        d, h = C.c_int(0), C.c_int(0)
        t = np.zeros(2, np.int64)
        self.chk(self.L.ucg_host_status(self.h, C.byref(d), C.byref(h), t.ctypes.data_as(c_ll_p)))
        return dict(device_newer=d.value, host_newer=h.value, uploads=int(t[0]), downloads=int(t[1]))

    def verlet_hooks_run(self, pair, nsteps, nve=True, langevin=False, ucgstate=False, sync_every=0, groupbit=1, sync_on_reneighbour=True):
        """the hooks in upstream Verlet's order, one C-ABI call each (drop-in emulation); returns the statistics"""
        kind = 2 if nve == "wall" else (1 if nve else 0)
        st = np.zeros(4, np.int64)
        self.chk(self.L.ucg_verlet_hooks_run(self.h, pair.h, int(nsteps), kind, int(bool(langevin)), int(bool(ucgstate)),
                                             int(groupbit), int(bool(sync_on_reneighbour)), int(sync_every), st.ctypes.data_as(c_ll_p)))
        return dict(rebuilds=int(st[0]), syncs=int(st[1]), uploads=int(st[2]), downloads=int(st[3]))

    # ---- measurement
    def profile_enable(self, on=True):
        self.chk(self.L.ucg_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        n, ms = C.c_longlong(0), C.c_double(0)
        self.chk(self.L.ucg_profile_read(self.h, C.byref(n), C.byref(ms), int(reset)))
        return n.value, ms.value


class _HostOnly:
    """stands in for a Context when a pair is created without a device (setup half only)"""

    def __init__(self):
        self.L = lib()
        self.h = None
        self.pair_h = None

    def chk(self, rc):
        if rc:
            raise UcgError(rc, self.L.ucg_pair_last_error(self.pair_h).decode())


class Pair:
    """pair_style table_ucgld | table_ucg_bethe | table_ucg_bethe_density on the GPU.

    ``Pair(None, style)`` builds a host-only pair: settings / coeff / init / single and table
    inspection work (that half runs on the host in the reference too); compute() is refused.
    """

    def __init__(self, ctx, style: str, boltz: float = 1.0):
        self.style = style
        h = C.c_void_p()
        if ctx is None:
            self.ctx = _HostOnly()
            rc = self.ctx.L.ucg_pair_create_host(STYLE_IDS[style], boltz, C.byref(h))
            if rc:
                raise UcgError(rc, "ucg_pair_create_host failed")
            self.ctx.pair_h = h
        else:
            self.ctx = ctx
            ctx.chk(ctx.L.ucg_pair_create(ctx.h, STYLE_IDS[style], C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and (isinstance(self.ctx, _HostOnly) or getattr(self.ctx, "h", None)):
            self.ctx.L.ucg_pair_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def settings(self, args):
        self.ctx.chk(self.ctx.L.ucg_pair_settings(self.h, len(args), _argv(args)))

    def coeff(self, args, ntypes=2):
        self.ctx.chk(self.ctx.L.ucg_pair_coeff(self.h, ntypes, len(args), _argv(args)))

    def init(self, ntypes=2, T=1.0):
        self.ctx.chk(self.ctx.L.ucg_pair_init(self.h, ntypes, T))

    def init_one(self, i, j):
        return self.ctx.L.ucg_pair_cut(self.h, i, j)

    @property
    def gather_slots(self):
        return self.ctx.L.ucg_pair_gather_slots(self.h)

    @property
    def sum_fixed(self):
        """True: this pair sums a bead's terms as order-free integer images (include/ucg_hip.h, ucg_pair_sum_fixed)"""
        return bool(self.ctx.L.ucg_pair_sum_fixed(self.h))

    @property
    def cutforce(self):
        return self.ctx.L.ucg_pair_cutforce(self.h)

    def single(self, itype, jtype, rsq, factor_lj=1.0):
        f, e = C.c_double(0), C.c_double(0)
        self.ctx.chk(self.ctx.L.ucg_pair_single(self.h, itype, jtype, rsq, factor_lj, C.byref(f), C.byref(e)))
        return e.value, f.value

    def table_count(self):
        return self.ctx.L.ucg_pair_table_count(self.h)

    def table_params(self, m):
        out = np.zeros(5)
        self.ctx.chk(self.ctx.L.ucg_pair_table_params(self.h, m, _dp(out)))
        return dict(innersq=out[0], delta=out[1], invdelta=out[2], deltasq6=out[3], cut=out[4])

    def table_array(self, m, which):
        n = self.ctx.L.ucg_pair_table_array(self.h, m, which.encode(), None, 0)
        if n <= 0:
            return None
        out = np.zeros(n)
        self.ctx.L.ucg_pair_table_array(self.h, m, which.encode(), _dp(out), n)
        return out

    def tabindex(self):
        n = self.ctx.L.ucg_pair_tabindex(self.h, None, 0)
        out = np.zeros(n, np.int32)
        self.ctx.L.ucg_pair_tabindex(self.h, _ip(out), n)
        return out

    def compute(self, eflag=0, vflag=0):
        e = C.c_double(0)
        v = np.zeros(6)
        self.ctx.chk(self.ctx.L.ucg_pair_compute(self.h, eflag, vflag, C.byref(e), _dp(v)))
        return e.value, v

    def compute_part(self, part):
        """1: the workgroups without ghost neighbours (before the halo arrives), 2: the rest"""
        self.ctx.chk(self.ctx.L.ucg_pair_compute_part(self.h, part))

    def check_errors(self):
        self.ctx.chk(self.ctx.L.ucg_pair_check_errors(self.h))

    # table_ucg_bethe_density on a decomposed run: one pass at a time (see multi.RankSim)
    def density_phase(self, phase, eflag=0, vflag=0):
        e = C.c_double(0)
        v = np.zeros(6)
        self.ctx.chk(self.ctx.L.ucg_pair_density_phase(self.h, phase, eflag, vflag, C.byref(e), _dp(v)))
        return e.value, v

    def density_buffer(self, which):
        """device pointer of the priors (0) / CV forces (1): double2 per owned + ghost bead"""
        return self.ctx.L.ucg_pair_density_buffer(self.h, which)
