"""ctypes binding of libfenris_hip.so (the C ABI in include/fenris_hip.h).

The product path has no CPU fallback: if the HIP library is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FENRIS_HIP_LIB: another build of the library (side-by-side timing of two builds on one box); default: the in-tree build
LIB_PATH = os.environ.get("FENRIS_HIP_LIB") or os.path.join(_HERE, "lib", "libfenris_hip.so")

FH_OK, FH_SINGULAR_JACOBIAN, FH_BAD_ARGUMENT, FH_HIP_ERROR, FH_OUT_OF_MEMORY, FH_INVALID_STATE, FH_UNSUPPORTED = 0, 1, 2, 3, 4, 5, 6
QUAD4, HEX8, TET4, HEX27, TRI3, TET10, QUAD9, TRI6, HEX20, TET20 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
LAPLACE, LINEAR_ELASTIC, NEO_HOOKEAN, STVK, MASS_SCALAR, MASS_VECTOR, TENSOR = 0, 1, 2, 3, 4, 5, 6
SCATTER_ATOMIC, SCATTER_COLORED, SCATTER_GATHER = 0, 1, 2
ASSEMBLE_OVERWRITE = 0x100
ASSEMBLE_REPRODUCIBLE = 0x200

ELEM_NODES = {QUAD4: 4, HEX8: 8, TET4: 4, HEX27: 27, TRI3: 3, TET10: 10, QUAD9: 9, TRI6: 6, HEX20: 20, TET20: 20}
ELEM_DIM = {QUAD4: 2, HEX8: 3, TET4: 3, HEX27: 3, TRI3: 2, TET10: 3, QUAD9: 2, TRI6: 2, HEX20: 3, TET20: 3}

u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)
u32p = C.POINTER(C.c_uint32)


class FenrisError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"fenris_hip error {code}: {message}")
        self.code = code
        self.message = message


class SingularJacobianError(FenrisError):
    """eyre!("Singular element Jacobian encountered") -- src/assembly/local/elliptic.rs:401-404"""

    def __init__(self, message, element):
        super().__init__(FH_SINGULAR_JACOBIAN, message)
        self.element = element


_SIGS = {
    "fh_abi_version": (C.c_int, []),
    "fh_create": (C.c_void_p, [C.c_int]),
    "fh_destroy": (None, [C.c_void_p]),
    "fh_last_error": (C.c_char_p, [C.c_void_p]),
    "fh_last_kernel_name": (C.c_char_p, [C.c_void_p]),
    "fh_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fh_synchronize": (C.c_int, [C.c_void_p]),
    "fh_set_mesh": (C.c_int, [C.c_void_p, C.c_int, f64p, C.c_uint64, u64p, C.c_uint64]),
    "fh_set_mesh_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]),
    "fh_update_vertices": (C.c_int, [C.c_void_p, f64p]),
    "fh_set_connectivity_ragged": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, u64p, u64p, C.c_uint64]),
    "fh_set_active_elements": (C.c_int, [C.c_void_p, C.c_char_p]),
    "fh_set_quadrature_compact": (C.c_int, [C.c_void_p, f64p, f64p, C.c_uint32, C.c_uint64, f64p, u64p]),
    "fh_set_row_range": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "fh_set_operator": (C.c_int, [C.c_void_p, C.c_int]),
    "fh_set_operator_tensor": (C.c_int, [C.c_void_p, f64p, C.c_uint32, C.c_int]),
    "fh_host_pool_trim": (C.c_uint64, []),
    "fh_add_mapped_matrix_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fh_add_mapped_vector_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_uint64, C.c_void_p]),
    "fh_add_mapped_vector_sdim_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_uint64, C.c_void_p]),
    "fh_group_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "fh_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "fh_time_assembly_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "fh_tune_placement_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "fh_vmm_alloc": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "fh_vmm_free": (C.c_int, [C.c_void_p]),
    "fh_group_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "fh_group_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "fh_group_destroy": (None, [C.c_void_p]),
    "fh_group_set_exchange": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64]),
    "fh_group_exchange_start": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fh_group_exchange_finish": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fh_set_quadrature_uniform_data": (C.c_int, [C.c_void_p, f64p, f64p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]),
    "fh_set_quadrature_rules": (C.c_int, [C.c_void_p, C.c_uint64, u64p, f64p, f64p, f64p, u64p]),
    "fh_quadrature_rule_groups": (C.c_int, [C.c_void_p, u64p]),
    "fh_set_affine_tolerance": (C.c_int, [C.c_void_p, C.c_double]),
    "fh_affine_stats": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "fh_set_quadrature_uniform": (C.c_int, [C.c_void_p, f64p, f64p, C.c_uint32, f64p]),
    "fh_set_u": (C.c_int, [C.c_void_p, f64p]),
    "fh_set_u_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fh_solution_dim": (C.c_uint64, [C.c_void_p]),
    "fh_num_elements": (C.c_uint64, [C.c_void_p]),
    "fh_num_nodes": (C.c_uint64, [C.c_void_p]),
    "fh_num_rows": (C.c_uint64, [C.c_void_p]),
    "fh_nnz": (C.c_uint64, [C.c_void_p]),
    "fh_pattern": (C.c_int, [C.c_void_p, u64p, u64p]),
    "fh_pattern_cols": (C.c_int, [C.c_void_p, u64p]),
    "fh_pattern_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fh_color": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "fh_color_parallel": (C.c_int, [C.c_void_p, u64p, u64p, u64p]),
    "fh_set_colors": (C.c_int, [C.c_void_p, C.c_uint64, u64p, u64p]),
    "fh_assemble_matrix": (C.c_int, [C.c_void_p, f64p, C.c_int, u64p]),
    "fh_assemble_matrix_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, u64p]),
    "fh_assemble_matrix_async_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "fh_assemble_vector_async_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fh_poll_status": (C.c_int, [C.c_void_p, u64p]),
    "fh_assemble_matrix_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, u64p]),
    "fh_assemble_matrix_rows_async_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64]),
    "fh_assemble_vector": (C.c_int, [C.c_void_p, f64p, u64p]),
    "fh_assemble_vector_dev": (C.c_int, [C.c_void_p, C.c_void_p, u64p]),
    "fh_assemble_scalar": (C.c_int, [C.c_void_p, f64p, u64p]),
    "fh_assemble_source_vector": (C.c_int, [C.c_void_p, C.c_uint32, f64p, f64p, f64p]),
    "fh_assemble_source_vector_dev": (C.c_int, [C.c_void_p, C.c_uint32, f64p, C.c_void_p, C.c_void_p]),
    "fh_physical_quadrature_points": (C.c_int, [C.c_void_p, f64p]),
    "fh_physical_quadrature_points_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "fh_spmv_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fh_cg_solve": (C.c_int, [C.c_void_p, f64p, f64p, f64p, C.c_int, C.c_double, C.c_uint64, u64p]),
    "fh_cg_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_uint64, u64p]),
    "fh_estimate_L2_error_squared": (C.c_int, [C.c_void_p, C.c_uint32, f64p, f64p, f64p]),
    "fh_estimate_L2_error_squared_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, f64p]),
    "fh_estimate_H1_seminorm_error_squared": (C.c_int, [C.c_void_p, C.c_uint32, f64p, f64p, f64p]),
    "fh_estimate_H1_seminorm_error_squared_dev": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, f64p]),
    "fh_assemble_element_matrices_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "fh_assemble_element_matrices": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, f64p]),
    "fh_apply_dirichlet_csr_dev": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_uint64]),
    "fh_apply_dirichlet_rhs_dev": (C.c_int, [C.c_void_p, C.c_void_p, u64p, C.c_uint64]),
    "fh_gauss": (C.c_int, [C.c_uint32, f64p, f64p]),
    "fh_quadrilateral_gauss": (C.c_int, [C.c_uint32, f64p, f64p]),
    "fh_hexahedron_gauss": (C.c_int, [C.c_uint32, f64p, f64p]),
    "fh_tetrahedron_rule": (C.c_int, [C.c_uint32, f64p, f64p, u32p]),
    "fh_triangle_rule": (C.c_int, [C.c_uint32, f64p, f64p, u32p]),
    "fh_quad_mesh_2d": (C.c_int, [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, f64p, f64p, u64p, u64p, u64p]),
    "fh_hex_mesh": (C.c_int, [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, f64p, u64p, u64p, u64p]),
    "fh_tet_mesh": (C.c_int, [C.c_double, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, f64p, u64p, u64p, u64p]),
    "fh_hex8_to_hex27": (C.c_int, [f64p, C.c_uint64, u64p, C.c_uint64, f64p, u64p, u64p]),
    "fh_load_msh": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, f64p, u64p, u64p, u64p]),
    "fh_msh_last_error": (C.c_char_p, []),
    "fh_tet4_to_tet20": (C.c_int, [f64p, C.c_uint64, u64p, C.c_uint64, f64p, u64p, u64p]),
    "fh_refine_to_quadratic": (C.c_int, [C.c_int, f64p, C.c_uint64, u64p, C.c_uint64, f64p, u64p, u64p]),
    "fh_cuthill_mckee": (C.c_int, [C.c_uint64, u64p, u64p, u64p]),
    "fh_reorder_mesh": (C.c_int, [C.c_uint64, C.c_uint64, u64p, C.c_uint64, u64p, u64p]),
    "fh_lame_from_young_poisson": (C.c_int, [C.c_double, C.c_double, f64p, f64p]),
    "fh_morton_partition": (C.c_int, [C.c_uint32, f64p, C.c_uint64, C.c_uint64, u64p, C.c_uint64, C.c_uint32, C.POINTER(C.c_int32)]),
    "fh_partition_create": (C.c_void_p, [C.c_uint64, C.c_uint64, u64p, C.c_uint64, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]),
    "fh_partition_destroy": (None, [C.c_void_p]),
    "fh_partition_sizes": (C.c_int, [C.c_void_p, u64p]),
    "fh_partition_mesh": (C.c_int, [C.c_void_p, u64p, u64p, u64p, C.POINTER(C.c_uint8), u64p]),
    "fh_partition_exchange": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), u64p, u64p, C.POINTER(C.c_int32), u64p, u64p]),
    "fh_group_exchange_vector_start": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "fh_group_exchange_vector_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    "fh_group_set_exchange_nodes": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), u64p, u64p, C.c_int, C.POINTER(C.c_int32), u64p, u64p]),
}

_lib = None


def lib():
    """Load libfenris_hip.so; fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). fenris_amd has no CPU fallback.")
        # torch ships its own HIP runtime; it must be the one already loaded when libfenris_hip.so is
        # bound, so that torch tensors, streams and this library share a single runtime instance.
        import torch  # noqa: F401

        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if os.environ.get("FENRIS_HIP_LIB") and not hasattr(_lib, name):
                continue              # (another build of the library, for side-by-side timing: it may predate a symbol)
            fn = getattr(_lib, name)  # AttributeError if the ABI symbol is missing
            fn.restype = res
            fn.argtypes = args
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def fp(a):
    return a.ctypes.data_as(f64p) if a is not None else None


def up(a):
    return a.ctypes.data_as(u64p) if a is not None else None


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)
