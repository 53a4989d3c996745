"""fenris_amd -- MI355X-native FEM assembly engine behind fenris's element-assembler interface.

Python host layer over the C ABI of libfenris_hip.so (include/fenris_hip.h).  Only the hot path of
fenris -- global stiffness / residual assembly -- lives here; see DESIGN.md.
"""
from . import _ffi, assembly, io, mesh, operators, quadrature, reorder
from ._ffi import (ASSEMBLE_OVERWRITE, ASSEMBLE_REPRODUCIBLE, HEX8, HEX27, LAPLACE, LINEAR_ELASTIC, NEO_HOOKEAN, QUAD4, SCATTER_ATOMIC,
                   SCATTER_COLORED, SCATTER_GATHER, STVK, TET4, TRI3, TET10, QUAD9, TRI6, HEX20, TET20, MASS_SCALAR, MASS_VECTOR, FenrisError, SingularJacobianError)
from .assembly import (CsrAssembler, CsrMatrix, CsrParAssembler, DisjointSubsetsColors, ElementEllipticAssembler, ElementMassAssembler,
                       ElementEllipticAssemblerBuilder, ElementSourceAssembler, ElementSourceAssemblerBuilder, Engine,
                       MockElementAssembler, UniformQuadratureTable, CompactQuadratureTable, GeneralQuadratureTable,
                       compact_quadrature_table,
                       VectorAssembler, VectorParAssembler, apply_homogeneous_dirichlet_bc_csr,
                       apply_homogeneous_dirichlet_bc_rhs, assemble_scalar, color_nodes, CgSolveError, ConjugateGradient,
                       IdentityOperator, JacobiPreconditioner, RelativeResidualCriterion, estimate_H1_seminorm_error,
                       estimate_H1_seminorm_error_squared, estimate_L2_error, estimate_L2_error_squared)
from .compose import (AggregateElementAssembler, MapElementNodes, TransformElementMatrix, TransformElementScalar,
                      TransformElementVector)
from .mesh import Mesh, hex20_mesh_from_hex8, hex27_mesh_from_hex8, procedural, quad9_mesh_from_quad4, tet10_mesh_from_tet4, tet20_mesh_from_tet4, tri6_mesh_from_tri3
from .operators import (Density, GravitySource, SourceFunction, LameParameters, LaplaceOperator, LinearElasticMaterial, MaterialEllipticOperator,
                        NeoHookeanMaterial, StVKMaterial, TensorEllipticOperator, YoungPoisson)

__all__ = [n for n in dir() if not n.startswith("_")]
