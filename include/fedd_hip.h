/* fedd_hip.h -- C ABI of the MI355X (gfx950) FE-assembly + Schwarz/GMRES hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch / Trilinos types.
 * Every entry point names the reference interface (path:line under /root/reference) whose work
 * it replaces.  Conventions:
 *   - all functions return 0 on success, non-zero on failure; fedd_last_error() gives the message
 *     (the reference throws std::logic_error / std::runtime_error via TEUCHOS_TEST_FOR_EXCEPTION);
 *   - the caller owns every host buffer; the library owns device memory behind fedd_ctx;
 *   - one fedd_ctx per GPU / rank, not thread-safe per handle;
 *   - scalar = double, local ordinal = int32_t, global ordinal = int64_t
 *     (feddlib/core/General/DefaultTypeDefs.hpp:6-15).
 *   - vector fields are node-wise interleaved: dof = dofs_per_node*node + d
 *     (feddlib/core/LinearAlgebra/Map_def.hpp:101-104).
 */
#ifndef FEDD_HIP_H
#define FEDD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fedd_ctx fedd_ctx;

/* ---- forms understood by fedd_assemble (feddlib/core/FE/FE_decl.hpp:130-257) ---- */
enum fedd_form {
    FEDD_FORM_LAPLACE     = 0, /* FE::assemblyLaplace          FE_def.hpp:604-667   (dofs 1)          */
    FEDD_FORM_LAPLACE_VEC = 1, /* FE::assemblyLaplaceVecField  FE_def.hpp:670-734   (dofs dim, diag)  */
    FEDD_FORM_MASS        = 2, /* FE::assemblyMass "Scalar"    FE_def.hpp:454-524   (dofs 1)          */
    FEDD_FORM_MASS_VEC    = 3, /* FE::assemblyMass "Vector"    FE_def.hpp:454-524   (dofs dim, diag)  */
    FEDD_FORM_LINELAS     = 4, /* FE::assemblyLinElasXDim      FE_def.hpp:2739-3040 (dofs dim, full);
                                  params = {lambda, mu}                                              */
    FEDD_FORM_BDSTAB      = 5  /* FE::assemblyBDStabilization  FE_def.hpp:2151-2220 (dofs 1, P1 only): the Bochev-Dohrmann
                                  pressure block of P1/P1 Stokes, C_ij = |det B| (sum_q w_q phi_i phi_j - |ref| scale),
                                  scaled by -1/viscosity by the caller (Stokes_def.hpp:98-105)        */
};

/* ---- how the dofs of a node couple in the CSR pattern ---- */
enum fedd_block_mode {
    FEDD_BLOCK_SCALAR = 0,     /* dofs_per_node == 1                                                  */
    FEDD_BLOCK_DIAG   = 1,     /* (dim*i+d, dim*j+d) only        -- what assemblyLaplaceVecField inserts */
    FEDD_BLOCK_FULL   = 2      /* every (a,b) pair                -- what assemblyLinElasXDim inserts    */
};

enum fedd_combine {            /* FROSch "Combine Values in Overlap" (laplace/parametersPrec.xml:31)    */
    FEDD_COMBINE_RESTRICTED = 0,
    FEDD_COMBINE_AVERAGING  = 1,
    FEDD_COMBINE_FULL       = 2
};

/* kernel classes whose device time the library accumulates with HIP events when timing is on */
enum fedd_timer {
    FEDD_T_SYMBOLIC = 0,  /* adjacency + CSR pattern                           */
    FEDD_T_ASSEMBLE = 1,  /* matrix gather-assembly kernel                     */
    FEDD_T_RHS      = 2,
    FEDD_T_DIRICHLET= 3,
    FEDD_T_SPMV     = 4,
    FEDD_T_SCHWARZ_SETUP = 5,
    FEDD_T_SCHWARZ_APPLY = 6,
    FEDD_T_ORTHO    = 7,  /* GMRES multi-dot / multi-axpy kernels              */
    FEDD_T_COARSE_SETUP = 8,  /* second level: Galerkin product + dense inverse */
    FEDD_T_COARSE_APPLY = 9,  /* second level: restrict, K0^-1, prolongate      */
    FEDD_T_HALO     = 10, /* ghost import: pack, send / receive, unpack (several ranks)  */
    FEDD_T_ALLREDUCE= 11, /* all-reduce calls (inside the classes that issue them)        */
    FEDD_T_SPMV_SETUP = 12, /* compaction of the solver's SpMV stream (once per assembled matrix) */
    FEDD_T_GS_DOT   = 13, /* Gram-Schmidt sweep 1 alone: the multi-dot kernel over the Krylov basis (inside ORTHO)   */
    FEDD_T_GS_UPDATE= 14, /* Gram-Schmidt sweep 2 alone: the multi-axpy kernel over the Krylov basis (inside ORTHO)  */
    FEDD_T_GS_FUSED = 15, /* s-step solver: first update and second dot of a block in one sweep, k_blockfuse (inside ORTHO) */
    FEDD_T_COUNT    = 16
};

/* ------------------------------------------------------------------------------------------------
 * context  (replaces the per-rank Teuchos::Comm + Tpetra node the reference gets from
 * Xpetra::DefaultPlatform, feddlib/problems/tests/laplace/main.cpp:60-62)
 * nccl_unique_id: NULL for a single rank; else the 128-byte ncclUniqueId shared by all ranks.
 * ---------------------------------------------------------------------------------------------- */
/* (the environment variable FEDD_OPTIONS=key=value,... is applied to every new context through fedd_set_option) */
int  fedd_ctx_create(fedd_ctx** out, int device, const void* nccl_unique_id, int rank, int nranks);
void fedd_ctx_destroy(fedd_ctx* ctx);
const char* fedd_last_error(void);
int  fedd_sync(fedd_ctx* ctx);                       /* hipStreamSynchronize on the context's stream */
int  fedd_nccl_unique_id(void* id128);               /* rank 0 calls this, then shares the 128 bytes */

/* ------------------------------------------------------------------------------------------------
 * structured mesh generator, host side
 * replaces MeshStructured::buildMesh2D/3D P1 branch + setStructuredMeshFlags + Map::buildUniqueMap
 * (feddlib/core/Mesh/MeshStructured_def.hpp:348-463, 703-806, 2974-3203;
 *  feddlib/core/LinearAlgebra/Map_def.hpp:184-210) as sequenced by Domain::buildMesh
 * (feddlib/core/FE/Domain_def.hpp:201-265).
 * decomp[3] = blocks per direction (the reference supports only N x N x N; {N,N,N} reproduces it
 * exactly, other shapes are the 1x1x2 / 1x2x2 splits of the same global grid used for the 2- and
 * 4-GPU scaling points).  cells[3] = cells per block and direction (the reference's M).
 * with_ghost_elements = 1 appends the neighbour blocks' elements that touch an owned node (and
 * their nodes), so that every owned row can be assembled without a matrix exchange.
 * with_ghost_elements = L >= 2 (at most 8) appends L layers of elements around the owned nodes on every
 * side with a neighbour, so that the rows of the ghost nodes within L - 1 layers are complete on this rank
 * too (row ghosts, see fedd_mesh_set_rows; what FROSch obtains by importing the overlapping matrix rows
 * from their owners).  fedd_mesh_structured_row_ghosts lists them (count with NULL arrays first).
 * L = 2 gives the Schwarz subdomains at a rank boundary true overlap rows; L = 4 (27-node boxes, overlap 1)
 * lets every rank build whole every box that holds one of its nodes, see fedd_schwarz_setup.
 * ---------------------------------------------------------------------------------------------- */
int fedd_mesh_structured_sizes(int dim, const int* decomp, const int* cells, int rank,
                               int with_ghost_elements,
                               int64_t* n_elem, int64_t* n_rep, int64_t* n_uni, int64_t* n_global);
int fedd_mesh_structured_build(int dim, const int* decomp, const int* cells, int rank,
                               const double* origin, const double* size, int flags_option,
                               int with_ghost_elements,
                               int32_t* conn /*[n_elem*(dim+1)] local repeated ids*/,
                               double* xyz /*[n_rep*dim]*/, int64_t* gid_rep /*[n_rep]*/,
                               int32_t* flag_rep /*[n_rep]*/,
                               int64_t* gid_uni /*[n_uni]*/, int32_t* flag_uni /*[n_uni]*/);
int fedd_mesh_structured_row_ghosts(int dim, const int* decomp, const int* cells, int rank, int with_ghost_elements,
                                    const double* origin, const double* size, int flags_option,
                                    int64_t* n_row_ghosts, int64_t* gid /*nullable*/, int32_t* flag /*nullable*/);

/* ------------------------------------------------------------------------------------------------
 * unstructured input, host side, one rank: INRIA/medit ".mesh" reader (MeshFileReader.cpp:16-106,
 * MeshFileReader.hpp:35-127, 1-based ids converted as MeshUnstructured_def.hpp:1181-1190) and the
 * P2-from-P1 construction (MeshUnstructured::buildP2ofP1MeshEdge, MeshUnstructured_def.hpp:129-410;
 * edge ids = rank in the sorted (min,max) list, EdgeElements.cpp:105-155; mid node id = n_vert + edge id;
 * slots (0,1)->4 (1,2)->5 (0,2)->6 (0,3)->7 (1,3)->8 (2,3)->9, :755-772; flags :806-900).
 * surf = boundary entities (Edges in 2D, Triangles in 3D), dim nodes each.
 * ---------------------------------------------------------------------------------------------- */
int fedd_mesh_read_sizes(const char* path, int dim, int64_t* n_vert, int64_t* n_elem, int64_t* n_surf);
int fedd_mesh_read(const char* path, int dim, double* xyz, int32_t* vflag, int32_t* conn, int32_t* eflag,
                   int32_t* surf, int32_t* sflag);
int fedd_mesh_p2_sizes(int dim, int64_t n_elem, const int32_t* conn_p1, int64_t* n_edges);
int fedd_mesh_p2_build(int dim, int64_t n_vert, int64_t n_elem, const int32_t* conn_p1, const double* xyz_p1,
                       const int32_t* vflag_p1, int64_t n_surf, const int32_t* surf, const int32_t* sflag,
                       int volume_id, int32_t* conn_p2, double* xyz_p2, int32_t* flag_p2);

/* ------------------------------------------------------------------------------------------------
 * element partitioner for unstructured meshes, host side: replaces MeshPartitioner::readAndPartitionMesh and the maps
 * it builds (feddlib/core/Mesh/MeshPartitioner_def.hpp:224-530; METIS_PartMeshDual :324, repeated map :358-397,
 * Map::buildUniqueMap Map_def.hpp:184-210) with a deterministic balanced recursive coordinate bisection of the element
 * centroids (every element on exactly one part).  conn holds GLOBAL node ids (0-based), nen nodes per element, the
 * first dim + 1 of them the vertices.
 *   fedd_mesh_partition          elem_part[n_elem] in [0, nparts)
 *   fedd_mesh_partition_sizes    sizes of rank's mesh: its own elements + `ghost_layers` layers of elements around
 *                                the owned nodes (1: owned rows complete; L >= 2: also the rows of the ghost nodes within
 *                                L - 1 layers, the row ghosts of fedd_mesh_set_rows)
 *   fedd_mesh_partition_extract  the rank's mesh in the form fedd_mesh_set / fedd_mesh_set_rows / fedd_halo_set_owners
 *                                take: local connectivity, coordinates, repeated map (ascending global ids) with flags
 *                                and owner ranks (lowest rank among the parts whose own elements hold the node),
 *                                unique map, row ghosts; elem_gid = global ids of the local elements.  Any output may
 *                                be NULL.
 * ---------------------------------------------------------------------------------------------- */
int fedd_mesh_partition(int dim, int nen, int64_t n_elem, const int32_t* conn, int64_t n_node, const double* xyz,
                        int nparts, int32_t* elem_part);
int fedd_mesh_partition_sizes(int nen, int64_t n_elem, const int32_t* conn, int64_t n_node, const int32_t* elem_part,
                              int nparts, int rank, int ghost_layers, int64_t* n_elem_loc, int64_t* n_rep, int64_t* n_uni,
                              int64_t* n_row_ghosts);
int fedd_mesh_partition_extract(int dim, int nen, int64_t n_elem, const int32_t* conn, int64_t n_node, const double* xyz,
                                const int32_t* flag, const int32_t* elem_part, int nparts, int rank, int ghost_layers,
                                int32_t* conn_loc, double* xyz_loc, int64_t* gid_rep, int32_t* flag_rep, int32_t* owner_rep,
                                int64_t* gid_uni, int32_t* flag_uni, int64_t* row_ghost_gid, int32_t* row_ghost_flag,
                                int64_t* elem_gid);

/* ------------------------------------------------------------------------------------------------
 * reference-element tables the assembly kernels stage in LDS (host side, no GPU needed): quadrature points and
 * weights of FE::getQuadratureValues (feddlib/core/FE/FE_def.hpp:6023-6727, degree remapping included) and the
 * values / gradients of FE::phi / FE::gradPhi (:4947-5087, :5565-5713) at those points.  Read-back for parity
 * tests against the reference's literals (tests/golden/ref_tables.json).
 * fedd_fe_quadrature: pts[nq*dim], w[nq] (call with NULL arrays for nq).  fedd_fe_basis: phi[nq*nen], dphi[nq*nen*dim].
 * ---------------------------------------------------------------------------------------------- */
int fedd_fe_quadrature(int dim, int degree, int* nq, double* pts, double* w);
int fedd_fe_basis(int dim, int nen, int degree, double* phi, double* dphi);

/* ------------------------------------------------------------------------------------------------
 * mesh upload: what FE::assemblyXxx reads through domainVec_[FEloc]->getElementsC(),
 * getPointsRepeated(), getMapRepeated() (feddlib/core/FE/FE_def.hpp:617-621) and what BCBuilder
 * reads through getBCFlagUnique()/getMapUnique() (feddlib/core/General/BCBuilder_def.hpp:625-626).
 * nen = nodes per element (P1: dim+1, P2: 6 / 10).  The first dim+1 nodes are the vertices
 * (FE::buildTransformation uses only those, FE_def.hpp:5342-5357).
 * ---------------------------------------------------------------------------------------------- */
int fedd_mesh_set(fedd_ctx* ctx, int dim, int nen, int64_t n_elem, const int32_t* conn,
                  int64_t n_rep, const double* xyz, const int64_t* gid_rep,
                  int64_t n_uni, const int64_t* gid_uni, const int32_t* bcflag_uni);
/* The same with row ghosts: nodes of the repeated map that another rank owns but ALL of whose elements are
 * in this rank's mesh.  Their matrix rows are then built, assembled and given the Dirichlet treatment
 * like owned rows (they follow the owned rows in the CSR arrays), and the overlapping Schwarz
 * subdomains take the true rows of such nodes instead of identity rows.  The boxes of the subdomains come
 * from one lattice over all ranks, and a box that holds an owned node is built whole -- with the other
 * ranks' nodes inside it and the full overlap -- wherever all its dofs have stored rows (with 27-node boxes
 * and overlap 1: row ghosts 3 node layers deep); each rank keeps the rows of its own nodes.  The
 * preconditioner is then the one a single rank would build, whatever the number of ranks.  Where the row
 * ghosts do not reach, a rank takes its own part of the box (true overlap rows still).  Everything else
 * (SpMV, vectors, fedd_csr_get, the coarse level) stays on the owned rows.  With row ghosts declared, only
 * they are imported in the halo exchange.  Every ghost node adjacent to an owned node must be listed. */
int fedd_mesh_set_rows(fedd_ctx* ctx, int dim, int nen, int64_t n_elem, const int32_t* conn,
                       int64_t n_rep, const double* xyz, const int64_t* gid_rep,
                       int64_t n_uni, const int64_t* gid_uni, const int32_t* bcflag_uni,
                       int64_t n_row_ghosts, const int64_t* row_ghost_gid, const int32_t* row_ghost_bcflag);

/* symbolic CSR on the owned (unique-map) rows: what Tpetra's dynamic insert + fillComplete
 * discover (feddlib/core/LinearAlgebra/Matrix_def.hpp:88-92,192-199). */
int fedd_pattern_build(fedd_ctx* ctx, int dofs_per_node, int block_mode, int64_t* nnz_out);

/* numeric assembly into the pattern; FE::assemblyXxx + fillComplete (file:line per form above). */
int fedd_assemble(fedd_ctx* ctx, int form, const double* params);

/* FE::assemblyRHS + MultiVector::exportFromVector(...,"Add") (FE_def.hpp:4694-4766,
 * feddlib/problems/abstract/Problem_def.hpp:184-216): constant f per dof, quadrature degree
 * determineDegree(dim,FE,Std) + extra_degree. */
int fedd_assemble_rhs(fedd_ctx* ctx, int dofs_per_node, const double* f_const, int extra_degree);

/* BCBuilder::setSystem + setRHS for constant boundary values (BCBuilder_def.hpp:589-707, 93-170):
 * rows of nodes whose flag is in flags[] become unit rows (pattern kept), rhs <- values.
 * comp_mask[n_bc*dofs] (nullable = all components), values[n_bc*dofs]. */
int fedd_dirichlet(fedd_ctx* ctx, int n_bc, const int32_t* flags, const int32_t* comp_mask,
                   const double* values);

/* same, for boundary values that depend on the node (the host evaluates the user's BC function at
 * every flagged unique node, BCBuilder_def.hpp:128-143, and passes the results): owned_nodes[n]
 * local unique-map node ids, comp_mask[n*dofs] (nullable), values[n*dofs]. */
int fedd_dirichlet_nodes(fedd_ctx* ctx, int64_t n, const int32_t* owned_nodes, const int32_t* comp_mask,
                         const double* values);

/* unit rows on arbitrary system rows (merged block systems): what setLocalRowOne on the diagonal
 * block + setLocalRowZero on the off-diagonal blocks (BCBuilder_def.hpp:653-707) leave in that row. */
int fedd_dirichlet_rows(fedd_ctx* ctx, int64_t n, const int32_t* rows, const double* values);

/* ------------------------------------------------------------------------------------------------
 * mixed / block problems (one rank for now).  Blocks live in numbered slots beside the system matrix.
 *   fedd_matrix_store     copy the system matrix into `slot` (0..3)
 *   fedd_matrix_scale     Matrix::scale (Stokes_def.hpp:83-85,102); slot < 0 = system matrix
 *   fedd_assemble_div     FE::assemblyDivAndDivT (FE_def.hpp:1932-2057): velocity = the mesh's element,
 *                         pressure = P1 on the vertices = the first n_pressure_nodes node ids; B -> slot_b
 *                         (n_p x dim*n_v), B^T -> slot_bt.  Overwrites the system slot (scratch).
 *   fedd_block_merge      BlockMatrix::merge + BlockMap::merge (BlockMatrix_def.hpp:119-148,212-287;
 *                         BlockMap_def.hpp:55-80): system <- [A B^T; B C] (slot < 0 = empty block); merged
 *                         global ids = block-local gid + cumulated (maxAllGlobalIndex + 1).
 *   fedd_matrix_sizes/get read a stored block back (local column ids).
 * ---------------------------------------------------------------------------------------------- */
int fedd_matrix_store(fedd_ctx* ctx, int slot);
int fedd_matrix_scale(fedd_ctx* ctx, int slot, double alpha);
int fedd_assemble_div(fedd_ctx* ctx, int64_t n_pressure_nodes, int slot_b, int slot_bt);
int fedd_block_merge(fedd_ctx* ctx, int slot_a, int slot_bt, int slot_b, int slot_c);
int fedd_matrix_sizes(fedd_ctx* ctx, int slot, int64_t* n_rows, int64_t* n_cols, int64_t* nnz);
int fedd_matrix_get(fedd_ctx* ctx, int slot, int64_t* rowptr, int32_t* colind, double* val);

/* read-back for Tpetra::CrsMatrix fill / parity (Matrix::getLocalRowView analog). col_gid maps
 * a local column index to its global dof id. */
int fedd_csr_sizes(fedd_ctx* ctx, int64_t* n_rows, int64_t* n_cols, int64_t* nnz);
int fedd_csr_get(fedd_ctx* ctx, int64_t* rowptr, int32_t* colind, double* val, int64_t* col_gid);
int fedd_rhs_get(fedd_ctx* ctx, double* rhs_owned);
int fedd_rhs_set(fedd_ctx* ctx, const double* rhs_owned);
int fedd_solution_get(fedd_ctx* ctx, double* x_owned);

/* y = A x on owned rows incl. ghost import: Matrix::apply (Matrix_def.hpp:245-254).
 * host pointers; fedd_spmv_device runs `reps` launches on the resident vectors (bench). */
int fedd_spmv(fedd_ctx* ctx, const double* x_owned, double* y_owned);
int fedd_spmv_device(fedd_ctx* ctx, int reps);
/* what the SpMV streams: nnz of the parity CSR (owned rows), nnz of the compacted stream actually read
 * (= nnz_pattern with "spmv_compact" 0 or when the windowed kernel is not in use) */
int fedd_spmv_info(fedd_ctx* ctx, int64_t* nnz_pattern, int64_t* nnz_streamed);

/* one-level overlapping additive Schwarz (replaces Thyra::initializePrec on the FROSch factory,
 * feddlib/problems/Solver/Preconditioner_def.hpp:243-463; options from
 * feddlib/problems/tests/laplace/parametersPrec.xml:10-61).  Subdomains are `target_nodes`-node
 * boxes of the rank's owned nodes (batched, many per GPU), extended by `overlap` graph
 * layers; exact dense local solves.
 * two_level != 0 adds a coarse level, M^-1 = M_one^-1 + Phi K0^-1 Phi^T, the role FROSch's
 * GDSWCoarseOperator plays under "TwoLevel" = true (parametersPrec.xml:62-122).  coarse_kind:
 * FEDD_COARSE_Q1 = multilinear hat functions of a regular lattice over the global bounding box
 * (DESIGN.md "two-level" gives the normative definition and why it stands in for GDSW here);
 * K0 = Phi^T A Phi is formed and inverted on the device, replicated on every rank. */
#define FEDD_COARSE_Q1 1
/* FEDD_COARSE_GDSW: FROSch's GDSWCoarseOperator on a second, coarse decomposition -- the cells of the same regular lattice
 * (fedd_schwarz_set_coarse; default one cell per 1000 nodes, at most what the dense coarse solver takes: 10^3 cells for scalar, 7^3 for 3-dof problems in 3D):
 * interface nodes are classified into the faces / edges / vertices between the cells, the coarse basis is the null space
 * (constants per dof component: what FROSch has without node coordinates, parametersPrec.xml:5 "Use node lists" = false)
 * restricted to each interface component and extended discrete-harmonically into the cell interiors (device GMRES +
 * one-level Schwarz on the constrained operator, option "gdsw_tol", default 1e-4 for GDSW and 1e-3 for RGDSW: the outer iteration count is that of exact extensions down to there, see
 * profiles/r02_gdsw_tol_sweep.txt and r03_gdsw_tol_sweep_stacked.txt; the parity tests ask for 1e-13; option "gdsw_block" 1 (default) = sixteen columns at a
 * time as one stacked system over an SpMM and a matrix-core Schwarz apply, multi.hip, 0 = column by column; "multi_ch" 4 / 8 / 16 = matrix-core
 * steps per flight of gathers of that apply), K0 = Phi^T A Phi inverted on the
 * matrix cores.  (2g - 1)^dim * dofs coarse dofs for g cells per direction.
 * Option "gdsw_rotations" 1 (default 0; GDSW and RGDSW, vector problems with dofs = dim): what FROSch builds with "Use node
 * lists" = true and "Rotations" = true (steadyLinElas/parametersPrec.xml:6, 100) -- every interface component carries the
 * linearised rotations about its centre next to the translations (3 + 3 functions in 3D, 2 + 1 in 2D; the coarse space then
 * holds the rigid-body modes of every cell).  Functions that are linearly dependent on a component's free dofs are dropped
 * (a Cholesky sweep over the component's Gram matrix: a one-node vertex keeps its translations, a straight edge 5 of 6);
 * (2g - 1)^dim * (dofs + rotations) coarse dofs, the dropped ones with a unit diagonal in K0.  Normative definition:
 * oracle/fedd_oracle.py CoarseGDSW(rotations=True). */
#define FEDD_COARSE_GDSW 2
/* FEDD_COARSE_RGDSW: the reduced GDSW space (FROSch RGDSWCoarseOperator, the one steadyLinElas_Perf/parametersPrec.xml:18
 * names; Dohrmann & Widlund 2017, option 1): coarse functions only for the coarse nodes of the same decomposition (its
 * vertices; for slab / pencil decompositions the interface components without lower-dimensional neighbours), an interface
 * node of component e carrying 1 / |C(e)| for each adjacent coarse node; harmonic extensions and K0 as for GDSW.
 * (g - 1)^dim * dofs coarse dofs: the default lattice is one cell per 400 nodes, at most what the dense coarse solver takes
 * (14^3 cells for 3-dof problems in 3D). */
#define FEDD_COARSE_RGDSW 3
int fedd_schwarz_setup(fedd_ctx* ctx, int overlap, int combine, int two_level, int coarse_kind);
/* target_nodes = 0 (the default): 27 nodes for scalar problems, 27 / dofs-per-node for vector ones */
int fedd_schwarz_set_target(fedd_ctx* ctx, int target_nodes, double scale);
/* number of lattice cells the coarse level aims at (0 = default: global nodes / 1000, clamped to
 * [1, 1728]); call before fedd_schwarz_setup */
int fedd_schwarz_set_coarse(fedd_ctx* ctx, double cells_target);
/* coarse level read-back (parity): lattice cells per direction, coarse dofs n0, K0^-1 row-major */
int fedd_schwarz_coarse_sizes(fedd_ctx* ctx, int32_t cells[3], int64_t* n0);
int fedd_schwarz_coarse_get(fedd_ctx* ctx, double* k0_inverse);
int fedd_schwarz_apply(fedd_ctx* ctx, const double* r_owned, double* z_owned);
int fedd_schwarz_apply_device(fedd_ctx* ctx, int reps);
int fedd_schwarz_info(fedd_ctx* ctx, int64_t* n_subdomains, int64_t* max_size, int64_t* inverse_bytes);
/* number of DISTINCT local matrices of the last setup: subdomains whose principal submatrices agree (to 2^-44 of each row's
 * largest entry; option "schwarz_dedupe", default 1) share one stored inverse, which fedd_schwarz_info's inverse_bytes
 * counts once.  On the structured cube of the headline a few hundred of the 389 017 subdomains are distinct. */
int fedd_schwarz_unique(fedd_ctx* ctx, int64_t* n_unique);
/* sum over this rank's subdomains of their sizes (owned + overlap dofs) and of their owned dofs: what one apply gathers
 * and scatters (byte model of the apply kernel in bench.py); either output may be NULL */
int fedd_schwarz_sizes(fedd_ctx* ctx, int64_t* sum_sizes, int64_t* sum_owned);
/* subdomains whose dof list is their representative's list shifted by a constant (every box of a class on a structured mesh):
 * the matrix-core apply computes their dof ids from the representative's offsets and does not read their lists */
int fedd_schwarz_conforming(fedd_ctx* ctx, int64_t* n_conforming);

/* right-preconditioned restarted GMRES (replaces Thyra::solve on the Belos "Block GMRES"
 * LOWS, feddlib/problems/Solver/LinearSolver_def.hpp:72-135; parametersSolver.xml:5-15).
 * b_owned / x_owned may be NULL: then the assembled rhs is used and the solution stays on the
 * device (fedd_solution_get reads it).  use_prec = 0 runs unpreconditioned. Returns the iteration
 * count the way Problem::solve does (its_out). */
int fedd_gmres(fedd_ctx* ctx, const double* b_owned, double* x_owned, double rtol, int max_it,
               int restart, int use_prec, int* its_out, double* relres_out);
/* The same solve from a given initial guess: the reference's "Zero Initial Guess" = false (LinearSolver_def.hpp:76-78, where
 * the solution vector is only cleared when the key is true).  x_owned in: x_0, out: the solution; NULL: x_0 is the vector the
 * device holds (the last solution, or what fedd_schwarz_coarse_apply(ctx, NULL, NULL) left there) and the solution stays on
 * the device.  The relative residual refers to ||r_0|| = ||b - A x_0||, as Belos' default scaling does. */
int fedd_gmres_x0(fedd_ctx* ctx, const double* b_owned, double* x_owned, double rtol, int max_it,
                  int restart, int use_prec, int* its_out, double* relres_out);
/* The structures that depend on the mesh alone are built once per mesh, at the first call that needs them, and reused by every
 * later assembly: the node -> element adjacency (fedd_pattern_build) and the element-major tile structures of the assembly
 * kernel (fedd_assemble, P1 forms).  Their wall time (ms, device synchronised before and after) is not part of a steady-state
 * step; a driver that assembles once (the reference's do: laplace/main.cpp:199-208) pays it once.  tiles_state: 0 = not built
 * (yet), 1 = built, -1 = the mesh does not fit the tile limits (pair kernels are used).  Outputs may be NULL.  Both are built
 * by device kernels (option "asm_tiles_host" 1: the tile structures by the round-3 host builder, for A/B). */
int fedd_mesh_setup_info(fedd_ctx* ctx, double* adjacency_ms, double* tiles_ms, int* tiles_state, int64_t* n_tiles);
/* how the last solve ended: floor_reached = 1 when the s-step solver stopped because b - A x had reached its rounding floor
 * (the recurrence residual kept falling, the true residual did not follow, and a restart from the true residual did not help
 * either; relres_out of that solve is the TRUE residual and may exceed rtol by up to 100x), 2 when three restart cycles in a
 * row made no progress; recurrence_relres = the recurrence residual at that point (-1 otherwise); outputs may be NULL */
int fedd_gmres_status(fedd_ctx* ctx, int* floor_reached, double* recurrence_relres);
/* The second level alone, z = Phi K0^-1 Phi^T r: FROSch's "Only apply coarse", which the reference uses for
 * "Level Combination" = "Multiplicative" (LinearSolver_def.hpp:98-104: one coarse pre-apply of the right-hand side into the
 * solution vector, then the solve).  r_owned / z_owned both NULL: r = the assembled right-hand side, z -> the device's solution
 * vector (then fedd_gmres_x0(ctx, NULL, NULL, ...) continues from it).  Needs fedd_schwarz_setup(two_level = 1). */
int fedd_schwarz_coarse_apply(fedd_ctx* ctx, const double* r_owned, double* z_owned);
/* the orthogonalisation in use ("gmres_kind") and, for the s-step form, its block length and the blocks of the last solve
 * (all, and those that were cut short because the block basis became numerically dependent); outputs may be NULL */
int fedd_gmres_info(fedd_ctx* ctx, int* kind, int* s, int* blocks, int* cut_blocks);
/* blocks of the last s-step solve orthogonalised with three sweeps instead of four (option "gmres_fuse": k_blockfuse) */
int fedd_gmres_fused_blocks(fedd_ctx* ctx, int* blocks);

/* tuning knobs (A/B tests), 0 is the default of each: "spmv_kind" 0 = CSR-window (CSR-stream when a row has more than 256 entries), 1 = row-per-lane-group, 2 = CSR-stream;
 * "pat_hash" 1 (default) = node pattern of vertex-only elements merged through a hash table of list positions (symbolic.hip), 0 = ordered insertion;
 * "asm_tiles" 1 (default) = the P1 Laplace / vector-Laplace / elasticity forms are assembled element-major over tiles of ~27 nodes
 * (every element of a tile evaluated once, contributions gathered per CSR slot from lists built once per mesh), 0 = the pair
 * kernels; "asm_kind" 4 = tiles whatever "asm_tiles" says, 0 = pair-parallel assembly (slot-addressed accumulation for the block forms -- elasticity, B / B^T --, slot sweep for
 * the scalar forms), 1 = lane-per-row gather, 2 = slot sweep always, 3 = slot-addressed always; "asm_u" pairs per lane whose
 * loads are in flight together in the slot-addressed kernel (P1; default 1); "asm_dbg" ablation switches (development); "apply_kind" 0 = restricted Schwarz
 * apply by the setup's outcome (batched matrix-core kernel when at most a quarter of at least 4096 subdomains have distinct
 * local matrices -- on the batch table of the setup, k_apply_bt, when every subdomain conforms to its representative, on chunk
 * records, k_apply_mfma, otherwise --, else the flat streaming kernel), 1 = strided, 2 = flat without the compact LDS layout,
 * 4 = matrix-core kernel whenever the inverses are shared (any number of subdomains), 6 = as 4, with the chunk-record kernel
 * k_apply_mfma also where the batch table exists (A/B and tests; same bits), 7 = the warp-specialised form of that kernel (four waves
 * multiply, four gather r three batches ahead; 33 ... 64 owned rows per subdomain; measured slower, kept for A/B); "apply_bt" 1
 * (default) = the setup builds the batch table (0: never); "apply_span" places per workgroup of the matrix-core kernels
 * (multiples of 16; 0 = one round of workgroups with the batch table, 32 / 64 / 96 / 128 by the number of subdomains without);
 * "apply_dbg" > 0 ablation bits of k_apply_mfma<4, 12> (development: wrong results by design), -1 = phase clocks of one wave of
 * k_apply_bt<4, 12> printed by the kernel (tools/apply_phases.py); "md2_gy" column groups in flight per row block of the
 * Gram-Schmidt dot sweep (0 = by vector length), "md2_nch" its 512-row chunks per workgroup (2 or 4; default 4);
 * "gmres_hostwrite" 1 (default) = the solver's small kernel writes the three numbers of the host's lagged convergence test
 * into mapped pinned memory itself, 0 = an asynchronous copy per iteration;
 * "spmv_pattern" 1 (default) = column patterns for matrices beyond the Infinity Cache (fedd_spmv_patterns), 2 = for every
 * matrix, 0 = off; "spmv_pat_nu" / "spmv_win_nu" window sizes of the pattern / per-entry SpMV kernels (0 = default); "inv_kind" 0 = scalar-pivot local inverses that drop finished overlap rows, 1 = rank-4 block sweep on the
 * matrix cores, 2 = scalar-pivot without dropping rows;
 * "gmres_kind" 0 = two-pass Gram-Schmidt with the second pass delayed (DCGS2), 1 = plain two passes, 2 = s-step GMRES: the
 * basis grows "gmres_s" (1 ... 16; 0, the default: 16 from 1.2 million rows per rank, else 8) vectors at a time -- blocks longer than 8 on the Newton block basis ("gmres_newton" 1, the
 * default: shifts = Leja-ordered Ritz values of the first gmres_s Arnoldi steps, which run in monomial blocks of at most 8; 0 =
 * monomial basis, blocks of at most 8) -- and each block is orthogonalised by block Gram-Schmidt with two
 * passes (four sweeps over the basis -- three with "gmres_fuse" -- and two reductions per block instead of two sweeps and one reduction per iteration; same
 * iterates as 0 / 1 in exact arithmetic; a block is cut where the squared sine of a new vector against its predecessors falls
 * to "gmres_chol_tol", default 1e-13; the convergence claim is checked against the true residual and the residual returned is
 * the true one; tolerances below 1e-9 / 1e-11 take blocks of at most 5 / 3 vectors, because the recurrence of a longer block
 * loses touch with the true residual there ("gmres_tol_blocks" 0 lifts that cap: measurements held to an iteration count
 * instead of a tolerance); "gmres_fuse" 1 = blocks of 16: the first pass' update and the second pass' dot products run as one
 * sweep (k_blockfuse: three sweeps per block; the second read of a workgroup's rows comes from the Infinity Cache), 0 = four
 * sweeps, -1 (default) = one sweep from 4 million rows per rank on; unless 0 the last update of a block that fills a restart cycle
 * also forms the solution update sum_c y_c V_c (one read of the basis less per cycle); "gmres_spec" n > 0: n operator applications of the next block are put into the
 * stream before the host reads a block's outcome, so that the GPU works through that round trip -- default 0: on one GPU the
 * round trip is not what the step waits for (tools/share_n8.py), an A/B switch for multi-GPU runs), see fedd_gmres_info;
 * "box_kind" 0 = Schwarz boxes from one lattice over the nodes of all ranks, 1 = a lattice per rank;
 * "asm_lds_kb" (default 37) = LDS budget in KB of the assembly kernel's contribution park, i.e. rows per workgroup;
 * "spmv_nt" 1 = the window SpMV streams the matrix non-temporally (x then survives in L2 between node planes:
 * -6 % back to back on a 1.8 GB matrix, 0 to -4 % inside the solver, slower on matrices that fit the Infinity
 * Cache), 0 = never, -1 (default) = for matrices larger than the Infinity Cache;
 * "spmv_compact" 1 (default) = SpMV streams a solver-private copy of the owned rows without their numerically zero
 * entries (the structural zeros kept for pattern parity, the zeroed entries of Dirichlet rows); fedd_csr_get always
 * returns the reference pattern and values; 0 = stream the parity CSR itself;
 * "spmv_drop_tol" (default 2^-52) = an entry is left out of that copy when |a_ij| <= tol * max_k |a_ik|: 0 drops the
 * entries that are exactly 0.0 only (y then identical bit for bit, for finite x, to the product with the parity CSR), the
 * default also drops cancellation noise below one ulp of the row's largest entry (cf. the reference's optional setZeros_
 * threshold, FE_def.hpp:719-721), which changes y by less than the rounding error of the row sum;
 * "spmv_exact_public" 1 (default) = fedd_spmv itself (Matrix::apply for the caller, residual checks) multiplies with the parity
 * CSR, every stored entry, so its y is the reference's product whatever the solver streams; 0 = fedd_spmv runs the solver's
 * compacted stream (tests and measurements of those kernels);
 * "schwarz_dedupe" 1 (default) = subdomains with the same local matrix share one inverse (see fedd_schwarz_unique), 0 = every
 * subdomain is inverted and stored on its own;
 * "schwarz_fp_kind" 0 (default) = the fingerprints of that sharing are built from one hash per matrix ROW (column offsets and
 * quantised values of all its entries) and the rows' positions in the subdomain, 1 = entry by entry over the entries inside the
 * subdomain (the round-2 form; finds the same classes on the structured grids, four times slower);
 * "halo_overlap" 1 = several ranks, restricted combine: the subdomains that hold no dof of another rank are applied while the
 * ghost entries of r are imported on a second stream, the others after the import (same operator bit for bit); 0 (default) =
 * import, then all subdomains.  The SpMV on row classes does the same with its import of x: all rows while it travels, the
 * rows that read ghost columns once more after it has arrived.  An A/B switch for multi-GPU runs: on one GPU there is
 * nothing to hide;
 * "whole_boxes" 1 (default) = with row ghosts, a box that a rank boundary crosses is built whole (with its full
 * overlap) on every rank that owns a part of it wherever the stored rows reach, 0 = each rank takes its part;
 * "asm_zero_eps" eps > 0 = FE::doSetZeros(eps) (FE_def.hpp:74-79): element contributions of magnitude below eps are set to zero
 * before they are added, in the forms the reference thresholds -- the vector Laplacian (FEDD_FORM_LAPLACE_VEC, :719-721) and the
 * divergence blocks (fedd_assemble_div, :2002-2004, 2032-2034); 0 (default) = off;
 * "asm_p2_elem" 1 (default) = the P2 scalar forms (Laplace, vector Laplace, mass) evaluate every element once, one element per
 * wavefront with the reference gradients and quadrature weights staged in LDS (k_elem_matrix), and the rows are summed from
 * those element matrices; 0 = the pair kernels alone re-derive the row of every (row, element) pair;
 * "asm_tiles_host" 1 = the tile structures of the assembly kernel are built by the round-3 host builder instead of the device
 * kernels (A/B and tests; see fedd_mesh_setup_info). */
int fedd_set_option(fedd_ctx* ctx, const char* key, double value);

/* device-time accounting (HIP events on the context's stream around each kernel class) */
/* on = 0 off, 1 every launch, N > 1: the per-iteration classes (SpMV, Schwarz apply, coarse apply,
 * orthogonalisation) are timed on every N-th launch only and fedd_timing_get scales the sampled
 * average by the launches seen (an event pair per launch costs 3-6 % of a solve). */
int fedd_timing_enable(fedd_ctx* ctx, int on);
int fedd_timing_reset(fedd_ctx* ctx);
int fedd_timing_get(fedd_ctx* ctx, int timer, double* total_ms, int64_t* launches);
/* the launches of a class that were actually timed: their device time, their number and the algorithmic bytes their launch
 * sites stated (Gram-Schmidt sweeps: 8 bytes x rows x columns read and written; 0 for classes that state none): the
 * achieved-bandwidth figure of a class is sampled_bytes / sampled_ms, whatever the sampling stride */
int fedd_timing_get_sampled(fedd_ctx* ctx, int timer, double* sampled_ms, int64_t* sampled_launches, double* sampled_bytes);
/* calibration for the roofline figures: GB/s of a read-only stream over a scratch buffer of `bytes` (16-byte
 * non-temporal loads, `reps` launches back to back, best of three batches) on this GPU, i.e. the ceiling that
 * the measured fractions of the 8 TB/s spec can be compared with ("of spec" and "of
 * measured") */
/* column patterns of the solver's compacted SpMV stream (option "spmv_pattern", default 1): rows that repeat their column
 * offsets (col - row) share an offset list, so the stream carries 8 instead of 12 bytes per entry (values per row, exact; a
 * 16-bit pattern id per row).  n_patterns = 0: not in use (option off, no solve yet, or the matrix has no repeated rows);
 * n_rows_explicit = rows that keep explicit column indices. */
int fedd_spmv_patterns(fedd_ctx* ctx, int64_t* n_patterns, int64_t* n_rows_explicit);
/* Row classes of the solver's stream (option "spmv_classes", default on; built with the column patterns, so for matrices beyond
 * the Infinity Cache): rows with the same column pattern AND the same values, bit for bit, share a class -- on a structured
 * grid the assembled matrix has a few thousand distinct rows among millions.  The SpMV then reads a 4-byte (class, pattern) word per row and
 * the row's values from a table of at most 16384 classes instead of 8 bytes per entry; same products, same order: y identical
 * bit for bit.  n_classes = 0: not in use (fewer than 90 % of the rows repeat); n_rows_in_classes: rows served from the table;
 * nnz_streamed_rest: stream entries of the other rows.  Outputs may be NULL.  Option "spmv_keep_dictionary" 1: the pattern
 * dictionary and the classes of the previous matrix are kept while the new matrix still matches them bit for bit (one
 * verifying pass instead of the build -- for drivers that reassemble the same operator); default 0: built for every matrix.
 * Option "spmv_classes_cover" (default 90): the percentage of the rows the classes must cover to be used. */
int fedd_spmv_classes(fedd_ctx* ctx, int64_t* n_classes, int64_t* n_rows_in_classes, int64_t* nnz_streamed_rest);
/* bytes per column index of the solver's SpMV stream: 0 = column patterns in use (above), 2 = 16-bit offsets from a base per
 * window of the stream (option "spmv_col16", default 1; 10 instead of 12 bytes per entry) in every window whose columns span less
 * than 65536 -- any mesh numbered with some locality; entries_with_32bit_columns (nullable) = the entries of the windows that
 * span more (ghost columns on several ranks, far neighbours) and keep plain indices --, 4 = plain indices throughout.  Replaces
 * nothing in the reference (Tpetra's local column indices are 32-bit, feddlib/core/LinearAlgebra/Matrix_def.hpp:88-92); a
 * property of this library's stream. */
int fedd_spmv_col_bytes(fedd_ctx* ctx, int* bytes_per_column_index, int64_t* entries_with_32bit_columns);

int fedd_read_bandwidth(fedd_ctx* ctx, int64_t bytes, int reps, double* gb_per_s);

/* multi-GPU: exchange plan for the owned/ghost split (import of ghost x / r entries before SpMV
 * and Schwarz -- the Tpetra Import behind Matrix::apply, Matrix_def.hpp:245-254; GMRES dots are
 * ncclAllReduce).  After fedd_mesh_set:
 *   fedd_halo_set_owners   owner rank of every repeated node (what the Tpetra directory knows);
 *   fedd_halo_exchange_setup  swaps the request lists over RCCL and finalises the plan;
 * or, transport-agnostic (used by the gloo CPU tests): fedd_halo_requests_sizes/_get on every
 * rank, all-to-all of the lists by the caller, fedd_halo_requests_set.
 * fedd_mesh_structured_owner: owner of structured-grid nodes under the lowest-rank rule. */
/* Host-staged transport for functional tests of the N > 1 path where RCCL cannot run (several ranks
 * on one GPU, CPU-side harnesses): the same plan, pack / unpack kernels and solver flow, with the
 * two communication steps handed to the caller.  exchange: send_buf / recv_buf are host buffers laid
 * out by send_ptr / recv_ptr (in nodes; multiply by dofs).  With callbacks set, fedd_ctx_create may
 * be given nranks > 1 and a NULL ncclUniqueId.  Production runs use RCCL (no callbacks). */
typedef int (*fedd_exchange_fn)(void* user, int n_peers, const int32_t* peers, const int64_t* send_ptr,
                                const double* send_buf, const int64_t* recv_ptr, double* recv_buf, int dofs);
typedef int (*fedd_allreduce_fn)(void* user, double* buf, int n);
int fedd_comm_set_host_callbacks(fedd_ctx* ctx, fedd_exchange_fn exchange, fedd_allreduce_fn allreduce, void* user);

/* one-rank RCCL self-test on the context's device and stream: communicator, grouped send / receive, in-place
 * all-reduce, all-gather -- the call shapes of the halo import and the Gram-Schmidt reductions (what can run of the
 * RCCL path on a one-GPU box).  max_abs_err = deviation from the expected values (0 when RCCL works). */
int fedd_rccl_selftest(fedd_ctx* ctx, int n, double* max_abs_err);
/* The collectives of the N > 1 path on the context's OWN communicator, in the shapes the solver uses them: the in-place
 * all-reduce of the Gram-Schmidt reductions, a grouped send / receive with every other rank (the halo import of a block that
 * touches all others) and a ring shift, n doubles each; max_abs_err = largest deviation from the expected values (0 on one
 * rank).  A pre-flight for multi-GPU runs (bench.py calls it under a watchdog before the first step): every rank must call it. */
int fedd_comm_selftest(fedd_ctx* ctx, int n, double* max_abs_err);

int fedd_mesh_structured_owner(int dim, const int* decomp, const int* cells, int64_t n,
                               const int64_t* gid, int32_t* owner_rank);
int fedd_halo_set_owners(fedd_ctx* ctx, int64_t n_rep, const int64_t* gid_rep, const int32_t* owner_rep);
int fedd_halo_requests_sizes(fedd_ctx* ctx, int64_t* count_to_rank /*[nranks]*/);
int fedd_halo_requests_get(fedd_ctx* ctx, int64_t* gids /*concatenated by rank*/);
int fedd_halo_requests_set(fedd_ctx* ctx, const int64_t* count_from_rank /*[nranks]*/, const int64_t* gids);
int fedd_halo_exchange_setup(fedd_ctx* ctx);
int fedd_halo_plan_sizes(fedd_ctx* ctx, int* n_peers, int64_t* n_send_total, int64_t* n_recv_total);
int fedd_halo_plan_get(fedd_ctx* ctx, int32_t* peers, int64_t* send_ptr, int32_t* send_lid,
                       int64_t* recv_ptr, int32_t* recv_lid);

#ifdef __cplusplus
}
#endif
#endif /* FEDD_HIP_H */
