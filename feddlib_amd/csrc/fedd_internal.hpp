// Internal declarations shared by the translation units of libfedd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdarg>
#include "../../include/fedd_hip.h"

struct ncclComm;

namespace fedd {

void set_error(const char* fmt, ...);

#define FEDD_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            fedd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

#define FEDD_CHECK(cond, ...)                                                                   \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            fedd::set_error(__VA_ARGS__);                                                       \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

#define FEDD_TRY(expr)                                                                          \
    do {                                                                                        \
        int r__ = (expr);                                                                       \
        if (r__) return r__;                                                                    \
    } while (0)

// A device allocation that remembers its size; grows on demand, never shrinks.
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;             // owns its allocation: no copies (a copy once leaked the halo buffers)
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int ensure(size_t n) {
        if (n <= cap && p) return 0;
        if (p) {
            hipError_t e = hipFree(p);
            (void)e;
            p = nullptr;
            cap = 0;
        }
        if (n == 0) n = 1;
        FEDD_HIP(hipMalloc((void**)&p, n * sizeof(T)));
        cap = n;
        return 0;
    }
    void release() {
        if (p) {
            hipError_t e = hipFree(p);
            (void)e;
        }
        p = nullptr;
        cap = 0;
    }
};

constexpr int MAX_BC = 16;
constexpr int MAX_DOFS = 3;
constexpr int SPMV_PAT_LMAX = 48;       // longest offset list of the SpMV pattern table (spmv.hip SPAT_L)
constexpr int SCHWARZ_NMAX = 256;  // largest overlapping subdomain (dofs) the register / LDS dense kernels take
constexpr int SCHWARZ_NMAX_BIG = 1024;  // ... and the batched matrix-core inversion of the large-subdomain path

struct HaloPlan {
    std::vector<int32_t> peers;
    std::vector<int64_t> send_ptr, recv_ptr;   // per peer, in dofs-per-node = 1 units (nodes)
    std::vector<int32_t> send_lid, recv_lid;   // owned node ids to pack / ghost node ids to fill
    std::vector<int64_t> req_count, req_gid;   // what this rank asks of every other rank (by rank, gid asc)
    DevBuf<int32_t> d_send_lid, d_recv_lid;
    DevBuf<double> d_send_buf, d_recv_buf;     // sized for MAX_DOFS * nodes
    bool ready = false;
    // back to the empty plan; the device buffers are kept (DevBuf grows on demand and is freed with the context)
    void reset() {
        peers.clear();
        send_ptr.clear();
        recv_ptr.clear();
        send_lid.clear();
        recv_lid.clear();
        req_count.clear();
        req_gid.clear();
        ready = false;
    }
};

// a device CSR matrix held beside the system matrix (blocks of a mixed problem before the merge)
struct DevCsr {
    DevBuf<int32_t> rowptr, colind;
    DevBuf<double> val;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int max_row_nnz = 0;
    bool valid = false;
};
constexpr int MAX_AUX = 4;

// regular lattice of the coarse level: g cells and np = g + 1 points per direction (1 point in
// directions beyond dim), over the global bounding box [lo, lo + L]
struct CoarseGeom {
    int dim = 0;
    int g[3] = {1, 1, 1}, np[3] = {1, 1, 1};
    double lo[3] = {0, 0, 0}, L[3] = {1, 1, 1};
    int reduced = 0;   // GDSW family: 1 = RGDSW (coarse dofs on the coarse nodes only), 0 = GDSW (every interface entity)
};
constexpr int COARSE_MAX_DOFS = 8192;   // dense K0^-1: 8192^2 * 8 B = 512 MiB

struct TimerSlot {
    double total_ms = 0.0;
    int64_t launches = 0;   // launches that were timed
    int64_t seen = 0;       // launches that passed through (timed or not)
    double bytes = 0.0;     // algorithmic bytes of the timed launches (classes whose launch sites state them)
    struct Pair {
        hipEvent_t first, second;
        bool cont;          // second half of a launch that was interrupted (e.g. by a collective): time, not a launch
    };
    std::vector<Pair> pending;
};

}  // namespace fedd

struct fedd_ctx {
    int device = -1;  // < 0: host-only context (no HIP calls; host logic tests)
    int rank = 0, nranks = 1;
    hipStream_t stream = nullptr;
    ncclComm* comm = nullptr;
    // host-staged transport (functional testing of the N > 1 path without RCCL, e.g. over gloo)
    fedd_exchange_fn cb_exchange = nullptr;
    fedd_allreduce_fn cb_allreduce = nullptr;
    void* cb_user = nullptr;
    std::vector<double> h_send, h_recv;

    // ---- mesh (column-local numbering: owned nodes [0,n_own) in unique-map order, then ghosts
    //      sorted by global id) ----
    int dim = 0, nen = 0;
    int64_t n_elem = 0, n_own = 0, n_node = 0;  // n_node = owned + ghost
    int64_t n_rowg = 0;                         // ghost nodes [n_own, n_own + n_rowg) whose rows are complete here (fedd_mesh_set_rows)
    std::vector<int64_t> h_node_gid;            // [n_node]
    std::vector<int32_t> h_ghost_owner;         // [n_node - n_own]
    fedd::DevBuf<int32_t> d_conn;               // [n_elem*nen]
    fedd::DevBuf<double> d_xyz;                 // [n_node*dim]
    fedd::DevBuf<int32_t> d_flag;               // [n_own] bc flags of owned nodes

    // ---- node -> (element, local index) adjacency of owned nodes ----
    fedd::DevBuf<int32_t> d_n2e_ptr, d_n2e;     // [n_own+1], [sum]
    int max_deg = 0;
    bool have_adj = false;

    // ---- CSR (dof level, owned rows) ----
    int dofs = 0, block_mode = 0;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    // with row ghosts the CSR arrays continue past the owned rows: rows [n_rows, n_rows_ext) are those of the
    // row-ghost nodes (read by the Schwarz local matrices only); n_rows / nnz stay the owned part
    int64_t n_rows_ext = 0, nnz_ext = 0;
    int max_row_nnz = 0;
    fedd::DevBuf<int32_t> d_rowptr, d_colind;
    fedd::DevBuf<double> d_val;
    fedd::DevBuf<double> d_rhs, d_x;            // [n_rows]
    fedd::DevBuf<double> d_xcol, d_ycol;        // [n_cols] work vectors with ghost tail
    fedd::DevBuf<int32_t> d_isdir;              // [n_rows] 1 = Dirichlet row
    bool have_pattern = false;
    int whole_boxes = 1;                        // row ghosts: build boxes whole across rank boundaries where the stored rows reach
    fedd::DevBuf<int32_t> d_fbin_ptr, d_fbin_nodes;   // [nsub+1], [row-ghost dofs] other ranks' dofs grouped by box
    int32_t sw_max_size_all = 0;                // largest subdomain over all ranks
    int spmv_nt = -1;                           // window SpMV streams the matrix non-temporally: -1 = if larger than the Infinity Cache, 0 / 1
    int asm_lds_kb = 37;                        // LDS budget of the assembly kernel's contribution park (KB): 4 workgroups per CU
    int box_kind = 0;                           // Schwarz boxes: 0 = one lattice over all ranks' nodes, 1 = per-rank lattice
    int spmv_kind = 0;                          // 0 = CSR-window / automatic, 1 = row-per-lane-group, 2 = CSR-stream
    int asm_dbg = 0;                            // ablation switches of the assembly kernel (development)
    int asm_u = 1;                              // slot-addressed assembly: pairs per lane with their loads in flight together (P1)
    int asm_kind = 0;                           // 0 = slot-addressed pair-parallel assembly, 2 = pair-parallel with slot sweep, 1 = lane-per-row gather
    // element-major tile structures of the current mesh (assemble.hip build_tiles): 0 = not built, 1 = ready, -1 = mesh does not fit
    int asm_p2_elem = 1;                        // option "asm_p2_elem": P2 scalar forms take their element matrices from k_elem_matrix (one element per wavefront); 0 = pair kernels alone
    int p2_state = 0;                           // gather lists of the P2 row sums (assemble.hip k_p2_lists): 0 = not built, 1 = ready, -1 = not applicable
    fedd::DevBuf<uint16_t> d_p2_soff, d_p2_src; // per node-level nonzero the start of its sources; the sources (adjacency entry << 4 | local column)
    fedd::DevBuf<double> d_ke;                  // [n_elem * nen * nen] element matrices of the assembly in progress (P2)
    double asm_zero_eps = 0.0;                  // option "asm_zero_eps": FE::doSetZeros(eps) -- element contributions below eps are dropped (vector Laplacian, B, B^T); 0 = off
    int asm_tiles_host = 0;                     // option "asm_tiles_host": 1 = the tile structures are built on the host (the round-3 builder; A/B and tests), 0 = on the device
    double tl_build_ms = 0.0, adj_build_ms = 0.0;   // wall ms of the per-mesh structures of the current mesh: tile build, node -> element adjacency
    int tl_state = 0, asm_tiles = 1;            // option "asm_tiles": the P1 Laplace / elasticity forms take the element-major tile kernel (0: pair kernels)
    int64_t tl_ntile = 0;
    int tl_max_el = 0, tl_max_ext = 0, tl_max_blob = 0;
    fedd::DevBuf<uint32_t> tl_hdr, tl_blob;     // per-tile headers (16 bytes each) and blobs (assemble.hip TileHdr)
    fedd::DevBuf<uint32_t> tl_shape;            // [tl_ntile] first word of the shape part every tile reads (its own or an earlier tile's)
    int tl_nshared = 0;                         // tiles that read the shape of an earlier tile
    fedd::DevBuf<int32_t> d_pat_stash;          // pattern build: merged node lists of the count pass, [k][node]
    fedd::DevBuf<int32_t> d_spmv_rows;          // CSR-stream: first row of every nnz window
    bool spmv_rows_ready = false;
    // solver-private compacted copy of the owned rows: the entries that are exactly 0.0 left out (the structural
    // zeros the reference's insertGlobalValues keeps in the pattern, and the zeroed entries of Dirichlet rows).
    // fedd_csr_get keeps returning the reference pattern; SpMV streams this one (same y bit for bit for finite x).
    int spmv_compact = 1;                       // option "spmv_compact": 1 = on (default), 0 = stream the parity CSR
    double spmv_drop_tol = 2.220446049250313e-16;   // option "spmv_drop_tol": drop |a_ij| <= tol * max_k |a_ik|; 0 = exact zeros only
    int spmv_exact_public = 1;                  // option "spmv_exact_public": fedd_spmv multiplies with the parity CSR (every stored entry), not with the solver's compacted stream
    bool cs_valid = false;                      // false after anything that writes d_val / the pattern
    int64_t cs_nnz = 0;
    int32_t cs_tot32 = 0;
    fedd::DevBuf<int32_t> d_cs_rowptr, d_cs_col, d_cs_rows, d_cs_wincnt;
    fedd::DevBuf<uint16_t> d_cs_col16;          // 16-bit column offsets of the compacted stream (k_cs_col16), d_cs_wbase their bases per window
    fedd::DevBuf<int32_t> d_cs_wbase;
    bool cs_col16 = false;
    int spmv_col16 = 1;                         // option "spmv_col16": 16-bit columns in the per-entry window SpMV where the windows allow it
    fedd::DevBuf<double> d_cs_val;
    int spmv_pattern = 1;                       // option "spmv_pattern": rows that repeat their column offsets share a pattern (spmv.hip)
    int spmv_win_nu = 0;                        // option "spmv_win_nu": entries per lane of k_spmv_win on the compacted stream (4 ... 8; 0 = by row length)
    int cs_win_nu = 8;
    int spmv_pat_nu = 0;                        // option "spmv_pat_nu": 16-byte loads per lane of k_spmv_pat (2 ... 8; 0 = by the usual row length)
    int cs_pat_len = fedd::SPMV_PAT_LMAX;           // longest pattern in the table
    int cs_pat_nu = 8;                          // pattern SpMV: window = 256 * cs_pat_nu values
    fedd::DevBuf<int32_t> d_cs_prows;           // first row of each of those windows
    int32_t cs_max_len = 1;                     // longest row of the compacted stream
    int32_t cs_npat = 0, cs_nexpl = 0;          // patterns in use (0: dictionary off), rows that keep explicit columns
    fedd::DevBuf<uint64_t> d_cs_hash;           // row hashes [n] | table keys
    fedd::DevBuf<int32_t> d_cs_pati;            // slot of row [n] | table min row | pattern of slot | lengths | offset lists | counters
    fedd::DevBuf<uint16_t> d_cs_pat;            // pattern id per row (0xffff: explicit columns)
    int spmv_classes = 1;                       // option "spmv_classes": rows that repeat their column pattern AND their values bit for bit share a class (spmv.hip k_spmv_cls)
    int32_t cs_cls_tab_n = -1, cs_cls_tab_len = 0, cs_cls_tab_ncls = 0, cs_cls_tab_rows = 0;   // the class table and row words on the device: rows, stride, classes and classed rows of the build they come from (kept while the next matrix still matches them bit for bit)
    int32_t cs_pat_tab_n = -1, cs_pat_tab_npat = 0, cs_pat_tab_nexpl = 0, cs_pat_tab_len = 0;   // the column-pattern dictionary on the device: rows, patterns, explicit rows, longest pattern of the build it comes from (kept while the next matrix still matches it)
    int spmv_cls_cover = 90;                    // option "spmv_classes_cover": the classes are used when they cover at least this percentage of the rows
    int spmv_keep_dict = 0;                     // option "spmv_keep_dictionary": 1 = the previous matrix' pattern dictionary and row classes are kept while the new stream matches them bit for bit (one verifying pass instead of the build: time loops that reassemble the same operator); 0 (default) = built per matrix
    int32_t cs_nbnd = -1;                       // rows of the compacted stream that read ghost columns (-1: not listed)
    fedd::DevBuf<int32_t> d_cs_bnd;             // flags / positions [n + 2] | the rows
    int cs_cls_len = 8;                         // stride of the class table (8, 16 or 48 values)
    int32_t cs_ncls = 0, cs_cls_rows = 0, cs_cls_rest = 0;   // classes in use (0: off), rows in a class, stream entries of the other rows
    fedd::DevBuf<uint32_t> d_cs_cls;            // per row: class << 8 | column pattern (0xffffffff: none)
    fedd::DevBuf<uint16_t> d_cs_clspat;         // column pattern of a class
    fedd::DevBuf<double> d_cs_clsval;           // [classes][8] values of a class
    fedd::DevBuf<int32_t> d_cs_clsi;            // slot of row [n] | table min row | class of slot | counters
    fedd::DevBuf<uint64_t> d_cs_clskey;         // table keys
    fedd::DevCsr aux[fedd::MAX_AUX];            // stored blocks (A, B, B^T, C) of a mixed problem
    bool merged = false;                        // system matrix = merged blocks (dof -> node map below)
    int64_t merged_nA = 0;                      // rows of block row 0
    int merged_dofsA = 1;                       // dofs per node of block row 0
    fedd::DevBuf<int32_t> d_dof_node;           // [n_rows] node whose coordinates place the dof (merged systems)

    // ---- Schwarz ----
    int sw_target = 0;                          // nodes per box; 0 = default (27 / dofs per node)
    double sw_scale = 1.0;
    int sw_overlap = 1, sw_combine = 0;
    int64_t sw_nsub = 0, sw_max_size = 0, sw_max_own = 0, sw_inv_elems = 0;
    int apply_kind = 0;                         // restricted apply: 0 = flat streaming kernel, 1 = strided (A/B)
    int inv_kind = 0;                           // local inverses: 0 = scalar-pivot kernel (drops finished rows), 1 = MFMA block sweep, 2 = scalar-pivot, all rows (A/B)
    int ghost_overlap = 1;                      // subdomains may contain ghost dofs (identity rows): 1 = yes (A/B)
    int gmres_kind = 2;                         // Gram-Schmidt: 0 = delayed second pass (DCGS2), 1 = two passes (CGS2), 2 = s-step blocks (BCGS-PIP2)
    int gmres_s = 0;                            // s-step GMRES: Krylov vectors per block (1 ... 16; 0 = 16 from 1.2 M rows per rank, else 8)
    int gmres_tol_blocks = 1;                   // ... tolerances below 1e-9 / 1e-11 take blocks of at most 5 / 3 vectors (0: no cap)
    int gmres_dot_gy = 0;                       // ... column groups in flight per row block of the block dot kernel (0 = by vector length) (A/B)
    int gmres_dotv = 0;                         // ... launch shape of the block dot kernel (A/B)
    int gmres_s_used = 0;                       // ... the block length of the last solve
    int gmres_spec = 0;                         // ... operator applications of the next block issued before the host reads a block's outcome (A/B switch for multi-GPU runs)
    int gmres_newton = 1;                       // ... Newton block basis (Leja-ordered Ritz shifts) once the first s Arnoldi steps exist; 0 = monomial, blocks <= 8
    double gmres_chol_tol = 1e-13;              // ... a block is cut where the squared sine of a new vector against its predecessors falls to this
    int gmres_blocks = 0, gmres_cut_blocks = 0; // ... blocks / blocks that were cut in the last solve
    int gmres_floor = 0;                        // ... the last solve stopped at the rounding floor of b - A x (relres returned = true residual, may exceed rtol)
    double gmres_rec_relres = -1.0;             // ... and the recurrence residual it had reached then (-1: not that case)
    int gm_x0 = 0;                              // the next solve starts from the vector in d_x ("Zero Initial Guess" = false), set per call by fedd_gmres_x0
    fedd::DevBuf<int32_t> d_node_bin;           // [n_own] compact bin id of each owned node
    fedd::DevBuf<int32_t> d_bin_ptr, d_bin_nodes;   // [nsub+1], [n_own]
    fedd::DevBuf<int32_t> d_sub_n, d_sub_nown;  // [nsub] total / owned dofs of each subdomain
    fedd::DevBuf<int32_t> d_sub_dofs;           // [nsub*NMAX] local dof ids (owned first)
    fedd::DevBuf<int64_t> d_inv_ptr;            // [nsub+1] offsets into d_inv (elements)
    fedd::DevBuf<double> d_inv;                 // per subdomain [n_i][rp_i] column-major slab
    fedd::DevBuf<double> d_mult;                // [n_cols] multiplicity (averaging)
    bool have_schwarz = false;
    int md2_nch = 4;                            // k_multidot2: 512-row chunks per workgroup (2 or 4)
    int md2_gy = 0;                             // k_multidot2: column groups in flight per row block (0 = by vector length)
    int halo_overlap = 0;                       // several ranks: interior subdomains first, ghost import of r on a second stream meanwhile
    hipStream_t stream2 = nullptr;              // (created on first use)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int64_t sw_nconf = 0;                       // subdomains whose dof list is their representative's list shifted (ids computed in the apply)
    int64_t sw_nint = -1;                       // subdomains without ghost dofs at the front of d_sw_order (-1: not split)
    int apply_dbg = 0;                          // ablation bits of k_apply_mfma<4, 12> (development; wrong results by design)
    int apply_span = 0;                         // grouped apply: subdomains per workgroup (0 = 64)
    int sw_dedupe = 1;                          // option "schwarz_dedupe": subdomains with the same local matrix share one slab
    int sw_fp_kind = 0;                         // option "schwarz_fp_kind": fingerprints from row hashes (0) or entry by entry (1)
    int64_t sw_nrep = 0;                        // distinct local matrices (= slabs) of the last setup
    fedd::DevBuf<double> d_sw_rmax;             // [n_rows_ext] largest magnitude of every stored row
    fedd::DevBuf<uint64_t> d_sw_fp;             // fingerprints [2 nsub] | hash table keys [2 tsize]
    fedd::DevBuf<int32_t> d_sw_order;           // subdomains sorted by representative [nsub] | sort scratch [2 nsub] | pad | records int4[nsub]
    int64_t sw_order_off = 0;                   // where the records start (in int32 units, 16-byte aligned)
    fedd::DevBuf<int32_t> d_sw_bt;              // batch table of k_apply_bt: BT_W ints per batch (schwarz.hip)
    fedd::DevBuf<int32_t> d_sw_btw;             // ... its build scratch
    int64_t sw_nbatch = 0;                      // batches in the table (0: no table, the chunk-record kernel runs)
    int64_t sw_nbatch_int = 0;                  // ... of which belong to the subdomains without ghost dofs (split order)
    int apply_bt = 1;                           // build and use the batch table (option "apply_bt")
    fedd::DevBuf<int32_t> d_sw_replist;         // the representatives (the subdomains that are inverted)
    fedd::DevBuf<int32_t> d_sw_rep;             // representative [nsub] | sizes for the inversion [nsub] | slot [nsub] | table min [tsize]
    int sw_big = -1;                            // large-subdomain path: -1 = for merged block systems, 0 = never, 1 = always
    bool sw_big_active = false;                 // the current preconditioner was built by schwarz_setup_big
    int sw_big_target = 0;                      // owned dofs per box of the bisection (0 = default 120)
    fedd::DevBuf<double> d_big_ws;              // dense matrices of one chunk of subdomains
    fedd::DevBuf<int32_t> d_big_nblk;           // 64-blocks per subdomain
    fedd::DevBuf<int32_t> d_big_ids, d_big_pos; // subdomains taken by the batched dense inversion (compacted ids)

    // ---- coarse level (two-level Schwarz) ----
    int sw_two_level = 0;
    double co_cells_target = 0.0;               // 0 = default
    fedd::CoarseGeom co_geom;
    int64_t co_ncell = 0, co_nlat = 0, co_n0 = 0, co_ld = 0;
    fedd::DevBuf<int32_t> d_co_key[2], d_co_val[2];   // radix split ping-pong (cell id, node)
    fedd::DevBuf<int32_t> d_co_cell_ptr;        // [ncell+1]; nodes of cell: d_co_val[co_sorted]
    int co_sorted = 0;
    fedd::DevBuf<double> d_co_mask;             // [n_cols] 1 = free dof, 0 = Dirichlet
    fedd::DevBuf<double> d_co_cellK;            // per-cell Galerkin blocks
    fedd::DevBuf<double> d_co_K;                // [ld*ld] K0, then K0^-1
    fedd::DevBuf<double> d_dense_ws;            // dense_invert_batched: per matrix Dinv [64*64] | R [64*ld] | C [ld*64]
    fedd::DevBuf<double> d_co_part, d_co_r0, d_co_z0;
    bool have_coarse = false;
    int co_kind = FEDD_COARSE_Q1;               // FEDD_COARSE_Q1 (lattice hat functions) / FEDD_COARSE_GDSW
    double gdsw_tol = 0.0;                      // option "gdsw_tol": relative residual of the interior extension solves; 0 = GDSW 1e-4, RGDSW 1e-3
    int gdsw_ext_its = 0;                       // most iterations / largest final residual of the last setup's extension solves
    double gdsw_ext_rel = 0.0;
    fedd::DevBuf<int32_t> d_gd_ent;             // [n_own] interface entity (doubled-lattice id) of every owned node
    fedd::DevBuf<double> d_gd_phi;              // [n_rows * (3^dim - 1) * dofs] coarse basis, row-wise by class of the home cell
    fedd::DevBuf<double> d_gd_imask;            // [n_rows] 1 = free interior dof
    fedd::DevBuf<double> d_gd_tmp;
    fedd::DevBuf<double> d_gd_stack;            // stacked vectors of the extension solves (gdsw_block)

    // ---- GMRES workspace ----
    fedd::DevBuf<double> d_V, d_Z;              // [(m+1)*n_rows], [n_cols] scratch
    fedd::DevBuf<double> d_w;                   // [n_rows]
    fedd::DevBuf<double> d_part;                // dot partials
    fedd::DevBuf<double> d_small;               // H, cs, sn, g, h, scalars
    double* h_pinned_map = nullptr;             // device pointer of h_pinned (kept while the option switches h_pinned_dev off)
    double* h_pinned_dev = nullptr;             // the same buffer as the device sees it (nullptr: not mapped, copies are used)
    double* h_pinned = nullptr;                 // small pinned host mirror
    int gm_restart_alloc = 0;
    int64_t gm_V_ldv = -1;                      // leading dimension the zeroed padding rows of d_V belong to (s-step solver)
    int gm_nr = 0;                              // > 1: GMRES runs on stacked vectors X[row * gm_nr + j] (multi.hip; the GDSW extension solves)
    int multi_ch = 4;                           // option "multi_ch": matrix-core steps per flight of gathers in k_apply_multi (4, 8, 16)
    int pat_hash = 1;                           // option "pat_hash": 1 = hashed node-pattern merge for vertex-only elements (symbolic.hip)
    int gmres_fused_blocks = 0;                 // blocks of the last s-step solve whose first update and second dot ran as one sweep
    int gmres_fuse = -1;                        // option "gmres_fuse": s-step solver, first update and second dot of a block in one sweep (k_blockfuse); -1 = from 4 M rows on
    int gdsw_rot = 0;                           // option "gdsw_rotations": rotations in the null space of vector problems (dofs = dim)
    int co_nns = 1;                             // coarse functions per entity of the last GDSW setup (dofs, or dofs + rotations)
    fedd::DevBuf<double> d_gd_gram;             // [entities * 21] Gram matrices of the entities' functions (selection of the independent ones)
    fedd::DevBuf<int32_t> d_gd_keep;            // [entities] bit masks of the functions kept
    int gdsw_block = 1;                         // option "gdsw_block": 1 = extension solves sixteen columns at a time, 0 = one by one
    const double* gm_mask = nullptr;            // != nullptr: GMRES solves the constrained system (dofs with mask 0 held), see gmres.hip

    // ---- generic scratch ----
    fedd::DevBuf<int32_t> d_itmp0, d_itmp1, d_itmp2;
    fedd::DevBuf<int64_t> d_scan[3];            // block sums of the device scan, one per level
    fedd::DevBuf<int32_t> d_rs_hist;            // radix sort: digit histograms [256][tiles]
    fedd::DevBuf<double> d_dtmp0;
    fedd::DevBuf<int32_t> d_flags;              // [16] device flags: 0 max scratch, 1 bad pivot, 2 DGKS gate, 3-5 coarse setup, 8-11 Schwarz setup counters, 12 s-step block length

    fedd::HaloPlan halo;

    // ---- timing ----
    bool timing = false;
    int timing_stride = 1;
    fedd::TimerSlot timers[FEDD_T_COUNT];
    std::vector<hipEvent_t> ev_pool;            // timer events waiting for reuse (ScopedTimer::take, timing_flush)
};

namespace fedd {

// RAII-ish helper: records a start event at construction and a stop event at stop().
struct ScopedTimer {
    fedd_ctx* c;
    int id;
    hipEvent_t a = nullptr, b = nullptr;
    bool sampled = false, cont = false;
    ScopedTimer(fedd_ctx* ctx, int timer) : c(ctx), id(timer) {   // timer < 0: no-op
        if (c->timing && timer >= 0) {
            // the per-iteration classes are sampled every timing_stride-th launch: an event pair
            // around each of several hundred small launches per solve costs a few percent
            const bool per_iteration = timer == FEDD_T_SPMV || timer == FEDD_T_SCHWARZ_APPLY ||
                                       timer == FEDD_T_ORTHO || timer == FEDD_T_COARSE_APPLY ||
                                       timer == FEDD_T_HALO || timer == FEDD_T_ALLREDUCE || timer == FEDD_T_GS_DOT ||
                                       timer == FEDD_T_GS_UPDATE || timer == FEDD_T_GS_FUSED;
            // (the s-step solver launches its sweeps once per block, each over a different number of columns: all timed)
            const bool block_sweeps = c->gmres_kind == 2 && (timer == FEDD_T_ORTHO || timer == FEDD_T_GS_DOT || timer == FEDD_T_GS_UPDATE || timer == FEDD_T_GS_FUSED);
            const int64_t k = c->timers[id].seen++;
            if (per_iteration && !block_sweeps && c->timing_stride > 1 && k % c->timing_stride != 0) return;
            sampled = true;
            if (take(&a) && take(&b)) (void)hipEventRecord(a, c->stream);
        }
    }
    // events come from the context's pool (timing_flush returns them): creating a pair per timed launch cost about 2 ms of
    // host time per 100 ms step -- inside the timed region of the bench
    bool take(hipEvent_t* e) {
        if (!c->ev_pool.empty()) {
            *e = c->ev_pool.back();
            c->ev_pool.pop_back();
            return true;
        }
        return hipEventCreate(e) == hipSuccess;
    }
    // algorithmic bytes of this launch (counted when the launch is timed)
    void bytes(double nbytes) {
        if (sampled && !cont) c->timers[id].bytes += nbytes;
    }
    void stop() {
        if (c->timing && a && b) {
            (void)hipEventRecord(b, c->stream);
            c->timers[id].pending.push_back({a, b, cont});
            a = b = nullptr;
        }
    }
    // after stop(): time the rest of the same launch (same sampling decision, not another launch)
    void resume() {
        if (c->timing && sampled && !a) {
            cont = true;
            if (take(&a) && take(&b)) (void)hipEventRecord(a, c->stream);
        }
    }
    ~ScopedTimer() { stop(); }
};

int timing_flush(fedd_ctx* c);

// device utilities (scan.hip)
int exclusive_scan_i32(fedd_ctx* c, const int32_t* d_in, int32_t* d_out, int64_t n, int64_t* total_out);
int exclusive_scan_i64(fedd_ctx* c, const int64_t* d_in, int64_t* d_out, int64_t n, int64_t* total_out);
int reduce_max_i32(fedd_ctx* c, const int32_t* d_in, int64_t n, int32_t* out);
// stable radix sort of (key, value) pairs on bits [0, bits) of the non-negative keys; keys[0] / vals[0] = input, the passes
// ping-pong between the two buffers, *cur_out = the one that holds the result
int radix_sort_pairs_i32(fedd_ctx* c, int32_t* keys[2], int32_t* vals[2], int32_t n, int bits, int* cur_out);

// symbolic.hip
int build_adjacency(fedd_ctx* c);
int build_pattern(fedd_ctx* c, int dofs, int block_mode);

// assemble.hip
int assemble_matrix(fedd_ctx* c, int form, const double* params);
int assemble_rhs(fedd_ctx* c, int dofs, const double* f_const, int extra_degree);
int apply_dirichlet(fedd_ctx* c, int n_bc, const int32_t* flags, const int32_t* comp_mask,
                    const double* values);
int apply_dirichlet_nodes(fedd_ctx* c, int64_t n, const int32_t* nodes, const int32_t* comp_mask,
                          const double* values);
int apply_dirichlet_rows(fedd_ctx* c, int64_t n, const int32_t* rows, const double* values);
int assemble_div(fedd_ctx* c, int64_t n_pressure_nodes, int slot_b, int slot_bt);

// blocks.hip
int matrix_store(fedd_ctx* c, int slot);
int matrix_scale(fedd_ctx* c, int slot, double alpha);
int block_merge(fedd_ctx* c, int slot_a, int slot_bt, int slot_b, int slot_c);

// spmv.hip
// x_has_tail: the buffer behind d_x_owned has room for all n_cols entries; the ghost values are then imported in
// place (behind the owned entries) instead of into a copy of x
// d_sub != nullptr: y = A x - theta * d_sub (owned rows), fused into the kernel's store
// use_compact: -1 = as option "spmv_compact" says, 0 = the parity CSR (every stored entry), 1 = the solver's compacted stream
int spmv_owned(fedd_ctx* c, const double* d_x_owned, double* d_y_owned, bool x_has_tail = false, const double* d_sub = nullptr,
               double theta = 0.0, int use_compact = -1);   // incl. ghost import
int halo_import(fedd_ctx* c, double* d_xcol, int dofs);                    // fill ghost tail
int read_bandwidth(fedd_ctx* c, int64_t bytes, int reps, double* gbs);     // read-only streaming calibration

// schwarz.hip
int schwarz_setup(fedd_ctx* c);
int schwarz_apply(fedd_ctx* c, const double* d_r_owned, double* d_z_owned, bool r_has_tail = false);
int bounding_box(fedd_ctx* c, int64_t n_nodes, double lo[3], double hi[3], double* d_scratch = nullptr);   // of d_xyz[0, n_nodes)
int global_box(fedd_ctx* c, int64_t n_own, double lo[3], double hi[3], double* n_global);   // over all ranks

// schwarz_big.hip: subdomains of up to 1024 dofs (balanced coordinate-bisection boxes, batched dense inverses on the
// f64 matrix cores); the default for merged block systems (option "schwarz_big": -1 auto, 0 never, 1 always)
bool schwarz_use_big(const fedd_ctx* c);
int schwarz_setup_big(fedd_ctx* c);
int schwarz_apply_big(fedd_ctx* c, const double* d_r_owned, double* d_z_owned, bool r_has_tail);
int schwarz_overlap_lists_big(fedd_ctx* c, int64_t nsub, int32_t* max_n, int32_t* max_own);
int schwarz_slab_offsets(fedd_ctx* c, int64_t nsub, int restricted);
int schwarz_dense_batched(fedd_ctx* c, int64_t nsub, int dstride, int n_lo, int32_t p_off, int restricted, int max_n,
                          int32_t* d_bad, const int32_t* d_sub_n_sel);

// invert_mfma.hip: local inverses of plain systems, n <= 128, on the f64 matrix cores
int schwarz_invert_mfma(fedd_ctx* c, int restricted, int32_t* d_bad, int max_n);

// dense.hip: in-place inverses of a batch of dense matrices on the f64 matrix cores (blocked Gauss-Jordan, no pivoting)
int dense_invert_batched(fedd_ctx* c, double* K, int64_t ld, int batch, int64_t stride, const int32_t* d_nblk,
                         int max_nblk, int spd, int32_t* d_bad);

// coarse.hip
int coarse_setup(fedd_ctx* c);
int coarse_apply_add(fedd_ctx* c, const double* d_r_owned, double* d_z_owned);   // z += Phi K0^-1 Phi^T r

// multi.hip: operator and one-level Schwarz preconditioner on sixteen stacked right-hand sides, X[row * MULTI_NR + j]
constexpr int MULTI_NR = 16;
bool multi_rhs_ok(const fedd_ctx* c);
int spmm_owned(fedd_ctx* c, double* d_X, double* d_Y, const double* mk, const double* alt);
int schwarz_apply_multi(fedd_ctx* c, double* d_R, double* d_Z, const double* mk);

// gmres.hip
int gmres_solve(fedd_ctx* c, const double* d_b, double* d_x, double rtol, int max_it, int restart,
                int use_prec, int* its_out, double* relres_out);
int allreduce_sum(fedd_ctx* c, double* d_buf, int n);

// FE tables (fe_tables.cpp): reference-simplex quadrature and basis values
struct FeTables {
    int dim = 0, nen = 0, nq = 0;
    std::vector<double> w;      // [nq]
    std::vector<double> phi;    // [nq*nen]
    std::vector<double> dphi;   // [nq*nen*dim]
};
int fe_quadrature(int dim, int degree, std::vector<double>& pts, std::vector<double>& w);
int fe_tables(int dim, int nen, int degree, FeTables& out);
int fe_degree(int nen, int dim, bool grad);   // determineDegree building block: P1 Std 1/Grad 0, P2 2/1

}  // namespace fedd
