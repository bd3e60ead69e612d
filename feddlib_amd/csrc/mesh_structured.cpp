// Host-side structured P1 mesh generator (square / cube, Kuhn split) of the product path.
//
// Follows the behaviour of the reference generator, not its code:
//   MeshStructured::buildMesh2D P1 branch   feddlib/core/Mesh/MeshStructured_def.hpp:348-463
//   MeshStructured::buildMesh3D P1 branch   feddlib/core/Mesh/MeshStructured_def.hpp:703-806
//   setStructuredMeshFlags(1)               feddlib/core/Mesh/MeshStructured_def.hpp:2974-3203
//   Map::buildUniqueMap                     feddlib/core/LinearAlgebra/Map_def.hpp:184-210
// Layout differences: flat SoA arrays instead of vector<vector<>>, per-direction block counts
// (the reference only knows N x N x N), optional ghost-element layers, lowest-rank owner rule.
// Ghost layers L: 0 = the reference's block; 1 = plus the layer of cells above the block that completes the
// rows of the owned nodes; L >= 2 = L layers of cells around the owned nodes on every side that has a
// neighbour, which completes the rows of the ghost nodes within L - 1 layers as well ("row ghosts": the
// Schwarz local matrices then hold true rows for nodes that belong to other ranks).
#include "fedd_internal.hpp"
#include <cmath>
#include <limits>

namespace {

struct Block {
    int dim;
    int N[3], M[3], off[3];
    int up[3];        // ghost lattice planes above the block in d (0 .. L, clipped to the grid)
    int dn[3];        // planes below (0 .. L - 1, clipped to the grid)
    int mode;         // L
    int lo_owned[3];  // first owned lattice index in d (0, or 1 when a lower neighbour owns the face)
    int64_t P[3];     // global points per direction
    int n1[3];        // own lattice points per direction (M+1)
    int ne[3];        // extended lattice points per direction (dn + M+1 + up)
};

int make_block(int dim, const int* decomp, const int* cells, int rank, int ghosts, Block& b) {
    b.dim = dim;
    b.mode = ghosts;
    if (ghosts < 0 || ghosts > 8) {
        fedd::set_error("structured mesh: %d ghost layers (0 .. 8)", ghosts);
        return 1;
    }
    int64_t nr = 1;
    for (int d = 0; d < 3; ++d) {
        b.N[d] = d < dim ? decomp[d] : 1;
        b.M[d] = d < dim ? cells[d] : 0;
        if (d < dim && (b.N[d] < 1 || b.M[d] < 1)) {
            fedd::set_error("structured mesh: decomp and cells must be >= 1 (H/h is too small)");
            return 1;
        }
        nr *= b.N[d];
    }
    if (rank < 0 || rank >= nr) {
        fedd::set_error("structured mesh: rank %d outside decomposition of %lld blocks", rank, (long long)nr);
        return 1;
    }
    // rank -> block offsets: x fastest (MeshStructured_def.hpp:712-722)
    b.off[0] = rank % b.N[0];
    b.off[1] = (rank / b.N[0]) % b.N[1];
    b.off[2] = rank / (b.N[0] * b.N[1]);
    for (int d = 0; d < 3; ++d) {
        b.P[d] = d < dim ? (int64_t)b.N[d] * b.M[d] + 1 : 1;
        b.n1[d] = d < dim ? b.M[d] + 1 : 1;
        // planes above: L, below: L - 1 (the lower neighbour owns the shared face), clipped to the grid
        const int64_t above = d < dim ? (int64_t)(b.N[d] - 1 - b.off[d]) * b.M[d] : 0;
        const int64_t below = d < dim ? (int64_t)b.off[d] * b.M[d] : 0;
        b.up[d] = (int)std::min<int64_t>(ghosts, above);
        b.dn[d] = (int)std::min<int64_t>(ghosts > 1 ? ghosts - 1 : 0, below);
        b.ne[d] = b.dn[d] + b.n1[d] + b.up[d];
        b.lo_owned[d] = (d < dim && b.off[d] > 0) ? 1 : 0;
    }
    return 0;
}

// lattice indices run over [-dn, M + up] per direction (0 .. M = the reference's own block)
inline bool in_own_lattice(const Block& b, int r, int s, int t) {
    return r >= 0 && r < b.n1[0] && s >= 0 && s < b.n1[1] && t >= 0 && t < b.n1[2];
}
// the nodes whose rows are complete with L >= 2 layers: both cells around the plane are present (or the
// plane is on the boundary of the grid) in every direction
inline bool in_row_box(const Block& b, int r, int s, int t) {
    const int l[3] = {r, s, t};
    for (int d = 0; d < b.dim; ++d) {
        const int64_t gl = (int64_t)l[d] + (int64_t)b.off[d] * b.M[d];   // global plane index
        const bool lo_ok = gl == 0 || l[d] - 1 >= -b.dn[d];
        const bool hi_ok = gl == b.P[d] - 1 || l[d] <= b.M[d] + b.up[d] - 1;
        if (!lo_ok || !hi_ok || l[d] < -b.dn[d] || l[d] > b.M[d] + b.up[d]) return false;
    }
    return true;
}

// local repeated id of lattice point (r,s,t): the reference's (M+1)^dim block first in its own order
// (r fastest), then the extra points of the first plane above the block, then (mode 2) the outer layer,
// each class in extended-lattice order.
struct Numbering {
    const Block& b;
    std::vector<int32_t> ext_id;  // only for points outside the own lattice
    int64_t n_own_lattice;
    int64_t n_first = 0;          // extra points of the first class
    explicit Numbering(const Block& blk) : b(blk) {
        n_own_lattice = (int64_t)b.n1[0] * b.n1[1] * b.n1[2];
        ext_id.assign((size_t)b.ne[0] * b.ne[1] * b.ne[2], -1);
        int32_t next = (int32_t)n_own_lattice;
        for (int cls = 0; cls < 2; ++cls) {
            for (int t = -b.dn[2]; t < b.n1[2] + b.up[2]; ++t)
                for (int s = -b.dn[1]; s < b.n1[1] + b.up[1]; ++s)
                    for (int r = -b.dn[0]; r < b.n1[0] + b.up[0]; ++r) {
                        if (in_own_lattice(b, r, s, t)) continue;
                        if ((cls == 0) != in_row_box(b, r, s, t)) continue;
                        ext_id[slot(r, s, t)] = next++;
                    }
            if (cls == 0) n_first = next - n_own_lattice;
        }
        n_rep = next;
    }
    int64_t n_rep = 0;
    size_t slot(int r, int s, int t) const {
        return ((size_t)(t + b.dn[2]) * b.ne[1] + (size_t)(s + b.dn[1])) * b.ne[0] + (size_t)(r + b.dn[0]);
    }
    int32_t id(int r, int s, int t) const {
        if (in_own_lattice(b, r, s, t)) return (int32_t)(r + (int64_t)b.n1[0] * (s + (int64_t)b.n1[1] * t));
        return ext_id[slot(r, s, t)];
    }
};

// Kuhn split of a cell into 6 tets, corner offsets in the reference's order
// (MeshStructured_def.hpp:772-801), and the 2 triangles of a 2D cell (:431-446).
const int KUHN[6][4][3] = {
    {{1, 0, 0}, {0, 0, 0}, {1, 0, 1}, {1, 1, 1}}, {{0, 0, 1}, {0, 0, 0}, {1, 0, 1}, {1, 1, 1}},
    {{1, 0, 0}, {0, 0, 0}, {1, 1, 0}, {1, 1, 1}}, {{0, 0, 0}, {0, 1, 0}, {1, 1, 0}, {1, 1, 1}},
    {{0, 0, 0}, {0, 1, 0}, {0, 1, 1}, {1, 1, 1}}, {{0, 0, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 1}}};
const int TRIS[2][3][2] = {{{1, 0}, {0, 0}, {1, 1}}, {{0, 1}, {0, 0}, {1, 1}}};

inline bool owned_pt(const Block& b, int r, int s, int t) {
    return r >= b.lo_owned[0] && r < b.n1[0] && s >= b.lo_owned[1] && s < b.n1[1] &&
           t >= b.lo_owned[2] && t < b.n1[2];
}

// visit every element: own cells first in the reference order, then the ghost cells that touch an owned
// point (mode 1) or a point of the row box (mode 2)
template <class F>
void for_each_element(const Block& b, F&& f) {
    const int dim = b.dim;
    const int nsub = dim == 3 ? 6 : 2;
    for (int pass = 0; pass < 2; ++pass) {
        const int t0 = pass ? -b.dn[2] : 0, t1 = dim == 3 ? b.M[2] + (pass ? b.up[2] : 0) : 1;
        const int s0 = pass ? -b.dn[1] : 0, s1 = b.M[1] + (pass ? b.up[1] : 0);
        const int r0 = pass ? -b.dn[0] : 0, r1 = b.M[0] + (pass ? b.up[0] : 0);
        for (int t = t0; t < t1; ++t)
            for (int s = s0; s < s1; ++s)
                for (int r = r0; r < r1; ++r) {
                    const bool ghost_cell = r < 0 || s < 0 || t < 0 || r >= b.M[0] || s >= b.M[1] || (dim == 3 && t >= b.M[2]);
                    if ((pass == 0) == ghost_cell) continue;
                    for (int k = 0; k < nsub; ++k) {
                        int pr[4], ps[4], pt[4];
                        bool touches = false;
                        for (int v = 0; v <= dim; ++v) {
                            if (dim == 3) {
                                pr[v] = r + KUHN[k][v][0];
                                ps[v] = s + KUHN[k][v][1];
                                pt[v] = t + KUHN[k][v][2];
                            } else {
                                pr[v] = r + TRIS[k][v][0];
                                ps[v] = s + TRIS[k][v][1];
                                pt[v] = 0;
                            }
                            touches = touches || (b.mode >= 2 ? in_row_box(b, pr[v], ps[v], pt[v])
                                                              : owned_pt(b, pr[v], ps[v], pt[v]));
                        }
                        if (ghost_cell && !touches) continue;
                        f(pr, ps, pt, ghost_cell);
                    }
                }
    }
}

int32_t flags_option1(int dim, const double* p, int32_t f, const double* o, const double* sz) {
    const double tol = 1.0e-12;  // MeshStructured_def.hpp:2976
    const double x = p[0], y = p[1];
    if (dim == 2) {  // :2984-2998, later tests override earlier ones
        if (x > o[0] - tol && y < o[1] + tol) f = 1;
        if (x > o[0] - tol && y > o[1] + sz[1] - tol) f = 1;
        if (x > o[0] + sz[0] - tol && y > o[1] + tol && y < o[1] + sz[1] - tol) f = 3;
        if (x < o[0] + tol) f = 2;
        return f;
    }
    const double z = p[2];  // :3136-3167
    if (x < o[0] + tol) f = 2;
    const bool in = x > o[0] + tol;
    if (in && z < o[2] + tol) f = 1;
    if (in && z > o[2] + sz[2] - tol) f = 1;
    if (in && y < o[1] + tol) f = 1;
    if (in && y > o[1] + sz[1] - tol) f = 1;
    if (x > o[0] + sz[0] - tol && y > o[1] + tol && y < o[1] + sz[1] - tol && z > o[2] + tol &&
        z < o[2] + sz[2] - tol)
        f = 3;
    return f;
}

}  // namespace

extern "C" int fedd_mesh_structured_sizes(int dim, const int* decomp, const int* cells, int rank,
                                          int with_ghost_elements, int64_t* n_elem, int64_t* n_rep,
                                          int64_t* n_uni, int64_t* n_global) {
    FEDD_CHECK(dim == 2 || dim == 3, "structured mesh: dimension must be 2 or 3");
    Block b;
    FEDD_TRY(make_block(dim, decomp, cells, rank, with_ghost_elements, b));
    int64_t ne = 0;
    for_each_element(b, [&](const int*, const int*, const int*, bool) { ++ne; });
    int64_t nrep = (int64_t)b.n1[0] * b.n1[1] * b.n1[2];
    int64_t ext = (int64_t)b.ne[0] * b.ne[1] * b.ne[2] - nrep;
    int64_t nuni = 1;
    for (int d = 0; d < dim; ++d) nuni *= b.n1[d] - b.lo_owned[d];
    if (n_elem) *n_elem = ne;
    if (n_rep) *n_rep = nrep + ext;
    if (n_uni) *n_uni = nuni;
    if (n_global) *n_global = b.P[0] * b.P[1] * b.P[2];
    return 0;
}

extern "C" int fedd_mesh_structured_build(int dim, const int* decomp, const int* cells, int rank,
                                          const double* origin, const double* size, int flags_option,
                                          int with_ghost_elements, int32_t* conn, double* xyz,
                                          int64_t* gid_rep, int32_t* flag_rep, int64_t* gid_uni,
                                          int32_t* flag_uni) {
    FEDD_CHECK(dim == 2 || dim == 3, "structured mesh: dimension must be 2 or 3");
    FEDD_CHECK(flags_option == 0 || flags_option == 1, "structured mesh: flags option %d not supported", flags_option);
    Block b;
    FEDD_TRY(make_block(dim, decomp, cells, rank, with_ghost_elements, b));
    Numbering num(b);
    double o[3] = {0, 0, 0}, sz[3] = {1, 1, 1}, h[3] = {0, 0, 0}, H[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
        if (origin) o[d] = origin[d];
        if (size) sz[d] = size[d];
        h[d] = sz[d] / (b.M[d] * b.N[d]);  // :646 h = length/(M*N)
        H[d] = sz[d] / b.N[d];             // :647
    }
    const double eps = std::numeric_limits<double>::epsilon();
    const double snap = dim == 3 ? eps : 100 * eps;  // :727-734 vs :371-374
    int64_t nu = 0;
    for (int t = -b.dn[2]; t < b.n1[2] + b.up[2]; ++t)
        for (int s = -b.dn[1]; s < b.n1[1] + b.up[1]; ++s)
            for (int r = -b.dn[0]; r < b.n1[0] + b.up[0]; ++r) {
                const int32_t id = num.id(r, s, t);
                const int l[3] = {r, s, t};
                double p[3] = {0, 0, 0};
                int64_t g[3] = {0, 0, 0};
                for (int d = 0; d < dim; ++d) {
                    // a point beyond the own lattice gets the coordinate its owning block computes
                    int ll = l[d], oo = b.off[d];
                    while (ll >= b.n1[d]) {
                        ll -= b.M[d];
                        oo += 1;
                    }
                    while (ll < 0) {
                        ll += b.M[d];
                        oo -= 1;
                    }
                    double c = ll * h[d] + oo * H[d];
                    if (c < snap && c > -snap) c = 0.0;
                    p[d] = c;
                    g[d] = (int64_t)l[d] + (int64_t)b.off[d] * b.M[d];
                    xyz[(int64_t)id * dim + d] = c;
                }
                gid_rep[id] = g[0] + b.P[0] * (g[1] + b.P[1] * g[2]);  // :736-737
                int32_t f = 0;
                for (int d = 0; d < dim; ++d)  // :739-744
                    if (p[d] > o[d] + sz[d] - snap || p[d] < o[d] + snap) f = 1;
                if (flags_option == 1) f = flags_option1(dim, p, f, o, sz);
                if (flag_rep) flag_rep[id] = f;
            }
    // unique list: repeated (own-lattice) order filtered by ownership (Map_def.hpp:201-206)
    for (int t = 0; t < b.n1[2]; ++t)
        for (int s = 0; s < b.n1[1]; ++s)
            for (int r = 0; r < b.n1[0]; ++r) {
                if (!owned_pt(b, r, s, t)) continue;
                const int32_t id = num.id(r, s, t);
                gid_uni[nu] = gid_rep[id];
                if (flag_uni) {
                    int32_t f = 0;
                    double p[3] = {0, 0, 0};
                    for (int d = 0; d < dim; ++d) p[d] = xyz[(int64_t)id * dim + d];
                    for (int d = 0; d < dim; ++d)
                        if (p[d] > o[d] + sz[d] - snap || p[d] < o[d] + snap) f = 1;
                    if (flags_option == 1) f = flags_option1(dim, p, f, o, sz);
                    flag_uni[nu] = f;
                }
                ++nu;
            }
    int64_t e = 0;
    const int nen = dim + 1;
    for_each_element(b, [&](const int* pr, const int* ps, const int* pt, bool) {
        for (int v = 0; v < nen; ++v) conn[e * nen + v] = num.id(pr[v], ps[v], pt[v]);
        ++e;
    });
    return 0;
}

/* L >= 2 ghost layers: the repeated-map nodes whose matrix rows are complete on this rank although another
 * rank owns them (the ghost nodes within L - 1 layers of the owned ones): count, or with arrays their
 * global ids and boundary flags -- the arguments of fedd_mesh_set_rows. */
extern "C" int fedd_mesh_structured_row_ghosts(int dim, const int* decomp, const int* cells, int rank,
                                               int with_ghost_elements, const double* origin, const double* size,
                                               int flags_option, int64_t* n_out, int64_t* gid, int32_t* flag) {
    FEDD_CHECK(dim == 2 || dim == 3, "structured mesh: dimension must be 2 or 3");
    FEDD_CHECK(with_ghost_elements >= 2, "structured mesh: row ghosts need at least 2 ghost layers");
    Block b;
    FEDD_TRY(make_block(dim, decomp, cells, rank, with_ghost_elements, b));
    double o[3] = {0, 0, 0}, sz[3] = {1, 1, 1}, h[3] = {0, 0, 0}, H[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
        if (origin) o[d] = origin[d];
        if (size) sz[d] = size[d];
        h[d] = sz[d] / (b.M[d] * b.N[d]);
        H[d] = sz[d] / b.N[d];
    }
    const double eps = std::numeric_limits<double>::epsilon();
    const double snap = dim == 3 ? eps : 100 * eps;
    int64_t n = 0;
    for (int t = -b.dn[2]; t < b.n1[2] + b.up[2]; ++t)
        for (int s = -b.dn[1]; s < b.n1[1] + b.up[1]; ++s)
            for (int r = -b.dn[0]; r < b.n1[0] + b.up[0]; ++r) {
                if (owned_pt(b, r, s, t) || !in_row_box(b, r, s, t)) continue;
                if (gid || flag) {
                    const int l[3] = {r, s, t};
                    double p[3] = {0, 0, 0};
                    int64_t g[3] = {0, 0, 0};
                    for (int d = 0; d < dim; ++d) {
                        int ll = l[d], oo = b.off[d];
                        while (ll >= b.n1[d]) {
                            ll -= b.M[d];
                            oo += 1;
                        }
                        while (ll < 0) {
                            ll += b.M[d];
                            oo -= 1;
                        }
                        double c = ll * h[d] + oo * H[d];
                        if (c < snap && c > -snap) c = 0.0;
                        p[d] = c;
                        g[d] = (int64_t)l[d] + (int64_t)b.off[d] * b.M[d];
                    }
                    if (gid) gid[n] = g[0] + b.P[0] * (g[1] + b.P[1] * g[2]);
                    if (flag) {
                        int32_t f = 0;
                        for (int d = 0; d < dim; ++d)
                            if (p[d] > o[d] + sz[d] - snap || p[d] < o[d] + snap) f = 1;
                        if (flags_option == 1) f = flags_option1(dim, p, f, o, sz);
                        flag[n] = f;
                    }
                }
                ++n;
            }
    if (n_out) *n_out = n;
    return 0;
}
