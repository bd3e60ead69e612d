// Element partitioner for unstructured meshes, host side.
//
// Replaces MeshPartitioner::readAndPartitionMesh + the map building behind it
// (feddlib/core/Mesh/MeshPartitioner_def.hpp:224-530: METIS_PartMeshDual on the dual graph, options :258-267, call :324;
// repeated map = nodes of the rank's elements :358-397; unique map by Map::buildUniqueMap, Map_def.hpp:184-210).
// METIS is not available (and its partitions are not reproducible across versions anyway): the elements are split by a
// balanced recursive coordinate bisection of their centroids instead (same kind of result: every element on exactly
// one rank, compact parts of equal size), which is deterministic and needs nothing but the coordinates.
// On top of the reference's maps this produces what the device path wants from a rank's mesh:
//   * L >= 1 layers of ghost elements around the owned nodes, so that the owned rows (L >= 1) and the rows of the
//     ghost nodes within L - 1 layers (the "row ghosts", L >= 2) are complete without a matrix exchange;
//   * the owner of every repeated node (lowest rank among the ranks whose OWN elements hold it), for the halo plan.
#include "fedd_internal.hpp"
#include <algorithm>
#include <numeric>
#include <unordered_map>
#include <vector>

using namespace fedd;

namespace {

// idx[lo, hi) -> parts [p0, p0 + np): split at the count proportional to floor(np / 2) along the longest edge of the
// bounding box of the centroids; ties by element id
void bisect(const std::vector<double>& cen, int dim, std::vector<int64_t>& idx, int64_t lo, int64_t hi, int p0, int np,
            int32_t* part) {
    if (np <= 1) {
        for (int64_t k = lo; k < hi; ++k) part[idx[(size_t)k]] = p0;
        return;
    }
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t k = lo; k < hi; ++k)
        for (int d = 0; d < dim; ++d) {
            const double v = cen[(size_t)idx[(size_t)k] * dim + d];
            mn[d] = std::min(mn[d], v);
            mx[d] = std::max(mx[d], v);
        }
    int ax = 0;
    for (int d = 1; d < dim; ++d)
        if (mx[d] - mn[d] > mx[ax] - mn[ax]) ax = d;
    const int npl = np / 2;
    const int64_t mid = lo + (hi - lo) * npl / np;
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int64_t a, int64_t b) {
        const double va = cen[(size_t)a * dim + ax], vb = cen[(size_t)b * dim + ax];
        return va < vb || (va == vb && a < b);
    });
    bisect(cen, dim, idx, lo, mid, p0, npl, part);
    bisect(cen, dim, idx, mid, hi, p0 + npl, np - npl, part);
}

struct Extract {
    std::vector<int64_t> elems;          // global element ids of the rank's mesh: own first (ascending), then ghost layers
    std::vector<int64_t> rep;            // global node ids, ascending
    std::vector<int64_t> uni;            // owned node ids, ascending
    std::vector<int64_t> rowg;           // row ghosts, ascending
    std::vector<int32_t> owner_rep;      // owner rank of every repeated node
};

int extract(int nen, int64_t n_elem, const int32_t* conn, int64_t n_node, const int32_t* elem_part, int nparts, int rank,
            int layers, Extract& out) {
    // owner of every node: lowest rank among the parts of its elements
    std::vector<int32_t> owner((size_t)n_node, INT32_MAX);
    for (int64_t e = 0; e < n_elem; ++e) {
        const int32_t p = elem_part[e];
        FEDD_CHECK(p >= 0 && p < nparts, "mesh partition: element %lld has part %d", (long long)e, p);
        for (int j = 0; j < nen; ++j) {
            const int32_t nd = conn[e * nen + j];
            FEDD_CHECK(nd >= 0 && nd < n_node, "mesh partition: element %lld refers to node %d", (long long)e, nd);
            owner[(size_t)nd] = std::min(owner[(size_t)nd], p);
        }
    }
    // node -> elements
    std::vector<int64_t> nptr((size_t)n_node + 1, 0);
    for (int64_t i = 0; i < n_elem * nen; ++i) ++nptr[(size_t)conn[i] + 1];
    for (int64_t i = 0; i < n_node; ++i) nptr[(size_t)i + 1] += nptr[(size_t)i];
    std::vector<int64_t> nel((size_t)(n_elem * nen)), cur(nptr.begin(), nptr.end() - 1);
    for (int64_t e = 0; e < n_elem; ++e)
        for (int j = 0; j < nen; ++j) nel[(size_t)cur[(size_t)conn[e * nen + j]]++] = e;
    std::vector<char> in_mesh((size_t)n_elem, 0), node_in((size_t)n_node, 0);
    out.elems.clear();
    for (int64_t e = 0; e < n_elem; ++e)
        if (elem_part[e] == rank) {
            in_mesh[(size_t)e] = 1;
            out.elems.push_back(e);
        }
    // layer 1 grows from the OWNED nodes (an owned node may lie on elements of higher ranks only partly held here);
    // layer k > 1 from all nodes of the mesh so far
    std::vector<int64_t> front;
    for (int64_t nd = 0; nd < n_node; ++nd)
        if (owner[(size_t)nd] == rank) front.push_back(nd);
    out.uni = front;
    for (int layer = 0; layer < layers; ++layer) {
        std::vector<int64_t> added;
        for (int64_t nd : front)
            for (int64_t p = nptr[(size_t)nd]; p < nptr[(size_t)nd + 1]; ++p) {
                const int64_t e = nel[(size_t)p];
                if (!in_mesh[(size_t)e]) {
                    in_mesh[(size_t)e] = 1;
                    added.push_back(e);
                }
            }
        std::sort(added.begin(), added.end());
        out.elems.insert(out.elems.end(), added.begin(), added.end());
        // next front: every node of the mesh so far
        front.clear();
        std::fill(node_in.begin(), node_in.end(), 0);
        for (int64_t e : out.elems)
            for (int j = 0; j < nen; ++j) {
                const int32_t nd = conn[e * nen + j];
                if (!node_in[(size_t)nd]) {
                    node_in[(size_t)nd] = 1;
                    front.push_back(nd);
                }
            }
    }
    std::fill(node_in.begin(), node_in.end(), 0);
    out.rep.clear();
    for (int64_t e : out.elems)
        for (int j = 0; j < nen; ++j) {
            const int32_t nd = conn[e * nen + j];
            if (!node_in[(size_t)nd]) {
                node_in[(size_t)nd] = 1;
                out.rep.push_back(nd);
            }
        }
    std::sort(out.rep.begin(), out.rep.end());
    out.owner_rep.resize(out.rep.size());
    for (size_t k = 0; k < out.rep.size(); ++k) out.owner_rep[k] = owner[(size_t)out.rep[k]];
    // row ghosts: ghost nodes all of whose elements are in the mesh (complete rows); only with L >= 2
    out.rowg.clear();
    if (layers >= 2)
        for (int64_t nd : out.rep) {
            if (owner[(size_t)nd] == rank) continue;
            bool complete = true;
            for (int64_t p = nptr[(size_t)nd]; p < nptr[(size_t)nd + 1] && complete; ++p) complete = in_mesh[(size_t)nel[(size_t)p]] != 0;
            if (complete) out.rowg.push_back(nd);
        }
    return 0;
}

}  // namespace

extern "C" int fedd_mesh_partition(int dim, int nen, int64_t n_elem, const int32_t* conn, int64_t n_node, const double* xyz,
                                   int nparts, int32_t* elem_part) {
    FEDD_CHECK(dim == 2 || dim == 3, "fedd_mesh_partition: dimension must be 2 or 3");
    FEDD_CHECK(nen >= dim + 1 && n_elem >= 0 && n_node >= 0 && nparts >= 1, "fedd_mesh_partition: bad sizes");
    FEDD_CHECK(conn && xyz && elem_part, "fedd_mesh_partition: null array");
    std::vector<double> cen((size_t)n_elem * dim, 0.0);
    for (int64_t e = 0; e < n_elem; ++e)
        for (int v = 0; v <= dim; ++v) {
            const int32_t nd = conn[e * nen + v];
            FEDD_CHECK(nd >= 0 && nd < n_node, "fedd_mesh_partition: element %lld refers to node %d", (long long)e, nd);
            for (int d = 0; d < dim; ++d) cen[(size_t)e * dim + d] += xyz[(size_t)nd * dim + d] / (dim + 1);
        }
    std::vector<int64_t> idx((size_t)n_elem);
    std::iota(idx.begin(), idx.end(), 0);
    bisect(cen, dim, idx, 0, n_elem, 0, nparts, elem_part);
    return 0;
}

extern "C" int fedd_mesh_partition_sizes(int nen, int64_t n_elem, const int32_t* conn, int64_t n_node, const int32_t* elem_part,
                                         int nparts, int rank, int ghost_layers, int64_t* n_elem_loc, int64_t* n_rep,
                                         int64_t* n_uni, int64_t* n_row_ghosts) {
    FEDD_CHECK(conn && elem_part && rank >= 0 && rank < nparts && ghost_layers >= 0 && ghost_layers <= 8,
               "fedd_mesh_partition_sizes: bad arguments");
    Extract x;
    FEDD_TRY(extract(nen, n_elem, conn, n_node, elem_part, nparts, rank, ghost_layers, x));
    if (n_elem_loc) *n_elem_loc = (int64_t)x.elems.size();
    if (n_rep) *n_rep = (int64_t)x.rep.size();
    if (n_uni) *n_uni = (int64_t)x.uni.size();
    if (n_row_ghosts) *n_row_ghosts = (int64_t)x.rowg.size();
    return 0;
}

extern "C" int fedd_mesh_partition_extract(int dim, int nen, int64_t n_elem, const int32_t* conn, int64_t n_node,
                                           const double* xyz, const int32_t* flag, const int32_t* elem_part, int nparts,
                                           int rank, int ghost_layers, int32_t* conn_loc, double* xyz_loc, int64_t* gid_rep,
                                           int32_t* flag_rep, int32_t* owner_rep, int64_t* gid_uni, int32_t* flag_uni,
                                           int64_t* row_ghost_gid, int32_t* row_ghost_flag, int64_t* elem_gid) {
    FEDD_CHECK(conn && xyz && elem_part && rank >= 0 && rank < nparts && ghost_layers >= 0 && ghost_layers <= 8,
               "fedd_mesh_partition_extract: bad arguments");
    Extract x;
    FEDD_TRY(extract(nen, n_elem, conn, n_node, elem_part, nparts, rank, ghost_layers, x));
    std::unordered_map<int64_t, int32_t> loc;
    loc.reserve(x.rep.size() * 2);
    for (size_t k = 0; k < x.rep.size(); ++k) {
        loc.emplace(x.rep[k], (int32_t)k);
        if (gid_rep) gid_rep[k] = x.rep[k];
        if (flag_rep) flag_rep[k] = flag ? flag[(size_t)x.rep[k]] : 0;
        if (owner_rep) owner_rep[k] = x.owner_rep[k];
        if (xyz_loc)
            for (int d = 0; d < dim; ++d) xyz_loc[k * dim + d] = xyz[(size_t)x.rep[k] * dim + d];
    }
    for (size_t k = 0; k < x.elems.size(); ++k) {
        if (elem_gid) elem_gid[k] = x.elems[k];
        if (conn_loc)
            for (int j = 0; j < nen; ++j) conn_loc[k * nen + j] = loc[conn[x.elems[k] * nen + j]];
    }
    for (size_t k = 0; k < x.uni.size(); ++k) {
        if (gid_uni) gid_uni[k] = x.uni[k];
        if (flag_uni) flag_uni[k] = flag ? flag[(size_t)x.uni[k]] : 0;
    }
    for (size_t k = 0; k < x.rowg.size(); ++k) {
        if (row_ghost_gid) row_ghost_gid[k] = x.rowg[k];
        if (row_ghost_flag) row_ghost_flag[k] = flag ? flag[(size_t)x.rowg[k]] : 0;
    }
    return 0;
}
