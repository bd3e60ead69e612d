// Host-side unstructured mesh input of the product path: INRIA/medit ".mesh" reader and the
// P2-from-P1 construction (edge mid-points), one rank.
//
// Behaviour follows
//   meshReadSize / meshReadData / readEntity   feddlib/core/Mesh/MeshFileReader.cpp:16-106, MeshFileReader.hpp:35-127
//     (keywords Dimension / Vertices / Edges / Triangles / Tetrahedra, count on the next line, vertex
//      lines always carry 3 coordinates, last token = flag, connectivity 1-based)
//   MeshUnstructured::readElements (1-based -> 0-based)        feddlib/core/Mesh/MeshUnstructured_def.hpp:1181-1190
//   MeshPartitioner::readAndPartitionMesh, 1-rank branch       feddlib/core/Mesh/MeshPartitioner_def.hpp:321-397
//   buildEdgeListParallel / EdgeElements sort+unique            feddlib/core/Mesh/MeshPartitioner_def.hpp:660-734, feddlib/core/FE/EdgeElements.cpp:105-155
//   MeshUnstructured::buildP2ofP1MeshEdge                       feddlib/core/Mesh/MeshUnstructured_def.hpp:129-410
//   determinePositionInElementP2 / determineFlagP2              feddlib/core/Mesh/MeshUnstructured_def.hpp:730-775, 806-900
// Layout: flat SoA arrays, no per-element objects.
#include "fedd_internal.hpp"
#include <algorithm>
#include <fstream>
#include <map>
#include <sstream>

namespace {

struct MeshFile {
    int dim = 0;
    std::vector<double> xyz;          // [nv*dim]
    std::vector<int32_t> vflag;       // [nv]
    std::vector<int32_t> elem, eflag; // [ne*(dim+1)], [ne]
    std::vector<int32_t> surf, sflag; // [ns*dim], [ns]   (2D: Edges, 3D: Triangles)
};

int read_file(const char* path, int dim, MeshFile& m) {
    std::ifstream in(path);
    if (!in) {
        fedd::set_error("cannot open mesh file %s", path);
        return 1;
    }
    m.dim = dim;
    const std::string key_elem = dim == 2 ? "Triangles" : "Tetrahedra";
    const std::string key_surf = dim == 2 ? "Edges" : "Triangles";
    std::string tok;
    while (in >> tok) {
        if (tok == "Vertices") {
            int64_t n;
            in >> n;
            m.xyz.resize((size_t)n * dim);
            m.vflag.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                double c[3];
                double f;
                in >> c[0] >> c[1] >> c[2] >> f;  // always 3 coordinates, then the flag
                for (int d = 0; d < dim; ++d) m.xyz[(size_t)i * dim + d] = c[d];
                m.vflag[(size_t)i] = (int32_t)f;
            }
        } else if (tok == key_elem || tok == key_surf || (dim == 3 && tok == "Edges")) {
            const bool is_elem = tok == key_elem, is_surf = tok == key_surf;
            const int nn = is_elem ? dim + 1 : (is_surf ? dim : 2);
            int64_t n;
            in >> n;
            std::vector<int32_t> tmp((size_t)n * nn), fl((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                for (int k = 0; k < nn; ++k) {
                    int64_t v;
                    in >> v;
                    tmp[(size_t)i * nn + k] = (int32_t)(v - 1);  // 1-based in the file
                }
                int64_t f;
                in >> f;
                fl[(size_t)i] = (int32_t)f;
            }
            if (is_elem) {
                m.elem.swap(tmp);
                m.eflag.swap(fl);
            } else if (is_surf) {
                m.surf.swap(tmp);
                m.sflag.swap(fl);
            }  // line segments of a 3D file are read past (not used by the hot path)
        }
        if (!in) {
            fedd::set_error("mesh file %s: parse error near keyword '%s'", path, tok.c_str());
            return 1;
        }
    }
    if (m.xyz.empty() || m.elem.empty()) {
        fedd::set_error("mesh file %s: no %s or no Vertices section for dimension %d", path, key_elem.c_str(), dim);
        return 1;
    }
    const int64_t nv = (int64_t)m.vflag.size();
    for (int32_t v : m.elem)
        if (v < 0 || v >= nv) {
            fedd::set_error("mesh file %s: element node id out of range", path);
            return 1;
        }
    return 0;
}

// local edge -> slot of the mid node in the P2 element
// 3D: (0,1)->4 (1,2)->5 (0,2)->6 (0,3)->7 (1,3)->8 (2,3)->9   2D: (0,1)->3 (1,2)->4 (0,2)->5
const int EDGE3[6][3] = {{0, 1, 4}, {1, 2, 5}, {0, 2, 6}, {0, 3, 7}, {1, 3, 8}, {2, 3, 9}};
const int EDGE2[3][3] = {{0, 1, 3}, {1, 2, 4}, {0, 2, 5}};

typedef std::pair<int32_t, int32_t> Edge;

void collect_edges(int dim, int64_t ne, const int32_t* conn, std::vector<Edge>& edges) {
    const int nen = dim + 1, nle = dim == 3 ? 6 : 3;
    edges.clear();
    edges.reserve((size_t)ne * nle);
    for (int64_t e = 0; e < ne; ++e)
        for (int k = 0; k < nle; ++k) {
            const int a = dim == 3 ? EDGE3[k][0] : EDGE2[k][0], b = dim == 3 ? EDGE3[k][1] : EDGE2[k][1];
            const int32_t u = conn[e * nen + a], v = conn[e * nen + b];
            edges.emplace_back(std::min(u, v), std::max(u, v));
        }
    std::sort(edges.begin(), edges.end());  // lexicographic (min, max): rank = global edge id
    edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
}

}  // namespace

extern "C" int fedd_mesh_read_sizes(const char* path, int dim, int64_t* n_vert, int64_t* n_elem, int64_t* n_surf) {
    FEDD_CHECK(path && (dim == 2 || dim == 3), "fedd_mesh_read_sizes: bad arguments");
    MeshFile m;
    FEDD_TRY(read_file(path, dim, m));
    if (n_vert) *n_vert = (int64_t)m.vflag.size();
    if (n_elem) *n_elem = (int64_t)m.eflag.size();
    if (n_surf) *n_surf = (int64_t)m.sflag.size();
    return 0;
}

extern "C" int fedd_mesh_read(const char* path, int dim, double* xyz, int32_t* vflag, int32_t* conn, int32_t* eflag,
                              int32_t* surf, int32_t* sflag) {
    FEDD_CHECK(path && (dim == 2 || dim == 3), "fedd_mesh_read: bad arguments");
    MeshFile m;
    FEDD_TRY(read_file(path, dim, m));
    if (xyz) std::copy(m.xyz.begin(), m.xyz.end(), xyz);
    if (vflag) std::copy(m.vflag.begin(), m.vflag.end(), vflag);
    if (conn) std::copy(m.elem.begin(), m.elem.end(), conn);
    if (eflag) std::copy(m.eflag.begin(), m.eflag.end(), eflag);
    if (surf) std::copy(m.surf.begin(), m.surf.end(), surf);
    if (sflag) std::copy(m.sflag.begin(), m.sflag.end(), sflag);
    return 0;
}

extern "C" int fedd_mesh_p2_sizes(int dim, int64_t n_elem, const int32_t* conn_p1, int64_t* n_edges) {
    FEDD_CHECK((dim == 2 || dim == 3) && conn_p1 && n_edges, "fedd_mesh_p2_sizes: bad arguments");
    std::vector<Edge> edges;
    collect_edges(dim, n_elem, conn_p1, edges);
    *n_edges = (int64_t)edges.size();
    return 0;
}

// P1 -> P2: node ids 0..n_vert-1 stay, mid node of edge k (k = rank in the sorted unique edge list)
// gets id n_vert + k (= P1Offset + edge id, MeshUnstructured_def.hpp:141,372-374), coordinate = the
// mid-point (:173-174), flag = volume_id if an end node is interior, else the lowest flag of the
// boundary entities (surf) containing both end nodes, volume_id if there is none (:806-900).
extern "C" int fedd_mesh_p2_build(int dim, int64_t n_vert, int64_t n_elem, const int32_t* conn_p1, const double* xyz_p1,
                                  const int32_t* vflag_p1, int64_t n_surf, const int32_t* surf, const int32_t* sflag,
                                  int volume_id, int32_t* conn_p2, double* xyz_p2, int32_t* flag_p2) {
    FEDD_CHECK((dim == 2 || dim == 3) && conn_p1 && xyz_p1 && vflag_p1 && conn_p2 && xyz_p2 && flag_p2,
               "fedd_mesh_p2_build: bad arguments");
    std::vector<Edge> edges;
    collect_edges(dim, n_elem, conn_p1, edges);
    const int nen1 = dim + 1, nen2 = dim == 3 ? 10 : 6, nle = dim == 3 ? 6 : 3;
    std::map<Edge, int32_t> surf_flag;  // node pair on the boundary -> lowest flag of the entities holding it
    for (int64_t s = 0; s < n_surf; ++s)
        for (int a = 0; a < dim; ++a)
            for (int b = a + 1; b < dim; ++b) {
                const int32_t u = surf[s * dim + a], v = surf[s * dim + b];
                const Edge key(std::min(u, v), std::max(u, v));
                auto it = surf_flag.find(key);
                if (it == surf_flag.end()) surf_flag.emplace(key, sflag[s]);
                else it->second = std::min(it->second, sflag[s]);
            }
    for (int64_t i = 0; i < n_vert; ++i) {
        for (int d = 0; d < dim; ++d) xyz_p2[i * dim + d] = xyz_p1[i * dim + d];
        flag_p2[i] = vflag_p1[i];
    }
    for (size_t k = 0; k < edges.size(); ++k) {
        const int32_t u = edges[k].first, v = edges[k].second;
        for (int d = 0; d < dim; ++d)
            xyz_p2[(n_vert + (int64_t)k) * dim + d] = (xyz_p1[(int64_t)u * dim + d] + xyz_p1[(int64_t)v * dim + d]) / 2.;
        int32_t f = volume_id;
        if (vflag_p1[u] != volume_id && vflag_p1[v] != volume_id) {
            auto it = surf_flag.find(edges[k]);
            if (it != surf_flag.end()) f = it->second;
        }
        flag_p2[n_vert + (int64_t)k] = f;
    }
    for (int64_t e = 0; e < n_elem; ++e) {
        for (int v = 0; v < nen1; ++v) conn_p2[e * nen2 + v] = conn_p1[e * nen1 + v];
        for (int k = 0; k < nle; ++k) {
            const int a = dim == 3 ? EDGE3[k][0] : EDGE2[k][0], b = dim == 3 ? EDGE3[k][1] : EDGE2[k][1];
            const int slot = dim == 3 ? EDGE3[k][2] : EDGE2[k][2];
            const int32_t u = conn_p1[e * nen1 + a], v = conn_p1[e * nen1 + b];
            const Edge key(std::min(u, v), std::max(u, v));
            const auto it = std::lower_bound(edges.begin(), edges.end(), key);
            conn_p2[e * nen2 + slot] = (int32_t)(n_vert + (it - edges.begin()));
        }
    }
    return 0;
}
