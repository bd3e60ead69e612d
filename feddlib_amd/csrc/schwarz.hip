// One-level overlapping additive Schwarz, batched: many small subdomains per GPU, exact dense
// local solves.
//
// Stands in for what the reference obtains from Trilinos through
//   Stratimikos::enableFROSch + Thyra::initializePrec   feddlib/problems/Solver/Preconditioner_def.hpp:43, 243-463
// with the options of feddlib/problems/tests/laplace/parametersPrec.xml:10-61 and
// feddlib/problems/tests/steadyLinElas_Perf/parametersPrec.xml:7-25 (AlgebraicOverlappingOperator,
// Overlap 1, CrsGraph layers, Restricted/Averaging/Full combine, exact local solver).  FROSch
// itself is not in the reference tree; the definition implemented here is normative:
//   subdomain i  = owned nodes of box i of a regular grid over the rank's owned nodes
//                  + `overlap` layers of the (Dirichlet-modified) matrix graph,
//   A_i          = principal submatrix, inverted exactly (register-tiled dense Gauss-Jordan),
//   M^-1 r       = sum_i P_i A_i^-1 R_i r, P_i restricted / averaged / full prolongation.
// Apply = one workgroup per subdomain streaming its slab of A_i^-1 once from HBM (HBM-bound).
// With fedd_schwarz_setup(two_level = 1) the coarse level of coarse.hip is added to the result.
#include "fedd_internal.hpp"
#include <algorithm>
#include <climits>
#include <cmath>

namespace fedd {
namespace {

constexpr int NMAX = SCHWARZ_NMAX;
constexpr int HS = 2048;  // LDS hash set slots for the overlap layer

struct BinGeom {
    int dim;
    double lo[3], w[3];
    int g[3];
};

__global__ void k_minmax(const double* __restrict__ xyz, int32_t n, int dim, double* __restrict__ part) {
    __shared__ double smin[3][256], smax[3][256];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int d = 0; d < dim; ++d) {
            const double v = xyz[(int64_t)i * dim + d];
            mn[d] = fmin(mn[d], v);
            mx[d] = fmax(mx[d], v);
        }
    for (int d = 0; d < 3; ++d) {
        smin[d][threadIdx.x] = mn[d];
        smax[d][threadIdx.x] = mx[d];
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int d = 0; d < 3; ++d) {
                smin[d][threadIdx.x] = fmin(smin[d][threadIdx.x], smin[d][threadIdx.x + s]);
                smax[d][threadIdx.x] = fmax(smax[d][threadIdx.x], smax[d][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int d = 0; d < 3; ++d) {
            part[blockIdx.x * 6 + d] = smin[d][0];
            part[blockIdx.x * 6 + 3 + d] = smax[d][0];
        }
}

// consecutive lanes of a wave that carry the same key form a run (nodes are numbered along x: four to six in a row share a
// box); one atomic per run instead of one per lane.  Returns the lane that starts this lane's run and the run's length.
__device__ __forceinline__ void wave_run(int32_t key, bool valid, int& start_lane, int& run_len) {
    const int lane = threadIdx.x & 63;
    const int32_t prev = __shfl_up(key, 1, 64);
    const bool pvalid = __shfl_up(valid ? 1 : 0, 1, 64) != 0;
    const uint64_t starts = __ballot(valid && (lane == 0 || !pvalid || key != prev));
    const uint64_t vmask = __ballot(valid);
    const uint64_t upto = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    start_lane = upto ? 63 - __builtin_clzll(upto) : lane;
    // the run ends before the next start or the first invalid lane after start_lane
    const uint64_t stop = (starts | ~vmask) & ((start_lane == 63) ? 0ull : (~0ull << (start_lane + 1)));
    run_len = stop ? __builtin_ctzll(stop) - start_lane : 64 - start_lane;
}

// box of every owned DOF = box of the node that carries it (dof / dofs, or the dof -> node map of a
// merged block system)
// n dofs get a box, the first n_count of them (the owned ones) are counted: a box exists where it holds an
// owned dof; the others are row-ghost dofs, which join the boxes of their position as foreign members
__global__ void k_bin_id(const double* __restrict__ xyz, int32_t n, int32_t n_count, int dofs,
                         const int32_t* __restrict__ dof_node, BinGeom gm, int32_t* __restrict__ raw, int32_t* cnt) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = r < n;
    const int32_t i = in ? (dof_node ? dof_node[r] : r / dofs) : 0;
    int32_t b = 0, mul = 1;
    for (int d = 0; d < gm.dim; ++d) {
        // a node on a box boundary (within 1e-9 box widths) goes to the box on the centre side: with the odd
        // number of boxes per direction the partition of a point set that is symmetric about the centre of
        // its bounding box is then symmetric too
        const double t = (xyz[(int64_t)i * gm.dim + d] - gm.lo[d]) / gm.w[d];
        const double kb = floor(t + 0.5);
        int ix = (int)floor(t);
        if (fabs(t - kb) <= 1e-9) ix = 2 * (int)kb <= gm.g[d] - 1 ? (int)kb : (int)kb - 1;
        ix = min(gm.g[d] - 1, max(0, ix));
        b += mul * ix;
        mul *= gm.g[d];
    }
    if (in) raw[r] = b;
    // (one atomic per run of equal boxes among consecutive lanes)
    int sl, len;
    wave_run(b, r < n_count, sl, len);
    if (r < n_count && (int)(threadIdx.x & 63) == sl) atomicAdd(&cnt[b], len);
}

// row-ghost dofs r in [n0, n1) whose box exists on this rank: count per box / fill the per-box lists
// (cid = exclusive scan of the non-empty flags: a box exists iff cid[raw + 1] > cid[raw])
__global__ void k_foreign_count(const int32_t* __restrict__ raw, const int32_t* __restrict__ cid, int32_t n0, int32_t n1,
                                int32_t* cnt) {
    const int32_t r = n0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n1) return;
    const int32_t b = raw[r];
    if (cid[b + 1] > cid[b]) atomicAdd(&cnt[cid[b]], 1);
}

__global__ void k_foreign_fill(const int32_t* __restrict__ raw, const int32_t* __restrict__ cid, int32_t n0, int32_t n1,
                               int32_t* cursor, int32_t* __restrict__ nodes) {
    const int32_t r = n0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n1) return;
    const int32_t b = raw[r];
    if (cid[b + 1] > cid[b]) nodes[atomicAdd(&cursor[cid[b]], 1)] = r;
}

__global__ void k_flag_nonempty(const int32_t* __restrict__ cnt, int32_t n, int32_t* __restrict__ flag) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = cnt[i] > 0 ? 1 : 0;
}

__global__ void k_compact_counts(const int32_t* __restrict__ cnt, const int32_t* __restrict__ cid, int32_t n,
                                 int32_t* __restrict__ out) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cnt[i] > 0) out[cid[i]] = cnt[i];
}

__global__ void k_fill_bins(const int32_t* __restrict__ raw, const int32_t* __restrict__ cid, int32_t n,
                            int32_t* cursor, int32_t* __restrict__ node_bin, int32_t* __restrict__ bin_nodes) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const int32_t c = in ? cid[raw[i]] : -1;
    if (in) node_bin[i] = c;
    // one atomic per run of equal boxes among consecutive lanes (k_sort_bins orders every box afterwards)
    int sl, len;
    wave_run(c, in, sl, len);
    const int lane = threadIdx.x & 63;
    int32_t base = 0;
    if (in && lane == sl) base = atomicAdd(&cursor[c], len);
    base = __shfl(base, sl, 64);
    if (in) bin_nodes[base + (lane - sl)] = i;
}

// one wave per bin: rank sort through LDS (the dofs of a bin are distinct; a lane per bin sorting in global memory by
// insertion was 1.0 ms for 166 375 bins of 64)
__global__ __launch_bounds__(64) void k_sort_bins(const int32_t* __restrict__ ptr, int32_t nb, int32_t* nodes) {
    __shared__ int32_t sh[SCHWARZ_NMAX];
    const int32_t r = blockIdx.x;
    const int lane = threadIdx.x;
    const int32_t b = ptr[r], e = ptr[r + 1];
    const int n = e - b;
    if (n > SCHWARZ_NMAX) {     // (a bin the dense solver will refuse anyway: keep it correct)
        if (lane == 0)
            for (int32_t i = b + 1; i < e; ++i) {
                const int32_t v = nodes[i];
                int32_t j = i - 1;
                while (j >= b && nodes[j] > v) {
                    nodes[j + 1] = nodes[j];
                    --j;
                }
                nodes[j + 1] = v;
            }
        return;
    }
    for (int i = lane; i < n; i += 64) sh[i] = nodes[b + i];
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const int32_t v = sh[i];
        int rank = 0;
        for (int k = 0; k < n; ++k) rank += sh[k] < v ? 1 : 0;
        nodes[b + rank] = v;
    }
}

// One workgroup per bin: owned dofs first (sorted), then the dofs reached by `overlap` graph
// layers (sorted).  Layers grow through stored rows: the owned ones and those of the row ghosts
// (n_stored >= n_rows; without row ghosts a ghost row is not stored on this rank).
template <int NM, int HSZ>
__global__ __launch_bounds__(256) void k_sub_dofs(const int32_t* __restrict__ bin_ptr,
                                                  const int32_t* __restrict__ bin_nodes,
                                                  const int32_t* __restrict__ node_bin,
                                                  const int32_t* __restrict__ fbin_ptr,
                                                  const int32_t* __restrict__ fbin_nodes,
                                                  const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ colind, int32_t n_rows, int32_t n_stored,
                                                  int ghost_overlap,
                                                  int overlap, int32_t* __restrict__ sub_n,
                                                  int32_t* __restrict__ sub_nown, int32_t* __restrict__ sub_dofs) {
    __shared__ int32_t tab[HSZ];
    __shared__ int32_t lst[HSZ];
    __shared__ int32_t s_cnt, s_prev, s_bad;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t nb = bin_ptr[b];
    const int n_own = bin_ptr[b + 1] - nb;   // the box lists dofs
    const int32_t fb = fbin_ptr ? fbin_ptr[b] : 0;
    const int n_for = fbin_ptr ? fbin_ptr[b + 1] - fb : 0;   // dofs of other ranks' nodes in this box (rows stored here)
    int32_t* out = sub_dofs + (int64_t)b * NM;
    for (int k = tid; k < n_own && k < NM; k += 256) out[k] = bin_nodes[nb + k];
    auto insert = [&](int32_t col) {
        uint32_t h = ((uint32_t)col * 2654435761u) % HSZ;
        for (int probe = 0; probe < HSZ; ++probe) {
            const int32_t old = atomicCAS(&tab[h], -1, col);
            if (old == -1 || old == col) break;
            h = (h + 1) % HSZ;
        }
    };
    // First with the foreign members: the subdomain is then the WHOLE box plus its overlap, the same on every
    // rank that holds a part of the box (each keeps the rows of its own nodes).  That needs a stored row for
    // every dof of the subdomain; if the ghost layers of the mesh do not reach that far, second attempt with
    // the owned part of the box alone (the rank boundary cuts the box).
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int nf = attempt == 0 ? n_for : 0;
        for (int k = tid; k < HSZ; k += 256) tab[k] = -1;
        if (tid == 0) {
            s_cnt = 0;
            s_prev = 0;
            s_bad = 0;
        }
        __syncthreads();
        for (int k = tid; k < nf; k += 256) insert(fbin_nodes[fb + k]);
        __syncthreads();
        for (int layer = 0; layer < overlap; ++layer) {
            // sources: the box (layer 0) or everything collected so far (later layers)
            const int nsrc = layer == 0 ? n_own + nf : s_prev;
            // 16 lanes per source row: consecutive lanes read consecutive entries of its CSR row
            for (int k = tid >> 4; k < nsrc; k += 16) {
                const int32_t src = layer == 0 ? (k < n_own ? bin_nodes[nb + k] : fbin_nodes[fb + k - n_own]) : lst[k];
                if (src >= n_stored) continue;
                const int32_t p_end = rowptr[src + 1];
                for (int32_t p = rowptr[src] + (tid & 15); p < p_end; p += 16) {
                    const int32_t col = colind[p];
                    if (col < n_rows && node_bin[col] == b) continue;
                    if (!ghost_overlap && col >= n_rows) continue;  // another rank's dof: its row is not stored here
                    // With row ghosts only they are imported by the halo exchange: a plain ghost beyond them has
                    // neither a stored row nor a defined entry in the work vectors, so it never joins a subdomain
                    // (overlap deeper than the mesh's row-ghost layers).  The whole-box attempt is given up instead.
                    if (n_stored > n_rows && col >= n_stored) {
                        if (nf > 0) s_bad = 1;
                        continue;
                    }
                    insert(col);
                }
            }
            __syncthreads();
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            for (int k = tid; k < HSZ; k += 256)
                if (tab[k] >= 0) lst[atomicAdd(&s_cnt, 1)] = tab[k];
            __syncthreads();
            if (tid == 0) s_prev = s_cnt;
            __syncthreads();
        }
        if (overlap == 0 && nf > 0) {   // no layer loop ran: list the foreign members
            for (int k = tid; k < HSZ; k += 256)
                if (tab[k] >= 0) lst[atomicAdd(&s_cnt, 1)] = tab[k];
            __syncthreads();
            if (tid == 0) s_prev = s_cnt;
            __syncthreads();
        }
        if (nf == 0) break;
        for (int k = tid; k < s_prev; k += 256)
            if (lst[k] >= n_stored) s_bad = 1;
        if (tid == 0 && n_own + s_prev > NM) s_bad = 1;   // the whole box would not fit the dense solver: cut it
        __syncthreads();
        if (!s_bad) break;
        __syncthreads();
    }
    const int n_ext = s_prev;
    // rank sort of the (distinct) collected dofs
    for (int k = tid; k < n_ext; k += 256) {
        const int32_t v = lst[k];
        int rank = 0;
        for (int m = 0; m < n_ext; ++m) rank += lst[m] < v ? 1 : 0;
        if (n_own + rank < NM) out[n_own + rank] = v;
    }
    // the tail of the list is a valid dof id too: the apply kernel reads all NM entries
    for (int k = n_own + n_ext + tid; k < NM; k += 256) out[k] = 0;
    if (tid == 0) {
        sub_n[b] = n_own + n_ext;
        sub_nown[b] = n_own;
    }
}

__device__ __forceinline__ int bsearch_i32(const int32_t* a, int n, int32_t v);

// ---- subdomains with the same local matrix share one slab ------------------------------------------------------------
// On a mesh with repeated cells (the structured cube of the headline: 389 017 subdomains, a few hundred distinct local
// matrices) most principal submatrices A_i are copies of one another up to rounding.  Each subdomain's matrix is
// fingerprinted -- entries quantised to 2^-44 of their row's largest magnitude, the rows' maxima quantised against the
// subdomain's largest one and that one with a 40-bit mantissa (the absolute scale: A and 2 A are two matrices), 128 bits
// of order-independent hash over (local row, local column, quantised value) plus the sizes --, the subdomain of lowest index with a given fingerprint is
// its REPRESENTATIVE, only representatives are inverted and stored, every other subdomain's slab pointer refers to its
// representative's slab.  Equal fingerprints = equal matrices to 6e-14 relative per entry (a collision of two
// independent 64-bit hashes aside), i.e. inverses equal to ~1e-12: far inside the 1e-10 parity bar; a fingerprint that
// differs through rounding only costs a shared slab, never correctness.  The apply then works from a few slabs that sit in
// L2 instead of streaming 8 GB (k_apply_mfma below): the one-level step at 214^3 cells drops from 564 to 341 ms.
__global__ void k_row_absmax(const int32_t* __restrict__ rowptr, const double* __restrict__ val, int32_t n_rows,
                             double* __restrict__ rmax) {
    // 16 lanes per row: consecutive lanes read consecutive entries (a lane per row made every load touch 64 lines)
    const int32_t r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int e = threadIdx.x & 15;
    double m = 0.0;
    if (r < n_rows) {
        const int32_t p_end = rowptr[r + 1];
        for (int32_t p = rowptr[r] + e; p < p_end; p += 16) m = fmax(m, fabs(val[p]));
    }
    for (int off = 8; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 16));
    if (r < n_rows && e == 0) rmax[r] = m;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// Row maxima and ROW hashes in one coalesced pass over the stored rows (16 lanes per row): 128 bits, order-independent, over
// (column - row, value quantised to 2^-44 of the row's largest magnitude) of every entry that does not quantise to zero.
// A subdomain's fingerprint is then built from the hashes of its rows (k_sub_fingerprint_rows below): 31 M row hashes to
// gather at the 214^3 grid instead of 460 M matrix entries to look up (3.4 ms -> well under 1 ms).
// (four rows per 16-lane group: their bounds, then the first sixteen entries of each -- values and columns -- requested together and
// kept in registers for both passes; longer rows loop over the rest)
constexpr int RH_R = 4;
__global__ void k_row_hash(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const double* __restrict__ val,
                           int32_t n_rows, double* __restrict__ rmax, uint64_t* __restrict__ rh) {
    const int32_t g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int e = threadIdx.x & 15;
    const int32_t r0 = g * RH_R;
    int32_t pb[RH_R + 1];
#pragma unroll
    for (int q = 0; q <= RH_R; ++q) pb[q] = rowptr[min(r0 + q, n_rows)];
    double v0[RH_R];
    int32_t c0[RH_R];
#pragma unroll
    for (int q = 0; q < RH_R; ++q) {
        const int32_t idx = pb[q] + e;
        const bool has = r0 + q < n_rows && idx < pb[q + 1];
        v0[q] = has ? val[idx] : 0.0;
        c0[q] = has ? colind[idx] : 0;
    }
#pragma unroll
    for (int q = 0; q < RH_R; ++q) {
        const int32_t r = r0 + q;
        const bool live = r < n_rows;
        const int32_t p0 = pb[q], p_end = live ? pb[q + 1] : pb[q];
        double m = fabs(v0[q]);
        for (int32_t p = p0 + 16 + e; p < p_end; p += 16) m = fmax(m, fabs(val[p]));
        for (int off = 8; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 16));
        const double scale = m > 0.0 ? 17592186044416.0 / m : 0.0;   // 2^44 / row max
        uint64_t h1 = 0, h2 = 0;
        auto add = [&](double a, int32_t cidx) {
            const int64_t qv = (int64_t)llrint(a * scale);
            if (qv == 0) return;
            const uint64_t key = (uint64_t)(uint32_t)(cidx - r);
            h1 += mix64((key + 0x2545f4914f6cdd1dull) * 0x9e3779b97f4a7c15ull + (uint64_t)qv);
            h2 += mix64((key + 0x632be59bd9b4e019ull) * 0xd6e8feb86659fd93ull ^ ((uint64_t)qv * 0xa0761d6478bd642full));
        };
        if (p0 + e < p_end) add(v0[q], c0[q]);
        for (int32_t p = p0 + 16 + e; p < p_end; p += 16) add(val[p], colind[p]);
        for (int off = 8; off > 0; off >>= 1) {
            h1 += __shfl_xor(h1, off, 16);
            h2 += __shfl_xor(h2, off, 16);
        }
        if (live && e == 0) {
            rmax[r] = m;
            rh[2 * (int64_t)r] = h1;
            rh[2 * (int64_t)r + 1] = h2;
        }
    }
}

// Fingerprint of every subdomain from the hashes of its rows: one wave per subdomain, a lane per local row.  Equal fingerprints
// = the same number of dofs at the same positions relative to the first one, and in every local row the same entries at the
// same column offsets (entries that leave the subdomain included: stricter than the local matrix needs, which costs
// nothing on a mesh of repeated cells) = equal local matrices.  The scale is part of the identity (see below).
template <int NM>
__global__ __launch_bounds__(64) void k_sub_fingerprint_rows(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_nown,
                                                             const int32_t* __restrict__ sub_dofs, const double* __restrict__ rmax,
                                                             const uint64_t* __restrict__ rh, int32_t n_stored, int32_t p_off,
                                                             uint64_t* __restrict__ fp) {
    constexpr int T = NM / 64;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = sub_n[b], no = sub_nown[b];
    int32_t g[T];
    double rm[T];
    double smax = 0.0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int i = lane + 64 * t;
        g[t] = i < n ? sub_dofs[(int64_t)b * NM + i] : -1;
        rm[t] = (g[t] >= 0 && g[t] < n_stored) ? rmax[g[t]] : 0.0;
        smax = fmax(smax, rm[t]);
    }
    for (int off = 32; off > 0; off >>= 1) smax = fmax(smax, __shfl_xor(smax, off, 64));
    const int32_t s0 = __shfl(g[0], 0, 64);
    const double sscale = smax > 0.0 ? 17592186044416.0 / smax : 0.0;
    uint64_t h1 = 0, h2 = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int i = lane + 64 * t;
        if (g[t] < 0) continue;
        if (g[t] >= n_stored) {   // ghost row without a stored row: identity
            h1 += mix64(((uint64_t)i << 32) ^ 0x9e3779b97f4a7c15ull);
            h2 += mix64(((uint64_t)i << 20) ^ 0xd1b54a32d192ed03ull);
            continue;
        }
        // pressure rows of a merged system are pivoted after the velocities: part of the matrix' identity
        const uint64_t tag = g[t] >= p_off ? 0x5851f42d4c957f2dull : 0ull;
        const uint64_t rel = (uint64_t)(uint32_t)(g[t] - s0);
        const uint64_t qs = (uint64_t)llrint(rm[t] * sscale);
        const uint64_t a = rh[2 * (int64_t)g[t]], c2 = rh[2 * (int64_t)g[t] + 1];
        h1 += mix64((a ^ ((uint64_t)i << 48)) + mix64(rel * 0x9fb21c651e98df25ull + qs) + tag);
        h2 += mix64((c2 + ((uint64_t)i << 40)) ^ mix64((rel + 0x94d049bb133111ebull) * 0xbf58476d1ce4e5b9ull ^ qs) ^ tag);
    }
    for (int off = 32; off > 0; off >>= 1) {
        h1 += __shfl_down(h1, off, 64);
        h2 += __shfl_down(h2, off, 64);
    }
    if (lane == 0) {
        const uint64_t sbits = ((uint64_t)__double_as_longlong(smax) + (1ull << 11)) & ~((1ull << 12) - 1ull);
        h1 = mix64(h1 + (uint64_t)n * 1000003ull + (uint64_t)no + mix64(sbits));
        h2 += mix64(sbits ^ 0xbf58476d1ce4e5b9ull);
        if (h1 == 0) h1 = 1;   // 0 marks an empty table slot
        fp[2 * (int64_t)b] = h1;
        fp[2 * (int64_t)b + 1] = h2;
    }
}

// (the entry-by-entry form of the fingerprint, kept for option "schwarz_fp_kind" 1: hashes only the entries inside the subdomain)
// one wave per subdomain: 16 lanes per local row (four rows at a time), the lanes of a group read consecutive entries of the
// CSR row (with a lane per row every load instruction touched 64 different cache lines: address-bound, 10.7 ms at cfg 3)
template <int NM>
__global__ __launch_bounds__(64) void k_sub_fingerprint(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_nown,
                                                        const int32_t* __restrict__ sub_dofs, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colind, const double* __restrict__ val,
                                                        const double* __restrict__ rmax, int32_t n_stored, int32_t p_off,
                                                        uint64_t* __restrict__ fp) {
    // global dof -> local index through a small open-addressing table in LDS (one or two probes per matrix entry; the two
    // binary searches of the first version -- 14 dependent LDS reads per entry -- were most of the kernel's 10.7 ms at cfg 3)
    constexpr int HS = 2 * NM;
    __shared__ int32_t sdof[NM];
    __shared__ int32_t hkey[HS], hval[HS];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = sub_n[b], no = sub_nown[b];
    for (int k = lane; k < HS; k += 64) hkey[k] = -1;
    for (int k = lane; k < n; k += 64) sdof[k] = sub_dofs[(int64_t)b * NM + k];
    __syncthreads();
    for (int k = lane; k < n; k += 64) {
        const int32_t g = sdof[k];
        unsigned slot = ((unsigned)g * 2654435761u) >> (32 - 9);
        static_assert(HS == 512, "hash shift assumes 512 slots");
        while (atomicCAS(&hkey[slot], -1, g) != -1) slot = (slot + 1) & (HS - 1);   // (dofs of a subdomain are distinct)
        hval[slot] = k;
    }
    __syncthreads();
    uint64_t h1 = 0, h2 = 0;
    const int grp = lane >> 4, e = lane & 15;
    // The entries of a row are quantised against the row's own largest magnitude, which loses the row's absolute scale: two
    // local matrices A_j = alpha A_i (boxes in mesh regions of different h, a partly scaled matrix) would share a
    // fingerprint and an inverse that is wrong by 1 / alpha.  So the scale is part of the identity: every row's maximum
    // relative to the subdomain's largest one (quantised like the entries), and that largest one itself with its
    // mantissa rounded to 40 bits (equal matrices differ by rounding noise ~1e-15; a scale that differs by more than
    // ~1e-12 is a different matrix).
    double smax = 0.0;
    for (int i = lane; i < n; i += 64) {
        const int32_t g = sdof[i];
        if (g < n_stored) smax = fmax(smax, rmax[g]);
    }
    for (int off = 32; off > 0; off >>= 1) smax = fmax(smax, __shfl_xor(smax, off, 64));
    const double sscale = smax > 0.0 ? 17592186044416.0 / smax : 0.0;
    for (int i = grp; i < n; i += 4) {
        const int32_t g = sdof[i];
        if (g >= n_stored) {   // ghost row without a stored row: identity
            if (e == 0) {
                h1 += mix64(((uint64_t)i << 32) ^ 0x9e3779b97f4a7c15ull);
                h2 += mix64(((uint64_t)i << 20) ^ 0xd1b54a32d192ed03ull);
            }
            continue;
        }
        if (e == 0) {
            const uint64_t qs = (uint64_t)llrint(rmax[g] * sscale);
            h1 += mix64((((uint64_t)i << 48) ^ qs) * 0x9fb21c651e98df25ull + 0x2545f4914f6cdd1dull);
            h2 += mix64((((uint64_t)i << 48) + qs) ^ 0x94d049bb133111ebull);
        }
        const double scale = rmax[g] > 0.0 ? 17592186044416.0 / rmax[g] : 0.0;   // 2^44 / row max
        // pressure rows of a merged system are pivoted after the velocities: part of the matrix' identity
        const uint64_t tag = g >= p_off ? 0x5851f42d4c957f2dull : 0ull;
        const int32_t p_end = rowptr[g + 1];
        for (int32_t p = rowptr[g] + e; p < p_end; p += 16) {
            const int32_t col = colind[p];
            int cidx = -1;
            for (unsigned slot = ((unsigned)col * 2654435761u) >> (32 - 9);; slot = (slot + 1) & (HS - 1)) {
                const int32_t kk = hkey[slot];
                if (kk == col) cidx = hval[slot];
                if (kk == col || kk == -1) break;
            }
            if (cidx < 0) continue;
            const int64_t q = (int64_t)llrint(val[p] * scale);
            if (q == 0) continue;
            const uint64_t key = ((uint64_t)i * NM + (uint64_t)cidx) ^ tag;
            h1 += mix64(key * 0x9e3779b97f4a7c15ull + (uint64_t)q);
            h2 += mix64((key + 0x632be59bd9b4e019ull) * 0xd6e8feb86659fd93ull ^ ((uint64_t)q * 0xa0761d6478bd642full));
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        h1 += __shfl_down(h1, off, 64);
        h2 += __shfl_down(h2, off, 64);
    }
    if (lane == 0) {
        const uint64_t sbits = ((uint64_t)__double_as_longlong(smax) + (1ull << 11)) & ~((1ull << 12) - 1ull);
        h1 = mix64(h1 + (uint64_t)n * 1000003ull + (uint64_t)no + mix64(sbits));
        h2 += mix64(sbits ^ 0xbf58476d1ce4e5b9ull);
        if (h1 == 0) h1 = 1;   // 0 marks an empty table slot
        fp[2 * (int64_t)b] = h1;
        fp[2 * (int64_t)b + 1] = h2;
    }
}

// open-addressing table keyed by the 128-bit fingerprint; the slot's value is the lowest subdomain index seen
__device__ __forceinline__ int32_t fp_insert_one(uint64_t h1, uint64_t h2, int32_t b, uint64_t* __restrict__ tkey,
                                                 int32_t* __restrict__ tmin, int64_t tsize) {
    int64_t s = (int64_t)(h1 % (uint64_t)tsize);
    for (int64_t probe = 0; probe < tsize; ++probe) {
        const unsigned long long old = atomicCAS((unsigned long long*)&tkey[2 * s], 0ull, (unsigned long long)h1);
        if (old == 0ull) {
            // this thread claimed the slot: publish the second word (readers wait for it)
            atomicExch((unsigned long long*)&tkey[2 * s + 1], (unsigned long long)(h2 | 1ull));
        }
        if (old == 0ull || old == h1) {
            unsigned long long second;
            do {
                second = atomicAdd((unsigned long long*)&tkey[2 * s + 1], 0ull);
            } while (second == 0ull);
            if (second == (h2 | 1ull)) {
                atomicMin(&tmin[s], b);
                return (int32_t)s;
            }
        }
        s = s + 1 == tsize ? 0 : s + 1;
    }
    return -1;   // table full (cannot happen: tsize >= 2 nsub): own representative
}

// The lanes of a wave that carry the same fingerprint (on a structured mesh: nearly all 64) send ONE of them to the
// table -- the lowest lane = the lowest subdomain index of the group: a few hundred thousand atomics on a few dozen
// addresses serialise in L2 otherwise (5.8 ms at 389 017 subdomains, more than the inversions that are left).
__global__ void k_fp_insert(const uint64_t* __restrict__ fp, int32_t nsub, uint64_t* __restrict__ tkey, int32_t* __restrict__ tmin,
                            int64_t tsize, int32_t* __restrict__ slot_of) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool active = b < nsub;
    const unsigned long long h1 = active ? fp[2 * (int64_t)b] : 0ull, h2 = active ? fp[2 * (int64_t)b + 1] : 0ull;
    uint64_t todo = __ballot(active);
    int32_t my_slot = -1;
    while (todo) {      // (uniform over the wave)
        const int leader = __builtin_ctzll(todo);
        const unsigned long long l1 = __shfl(h1, leader, 64), l2 = __shfl(h2, leader, 64);
        const uint64_t same = __ballot(active && h1 == l1 && h2 == l2) & todo;
        int32_t s_found = -1;
        if (lane == leader) s_found = fp_insert_one(h1, h2, b, tkey, tmin, tsize);
        s_found = __shfl(s_found, leader, 64);
        if ((same >> lane) & 1ull) my_slot = s_found;
        todo &= ~same;
    }
    if (active) slot_of[b] = my_slot;
}

__global__ void k_fp_resolve(const int32_t* __restrict__ slot_of, const int32_t* __restrict__ tmin, const int32_t* __restrict__ sub_n,
                             int32_t nsub, int32_t* __restrict__ rep, int32_t* __restrict__ sub_n_inv, int32_t* __restrict__ n_rep) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsub) return;
    int32_t r = slot_of[b] >= 0 ? tmin[slot_of[b]] : b;
    if (sub_n[r] != sub_n[b]) r = b;          // (a colliding fingerprint of a different size is never shared)
    rep[b] = r;
    sub_n_inv[b] = r == b ? sub_n[b] : 0;     // the inversion kernels skip subdomains of size 0
    if (r == b) atomicAdd(n_rep, 1);
}

__global__ void k_iota(int32_t* __restrict__ v, int32_t n) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// subdomains that hold a dof of another rank (index >= n_owned in the column space) wait for the ghost import of r; the
// others can be applied while it is in flight (option "halo_overlap"): sort key = representative, +2^30 for the former
__global__ void k_order_key(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_dofs, const int32_t* __restrict__ rep,
                            int32_t nsub, int32_t n_owned, int split, int32_t* __restrict__ key, int32_t* __restrict__ n_int) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsub) return;
    int32_t k = rep ? rep[b] : b;
    if (split) {
        bool ghost = false;
        const int n = sub_n[b];
        for (int i = 0; i < n; ++i) ghost |= sub_dofs[(int64_t)b * NMAX + i] >= n_owned;
        if (ghost) k += 1 << 30;
        else atomicAdd(n_int, 1);
    }
    key[b] = k;
}

// record of a place of the apply order: (subdomain, representative, columns | owned rows << 10 | conforming << 20, first dof).
// conforming = the dof list is the representative's list shifted (dof_k - dof_0 equal for all k): on a structured mesh
// every box of a class; the apply then computes the ids from the representative's offsets instead of reading the list.
// 16 lanes per place.
__global__ void k_pack_order(const int32_t* __restrict__ ord, const int32_t* __restrict__ rep, const int32_t* __restrict__ sub_n,
                             const int32_t* __restrict__ sub_nown, const int32_t* __restrict__ sub_dofs, int32_t n,
                             int4* __restrict__ rec, int32_t* __restrict__ n_conf) {
    const int32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int e = threadIdx.x & 15;
    bool same = true;
    int32_t sidx = 0, rp = 0, nn = 0, a0 = 0;
    if (i < n) {
        sidx = ord[i];
        rp = rep ? rep[sidx] : sidx;
        nn = sub_n[sidx];
        const int32_t* __restrict__ mine = sub_dofs + (int64_t)sidx * NMAX;
        const int32_t* __restrict__ ref = sub_dofs + (int64_t)rp * NMAX;
        a0 = mine[0];
        const int32_t r0 = ref[0];
        same = sub_n[rp] == nn;
        if (same) {     // (differences OR-ed, no short-circuit: the loads of all entries fly together)
            int32_t diff = 0;
            for (int k = e; k < nn; k += 16) diff |= (mine[k] - a0) ^ (ref[k] - r0);
            same = diff == 0;
        }
    }
    unsigned bad = same ? 0u : 1u;
    for (int off = 8; off > 0; off >>= 1) bad |= __shfl_xor(bad, off, 16);
    if (i < n && e == 0) rec[i] = make_int4(sidx, rp, nn | (sub_nown[sidx] << 10) | (bad ? 0 : 1 << 20), a0);
    const uint64_t conf = __ballot(i < n && e == 0 && !bad);
    if (conf && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(conf)) atomicAdd(n_conf, (int32_t)__builtin_popcountll(conf));
}

// the representatives (any order: every one is inverted by its own workgroup)
__global__ void k_rep_list(const int32_t* __restrict__ rep, int32_t nsub, int32_t* __restrict__ list, int32_t* __restrict__ cnt) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nsub && rep[b] == b) list[atomicAdd(cnt, 1)] = b;
}

__global__ void k_fp_share(const int32_t* __restrict__ rep, int32_t nsub, int64_t* __restrict__ inv_ptr) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsub) return;
    const int32_t r = rep[b];
    if (r != b) inv_ptr[b] = inv_ptr[r];      // (a representative's own entry is never rewritten)
}

// leading dimension of a slab = number of stored rows: the apply kernel's lanes (row, column group)
// then walk the slab as one contiguous stream without padding bytes
__device__ __forceinline__ int slab_ld(int v) { return v; }

__global__ void k_slab_sizes(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_nown, int32_t nb,
                             int restricted, int64_t* __restrict__ sz) {
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const int n = sub_n[b];
    const int no = restricted ? sub_nown[b] : n;
    const int64_t s = (int64_t)n * slab_ld(no);
    sz[b] = (s + 15) & ~(int64_t)15;
}

__device__ __forceinline__ int bsearch_i32(const int32_t* a, int n, int32_t v) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t m = a[mid];
        if (m == v) return mid;
        if (m < v) lo = mid + 1;
        else hi = mid - 1;
    }
    return -1;
}

// Register-tiled Gauss-Jordan inversion of one subdomain matrix per workgroup, n <= 16*T.
// The 256 threads form a 16 x 16 grid; thread (ty, tx) keeps the cyclic sub-matrix
// A[ty + 16 a][tx + 16 b], a, b < T, in registers for the whole elimination (T*T f64 = 2 T^2 VGPRs).
// Per pivot k = 16 kb + kc: the owners of column k / row k publish them through a double-buffered
// LDS line (one barrier per pivot), every thread then does T*T FMAs.  The pivot's tile index kb
// is a compile-time constant (outer loop unrolled), so no register array is indexed dynamically.
// Rows/columns n..16T-1 are identity padding.  No pivoting: see k_invert's note.
// TA > 0 (restricted combine, plain system, owned dofs within the first TA tiles): only the owned
// rows of the inverse are needed, so the overlap dofs are pivoted first and a row that has been
// pivoted and is not an owned row is never read again: its tile drops out of the update at compile
// time (tiles below TA hold the owned rows and stay).  The kernel is bound by VALU issue, so the
// dropped FMAs are time: 36 % of them for a 27 + 74 dof box, 45 % for the 24 + 114 dof elasticity box.
template <int T, int TA>
__global__ __launch_bounds__(256, (T <= 7 ? 4 : 2)) void k_invert_reg(const int32_t* __restrict__ sub_n,
                                                       const int32_t* __restrict__ sub_nown,
                                                       const int32_t* __restrict__ sub_dofs,
                                                       const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ colind,
                                                       const double* __restrict__ val, int32_t n_rows,
                                                       int restricted, const int64_t* __restrict__ inv_ptr,
                                                       double* __restrict__ inv, int32_t* __restrict__ bad,
                                                       int n_lo, int n_hi, int32_t p_off, int own_le,
                                                       const int32_t* __restrict__ ids) {
    constexpr int NP = 16 * T;
    __shared__ int32_t sdof[NP];
    __shared__ double stage[16][NP + 1];
    __shared__ double colbuf[2][NP], rowbuf[2][NP];
    // ids: the subdomains to invert (the representatives, when equal local matrices share an inverse); else all
    const int b = ids ? ids[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
    const int n = sub_n[b];
    if (n <= n_lo || n > n_hi) return;  // another size class handles this subdomain
    const int no = sub_nown[b];
    if (TA > 0 ? no > 16 * TA : no <= own_le) return;  // the other variant of this size class handles it
    const int ty = tid & 15, tx = tid >> 4;
    for (int k = tid; k < NP; k += 256) sdof[k] = k < n ? sub_dofs[(int64_t)b * NMAX + k] : -1;
    __syncthreads();
    double A[T][T];
    // ---- dense extraction, 16 rows at a time through the LDS stage ----
    // Lane (r, l) = (tid >> 4, tid & 15) handles entry l (+16, +32, ...) of rows r + 16 a.  The global
    // loads of all T tiles are issued first (row bounds, then the first column id and value of each
    // row), so that the T dependent load chains overlap instead of running one tile after another.
    int32_t pb[T], pe[T], col0[T];
    double v0[T];
    {
        const int r = tid >> 4, l = tid & 15;
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const int i = r + 16 * a;
            const int32_t g = i < n ? sdof[i] : -1;
            const bool stored = g >= 0 && g < n_rows;
            pb[a] = stored ? rowptr[g] + l : 0;
            pe[a] = stored ? rowptr[g + 1] : 0;
        }
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const bool have = pb[a] < pe[a];
            col0[a] = have ? colind[pb[a]] : -1;
            v0[a] = have ? val[pb[a]] : 0.0;
        }
    }
#pragma unroll
    for (int a = 0; a < T; ++a) {
        for (int e = tid; e < 16 * (NP + 1); e += 256) (&stage[0][0])[e] = 0.0;
        __syncthreads();
        {
            const int r = tid >> 4, l = tid & 15;
            const int i = r + 16 * a;
            if (i < n) {
                const int32_t g = sdof[i];
                if (g < n_rows) {
                    int32_t col = col0[a];
                    double v = v0[a];
                    for (int32_t p = pb[a]; p < pe[a]; p += 16) {
                        if (p != pb[a]) {
                            col = colind[p];
                            v = val[p];
                        }
                        int cidx = bsearch_i32(sdof, no, col);
                        if (cidx < 0) {
                            cidx = bsearch_i32(sdof + no, n - no, col);
                            if (cidx >= 0) cidx += no;
                        }
                        if (cidx >= 0) stage[r][cidx] = v;
                    }
                } else if (l == 0) {
                    stage[r][i] = 1.0;  // ghost row (not stored on this rank): identity
                }
            } else if (l == 0) {
                stage[r][i] = 1.0;  // padding
            }
        }
        __syncthreads();
#pragma unroll
        for (int bb = 0; bb < T; ++bb) A[a][bb] = stage[ty][tx + 16 * bb];
        __syncthreads();
    }
    // ---- elimination ----
    // Pivot order: for a merged saddle-point system (dofs >= p_off are pressures, zero diagonal)
    // all velocity pivots first, then the pressures (their Schur complement is definite), which is
    // safe without row exchanges; storage order is untouched (any symmetric pivot order gives the
    // same inverse).  Plain systems: one pass.
    bool singular = false;
    int step = 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npass = (TA > 0 || p_off != INT32_MAX) ? 2 : 1;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {  // unrolled: `pass` is a constant in the tile conditions below
    if (pass >= npass) break;
#pragma unroll
    for (int kb = 0; kb < T; ++kb) {
        for (int kc = 0; kc < 16; ++kc) {
            const int k = 16 * kb + kc;
            if (k >= n) break;
            if (TA > 0) {
                if ((k < no) != (pass == 1)) continue;  // overlap dofs first, owned dofs last
            } else if (npass == 2 && (sdof[k] >= p_off) != (pass == 1)) {
                continue;
            }
            const int buf = (step++) & 1;
            // tx == kc holds in one wave only (tx = tid >> 4): the other three skip by a scalar branch
            const bool col_wave = wave == (kc >> 2);
            if (col_wave) {
                if (tx == kc) {
#pragma unroll
                    for (int a = 0; a < T; ++a) colbuf[buf][ty + 16 * a] = A[a][kb];
                }
            }
            if (ty == kc) {
#pragma unroll
                for (int bb = 0; bb < T; ++bb) rowbuf[buf][tx + 16 * bb] = A[kb][bb];
            }
            __syncthreads();
            const double piv = rowbuf[buf][k];
            singular = singular || !(fabs(piv) > 1e-300);
            // reciprocal: hardware estimate (about 23 bits) + two Newton steps instead of the IEEE
            // division sequence
            double pinv = __builtin_amdgcn_rcp(piv);
            pinv = fma(fma(-piv, pinv, 1.0), pinv, pinv);
            pinv = fma(fma(-piv, pinv, 1.0), pinv, pinv);
            double cc[T];
#pragma unroll
            for (int a = 0; a < T; ++a) cc[a] = colbuf[buf][ty + 16 * a];
            const bool prow = ty == kc;
            const bool pcol = col_wave && tx == kc;
            if (prow) cc[kb] = -1.0;
#pragma unroll
            for (int bb = 0; bb < T; ++bb) {
                double r_b = rowbuf[buf][tx + 16 * bb] * pinv;
                if (bb == kb) r_b = pcol ? pinv : r_b;
#pragma unroll
                for (int a = 0; a < T; ++a) {
                    // finished overlap rows (pass 0: tiles TA .. kb-1) and all overlap rows once the
                    // owned dofs are pivoted (pass 1) are not needed any more
                    if (TA > 0 && a >= TA && (pass == 1 || a < kb)) continue;
                    double av = A[a][bb];
                    if (a == kb) av = prow ? 0.0 : av;
                    if (bb == kb) av = pcol ? 0.0 : av;
                    A[a][bb] = fma(-cc[a], r_b, av);
                }
            }
        }
    }
    }
    if (singular && tid == 0) bad[0] = 1;
    // ---- needed rows of the inverse -> column-major slab [c][rp] ----
    const int nrow = restricted ? no : n;
    const int rp = slab_ld(nrow);
    double* __restrict__ slab = inv + inv_ptr[b];
#pragma unroll
    for (int a = 0; a < T; ++a) {
        const int i = ty + 16 * a;
        if (i >= rp) continue;
#pragma unroll
        for (int bb = 0; bb < T; ++bb) {
            const int j = tx + 16 * bb;
            if (j < n) __builtin_nontemporal_store(i < nrow ? A[a][bb] : 0.0, slab + (int64_t)j * rp + i);
        }
    }
}

// (Subdomains beyond the register-tiled classes, 161 .. 256 dofs, are inverted by the batched blocked Gauss-Jordan of
// dense.hip through schwarz_dense_batched; the LDS / global-memory scalar sweep that used to take them ran at ~1 % of peak.)

// z (+)= P_i A_i^-1 R_i r for every subdomain; thread (r, s) accumulates row r over columns
// c == s (mod S), so the S*rp active lanes read the slab as one contiguous stream.
template <bool RESTRICTED>
__global__ __launch_bounds__(256) void k_apply(const int32_t* __restrict__ sub_n,
                                               const int32_t* __restrict__ sub_nown,
                                               const int32_t* __restrict__ sub_dofs,
                                               const int64_t* __restrict__ inv_ptr,
                                               const double* __restrict__ inv, const double* __restrict__ r,
                                               double* __restrict__ z, const int4* __restrict__ perm, int32_t p0) {
    __shared__ double rsub[NMAX];
    __shared__ double part[256];
    __shared__ int32_t sdof[NMAX];
    // perm: the workgroups walk places p0 ... of the setup's order records (interior-first split), else subdomain = workgroup
    const int b = perm ? perm[p0 + (int32_t)blockIdx.x].x : (int)blockIdx.x, tid = threadIdx.x;
    const int n = sub_n[b];
    const int nrow = RESTRICTED ? sub_nown[b] : n;
    const int rp = slab_ld(nrow);
    const int S = 256 / rp;
    for (int c = tid; c < n; c += 256) {
        const int32_t d = sub_dofs[(int64_t)b * NMAX + c];
        sdof[c] = d;
        rsub[c] = r[d];
    }
    __syncthreads();
    const int rr = tid % rp, s = tid / rp;
    double acc = 0.0;
    if (s < S) {
        const double* __restrict__ slab = inv + inv_ptr[b] + rr;
        int c = s;
        for (; c + 3 * S < n; c += 4 * S) {
            const double a0 = slab[(int64_t)c * rp];
            const double a1 = slab[(int64_t)(c + S) * rp];
            const double a2 = slab[(int64_t)(c + 2 * S) * rp];
            const double a3 = slab[(int64_t)(c + 3 * S) * rp];
            acc += a0 * rsub[c] + a1 * rsub[c + S] + a2 * rsub[c + 2 * S] + a3 * rsub[c + 3 * S];
        }
        for (; c < n; c += S) acc += slab[(int64_t)c * rp] * rsub[c];
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < nrow) {
        double sum = 0.0;
        for (int q = 0; q < S; ++q) sum += part[q * rp + tid];
        if (RESTRICTED) z[sdof[tid]] = sum;
        else atomicAdd(&z[sdof[tid]], sum);
    }
}

// Restricted apply, flat streaming: the slab [n][nrow] is one contiguous, 128-byte aligned run of
// n * nrow doubles; lane t loads elements 2t + 512 k with 16-byte loads (every wave request is a
// whole aligned kilobyte), multiplies by r[column] and parks the products in LDS; then lane
// (row, s) adds the products of its row over the columns c == s (mod S) and the S partial sums of
// a row are added in order.  prod[] is dynamic LDS (max n * nrow of the launch).
// Vector memory returns in issue order: the dof ids are requested first so that the dependent
// gather of r can be issued while the slab loads are still in flight; all loads are unconditional
// (clamped, always valid addresses) to keep the compiler's vmcnt bookkeeping exact.
// Measured on cfg 2 (801 MB of slabs): 159 us against 168 us for the strided kernel above; a
// persistent, software-pipelined variant of this kernel was slower (183 us) and is not kept.
constexpr int AP_BATCH = 6;  // 16-byte loads in flight per lane (6 * 512 elements cover 101 x 27)

// COMPACT (subdomains of at most 128 dofs): the gathered r and, after the products, the partial row sums share
// 152 doubles of static LDS, so that an interior 101 x 27 subdomain takes 21824 + 1216 = 23040 B and seven
// workgroups fit the 160 KB of a CU instead of six (the partial sums then use 152 / nrow lane groups instead of
// 256 / nrow).  Otherwise 256 + 256 doubles.
template <bool COMPACT>
__global__ __launch_bounds__(256) void k_apply_flat(const int32_t* __restrict__ sub_n,
                                                    const int32_t* __restrict__ sub_nown,
                                                    const int32_t* __restrict__ sub_dofs,
                                                    const int64_t* __restrict__ inv_ptr,
                                                    const double* __restrict__ inv, const double* __restrict__ r,
                                                    double* __restrict__ z, int span, const int4* __restrict__ perm, int32_t p0) {
    constexpr int PS = COMPACT ? 152 : 256;          // partial-sum slots
    __shared__ double shbuf[COMPACT ? 152 : NMAX + 256];
    double* const rsub = shbuf;
    double* const part = COMPACT ? shbuf : shbuf + NMAX;
    extern __shared__ double prod[];
    const int tid = threadIdx.x;
    // bijective XCD remap (workgroups i and i + 8 share an XCD and its L2): XCD k takes a contiguous
    // eighth of the subdomains, so the part of r that neighbouring subdomains gather stays in one L2
    const int nb_ = gridDim.x, q_ = nb_ >> 3, rem_ = nb_ & 7, xcd_ = blockIdx.x & 7, within_ = blockIdx.x >> 3;
    const int place = (xcd_ < rem_ ? xcd_ * (q_ + 1) : rem_ * (q_ + 1) + (xcd_ - rem_) * q_) + within_;
    const int b = perm ? perm[p0 + place].x : place;      // (perm: places of the setup's order records, interior-first split)
    const int n = sub_n[b], nrow = sub_nown[b];
    const int total = n * nrow;
    const double* __restrict__ slab = inv + inv_ptr[b];
    // entries past n hold dof 0 (k_sub_dofs); span = largest subdomain rounded up to whole waves: the waves
    // beyond it do not fetch their part of the 1 KB list (2.4 % of the apply's traffic at 101 dofs)
    const int32_t d = tid < span ? __builtin_nontemporal_load(sub_dofs + (int64_t)b * NMAX + tid) : 0;
    // slabs are padded to a multiple of 16 doubles: the 16-byte load of an odd tail stays inside
    const int last = ((total + 1) & ~1) - 2;
    double2 a[AP_BATCH];
#pragma unroll
    for (int k = 0; k < AP_BATCH; ++k) {
        // streamed once per application: non-temporal, so that 0.8 GB of slabs do not push the system
        // matrix and the vectors out of the Infinity Cache between two SpMVs
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(slab + min(2 * tid + 512 * k, last)));
        a[k].x = t.x;
        a[k].y = t.y;
    }
    const double rv = r[d];
    if (tid < n) rsub[tid] = rv;
    __syncthreads();
    const int dc = 512 / nrow, dr = 512 - dc * nrow;
    int c0 = (2 * tid) / nrow, r0 = 2 * tid - c0 * nrow;  // column / row of element f
#pragma unroll
    for (int k = 0; k < AP_BATCH; ++k) {
        const int f = 2 * tid + 512 * k;
        if (f < total) {
            const int c1 = r0 + 1 == nrow ? c0 + 1 : c0;
            prod[f] = a[k].x * rsub[c0];
            if (f + 1 < total) prod[f + 1] = a[k].y * rsub[c1];
        }
        c0 += dc;
        r0 += dr;
        if (r0 >= nrow) {
            r0 -= nrow;
            ++c0;
        }
    }
    for (int f = 2 * tid + 512 * AP_BATCH; f < total; f += 512) {  // larger slabs: the rest
        const double2 v = *reinterpret_cast<const double2*>(slab + f);
        const int c1 = r0 + 1 == nrow ? c0 + 1 : c0;
        prod[f] = v.x * rsub[c0];
        if (f + 1 < total) prod[f + 1] = v.y * rsub[c1];
        c0 += dc;
        r0 += dr;
        if (r0 >= nrow) {
            r0 -= nrow;
            ++c0;
        }
    }
    __syncthreads();   // all products parked; rsub is dead from here on (COMPACT: its space becomes `part`)
    const int S = PS / nrow;
    const int rr = tid % nrow, s = tid / nrow;
    if (s < S) {
        // four independent LDS reads per step (fixed association, so still reproducible)
        double acc = 0.0;
        const double* pr = prod + rr;
        int c = s;
        for (; c + 3 * S < n; c += 4 * S) {
            const double p0 = pr[c * nrow], p1 = pr[(c + S) * nrow], p2 = pr[(c + 2 * S) * nrow], p3 = pr[(c + 3 * S) * nrow];
            acc += (p0 + p1) + (p2 + p3);
        }
        for (; c < n; c += S) acc += pr[c * nrow];
        part[tid] = acc;
    }
    __syncthreads();
    if (tid < nrow) {   // lane tid < nrow loaded the dof id of owned row tid itself
        double sum = 0.0;
        for (int q = 0; q < S; ++q) sum += part[q * nrow + tid];
        z[d] = sum;
    }
}

// Restricted apply when most subdomains share their inverse with others (schwarz_dedupe on a mesh with repeated cells), on
// the f64 matrix cores.  The subdomains are walked in the order of `order` (sorted by representative at setup, lattice
// order within one), so that 16 consecutive ones nearly always share their inverse: such a batch is the product
//   Z[16 RT x 16] = Ainv[16 RT x n] R[n x 16]      (v_mfma_f64_16x16x4_f64: RT row tiles, n / 4 steps).
// The four waves split the steps (wave w takes steps w, w + 4, ...), keep their share of Ainv in registers as A fragments
// -- reloaded (from L2) only when the representative changes -- and gather their B fragments -- lane (k, j) = entry
// 4 step + k of subdomain j's restriction of r -- straight from global memory.  The dof lists of a batch are read as
// whole rows (16-byte loads, 16 lanes per subdomain) one batch ahead and handed to the fragment lanes through LDS; the
// records of `order` (subdomain, representative, sizes) are read 64 places at a time, one chunk ahead, and the batch
// boundaries inside a chunk come from bit scans of a ballot.  The four partial tiles are added through LDS in wave
// order: fixed summation order => reproducible; no atomics (every owned dof belongs to one subdomain).
// Without the sharing the slabs stream from HBM once per apply and the flat kernel above is the right one.
typedef double ap_d4 __attribute__((ext_vector_type(4)));
constexpr int AM_MB = 16;
template <int RT, int KW, int ABL = 0>   // owned rows <= 16 RT, columns <= 16 KW; ABL: ablation bits of tools/ab_apply.py (1 no gathers of r, 2 no stores of z, 4 no exchange of the partial tiles, 8 no products)
__global__ __launch_bounds__(256, (RT <= 2 && KW <= 10) ? 3 : ((RT <= 4 && KW <= 12) ? 2 : 1)) void k_apply_mfma(const int4* __restrict__ order, const int32_t* __restrict__ sub_dofs,
                                                    const int64_t* __restrict__ inv_ptr, const double* __restrict__ inv,
                                                    const double* __restrict__ r, double* __restrict__ z, int32_t nsub, int span) {
    constexpr int S = 16 * KW + 2;              // row stride of the id lists in LDS: fragment reads hit 32 distinct banks
    constexpr int U = (16 * KW + 63) / 64;      // 16-byte loads per lane and batch
    __shared__ double part[4][RT][4][64];
    __shared__ int32_t ids[2][AM_MB][S];
    __shared__ __attribute__((aligned(16))) int32_t soff[16 * KW];      // dof offsets of the current representative's list
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, lj = lane & 15, lk = lane >> 4;
    const int lm = tid >> 4, lc = tid & 15;     // id loads: subdomain lm of the batch, columns 4 lc + 64 u ...
    // XCD-contiguous ranges (neighbouring boxes gather overlapping parts of r: one L2)
    const int nwg = gridDim.x, q_ = nwg >> 3, rem_ = nwg & 7, xcd_ = blockIdx.x & 7, within_ = blockIdx.x >> 3;
    const int wg = (xcd_ < rem_ ? xcd_ * (q_ + 1) : rem_ * (q_ + 1) + (xcd_ - rem_) * q_) + within_;
    const int32_t p_end = min(nsub, (wg + 1) * span);   // span = a multiple of 16
    int32_t p_chunk = wg * span;
    if (p_chunk >= p_end) return;
    // chunk = 64 places: lane = place; record = (subdomain, representative, columns | owned rows << 10 | conforming << 20,
    // first dof), see k_pack_order
    const int4 none = make_int4(0, -1, 0, 0);
    int32_t soff_rep = -1;                      // representative whose offsets soff holds
    int4 hdr = p_chunk + lane < p_end ? order[p_chunk + lane] : none;
    int4 hdr_n = p_chunk + 64 + lane < p_end ? order[p_chunk + 64 + lane] : none;
    int cnt = min(64, p_end - p_chunk);
    auto run_starts = [&](const int4& h, int cnt_) {
        const int32_t pr_rep = __shfl_up(h.y, 1, 64), pr_n = __shfl_up(h.z, 1, 64);
        return __ballot(lane == 0 || lane >= cnt_ || h.y != pr_rep || h.z != pr_n);
    };
    auto batch_len = [&](uint64_t starts_, int cnt_, int pos_) {   // a run of equal representatives, at most 16
        const uint64_t later = pos_ < 63 ? starts_ >> (pos_ + 1) : 0ull;      // bit k: a run starts at pos + 1 + k
        const int mb_ = later ? __builtin_ctzll(later) + 1 : 64 - pos_;
        return min(min(mb_, AM_MB), cnt_ - pos_);
    };
    // the dof list of subdomain lm of a batch, this lane's columns (whole rows: sub_dofs rows are NMAX long)
    // ... or, for a batch of conforming subdomains of the representative whose offsets are in soff, computed: first dof +
    // offset (no read of the list: 124 MB per apply at 214^3 cells)
    auto load_ids = [&](const int4& h, int pos_, int mb_, int n_, int4 (&v)[U]) {
        const int32_t sm = __shfl(h.x, pos_ + lm, 64);
        const int32_t a0 = __shfl(h.w, pos_ + lm, 64);
        const bool computed = ((__builtin_amdgcn_readlane(h.z, pos_) >> 20) & 1) && __builtin_amdgcn_readlane(h.y, pos_) == soff_rep;
        const int4* __restrict__ row = (const int4*)(sub_dofs + (int64_t)sm * NMAX);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c4 = lc + 16 * u;
            const bool on = lm < mb_ && 4 * c4 < n_ && 4 * c4 < 16 * KW;
            if (computed) {     // (uniform over the workgroup)
                const int4 o = on ? reinterpret_cast<const int4*>(soff)[c4] : make_int4(0, 0, 0, 0);
                v[u] = make_int4(a0 + o.x, a0 + o.y, a0 + o.z, a0 + o.w);
            } else {
                v[u] = on ? row[c4] : make_int4(0, 0, 0, 0);
            }
        }
    };
    auto park_ids = [&](int buf, const int4 (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = 4 * (lc + 16 * u);
            if (c < 16 * KW) {
                int32_t* dst = &ids[buf][lm][c];
                dst[0] = v[u].x;
                dst[1] = v[u].y;
                dst[2] = v[u].z;
                dst[3] = v[u].w;
            }
        }
    };
    // a batch is "direct" when its subdomains conform and the offsets of its representative are the ones in soff: its
    // ids are then formed where they are used (first dof + offset) and never pass through the LDS lists
    auto is_direct = [&](const int4& h, int pos_) {
        return ((__builtin_amdgcn_readlane(h.z, pos_) >> 20) & 1) && __builtin_amdgcn_readlane(h.y, pos_) == soff_rep;
    };
    uint64_t starts = run_starts(hdr, cnt);
    int pos = 0, mb = batch_len(starts, cnt, 0), buf = 0;
    {
        int4 v[U];
        load_ids(hdr, 0, mb, __builtin_amdgcn_readlane(hdr.z, 0) & 1023, v);
        park_ids(0, v);
    }
    __syncthreads();
    int32_t cur = -1;
    bool direct = false;        // (the first batch of a workgroup reads its lists: soff is not loaded yet)
    double a[RT][KW];
    // B fragments and output rows of a direct batch are gathered one batch ahead (while the current one is multiplied,
    // exchanged and stored): the kernel is bound by the latency of these gathers, not by the matrix cores
    double bv[KW];
    int32_t od_p[RT];
    for (;;) {
        const int32_t rp = __builtin_amdgcn_readlane(hdr.y, pos);
        const int32_t packed = __builtin_amdgcn_readlane(hdr.z, pos);
        const int n = packed & 1023, nrow = (packed >> 10) & 1023;
        if (rp != cur) {    // (uniform) this lane's A fragments of the new inverse, and the representative's dof offsets
            const int32_t s0 = __builtin_amdgcn_readlane(hdr.x, pos);
            const double* __restrict__ src = inv + inv_ptr[s0];
            {
                const int32_t* __restrict__ ref = sub_dofs + (int64_t)rp * NMAX;
                const int32_t r0 = ref[0];
                if (tid < 16 * KW) soff[tid] = tid < n ? ref[tid] - r0 : 0;
            }
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const int c = 4 * (w + 4 * kk) + lk;
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int i = 16 * t + lj;
                    a[t][kk] = (c < n && i < nrow) ? src[c * nrow + i] : 0.0;
                }
            }
            cur = rp;
            __syncthreads();    // soff is read below (next batch's ids); rare: once per run of a representative
            soff_rep = rp;
        }
        // B fragments: entries of r at the dof ids of subdomain lj
        int32_t od[RT];     // the rows this lane writes at the end: result register w of tile t = row 16 t + lk + 4 w of subdomain lj
        if (direct) {       // (uniform over the workgroup) bv was requested during the previous batch
#pragma unroll
            for (int t = 0; t < RT; ++t) od[t] = od_p[t];
        } else {
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const int c = 4 * (w + 4 * kk) + lk;
                bv[kk] = (ABL & 1) ? (double)(c + lj) : ((lj < mb && c < n) ? r[ids[buf][lj][c]] : 0.0);
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int i = 16 * t + lk + 4 * w;
                od[t] = (lj < mb && i < nrow) ? ids[buf][lj][i] : -1;
            }
        }
        // where the next batch is: in this chunk or at the start of the next one; its dof lists are requested now
        int pos_n = pos + mb;
        const bool cross = pos_n >= cnt;
        const bool last = cross && p_chunk + 64 >= p_end;
        int4 v[U];
        int mb_n = 0;
        uint64_t starts_n = starts;
        int cnt_n = cnt;
        bool direct_n = false;
        if (!last) {
            if (cross) {
                cnt_n = min(64, p_end - (p_chunk + 64));
                starts_n = run_starts(hdr_n, cnt_n);
                pos_n = 0;
                mb_n = batch_len(starts_n, cnt_n, 0);
                direct_n = is_direct(hdr_n, 0);
                if (!direct_n) load_ids(hdr_n, 0, mb_n, __builtin_amdgcn_readlane(hdr_n.z, 0) & 1023, v);
            } else {
                mb_n = batch_len(starts, cnt, pos_n);
                direct_n = is_direct(hdr, pos_n);
                if (!direct_n) load_ids(hdr, pos_n, mb_n, __builtin_amdgcn_readlane(hdr.z, pos_n) & 1023, v);
            }
        }
        ap_d4 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = ap_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KW; ++kk)
            if ((ABL & 32) || 4 * (w + 4 * kk) < n) {     // (uniform over the wave)
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    if (ABL & 8) acc[t][kk & 3] += a[t][kk] * bv[kk];
                    else acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][kk], bv[kk], acc[t], 0, 0, 0);
                }
            }
        // (uniform) the next batch's entries of r and its output rows (first dof + offset): the fragments of this batch are
        // spent, the gathers fly while the partial tiles are exchanged, added and stored
        if (direct_n) {
            const int32_t a0n = cross ? __shfl(hdr_n.w, pos_n + lj, 64) : __shfl(hdr.w, pos_n + lj, 64);
            const int32_t pk = cross ? __builtin_amdgcn_readlane(hdr_n.z, pos_n) : __builtin_amdgcn_readlane(hdr.z, pos_n);
            const int n_n = pk & 1023, nrow_n = (pk >> 10) & 1023;
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const int c = 4 * (w + 4 * kk) + lk;
                bv[kk] = (ABL & 1) ? (double)(c + a0n) : ((lj < mb_n && c < n_n) ? r[a0n + soff[c]] : 0.0);
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                const int i = 16 * t + lk + 4 * w;
                od_p[t] = (lj < mb_n && i < nrow_n) ? a0n + soff[i] : -1;
            }
        }
        if (!(ABL & 4)) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) part[w][t][q][lane] = acc[t][q];
        }
        if (!(ABL & 16)) __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const double sum = (ABL & 4) ? ((acc[t][0] + acc[t][1]) + acc[t][2]) + acc[t][3]
                                         : ((part[0][t][w][lane] + part[1][t][w][lane]) + part[2][t][w][lane]) + part[3][t][w][lane];
            if (od[t] >= 0 && (!(ABL & 2) || sum == 1.2345e300)) z[od[t]] = sum;
        }
        if (last) break;
        if (!direct_n) park_ids(buf ^ 1, v);
        if (cross) {        // (uniform) next chunk: its records were requested a chunk ago; request the one after it
            p_chunk += 64;
            hdr = hdr_n;
            hdr_n = p_chunk + 64 + lane < p_end ? order[p_chunk + 64 + lane] : none;
        }
        if (!(ABL & 16)) __syncthreads();    // part and ids[buf] are rewritten by the next batch; ids[buf ^ 1] is complete
        buf ^= 1;
        direct = direct_n;
        pos = pos_n;
        mb = mb_n;
        starts = starts_n;
        cnt = cnt_n;
    }
}


// The same batched product on a BATCH TABLE (round 4, the default when every subdomain conforms).  What bounds k_apply_mfma is
// neither its gathers nor the matrix cores: with the gathers of r, the stores of z and the exchange of the partial tiles
// taken out it still runs 103 of its 127 us at 214^3 cells, while its 48 matrix instructions per wave and batch alone
// sustain 69-74 TFLOP/s on this chip (tools/microbench/mfma_f64.hip: 57 us for the same products).  The rest is the
// instruction stream around them -- chunk records, ballots and bit scans to find the batch boundaries, predicated loads
// compiled to one exec-mask branch each, the list path woven through the loop --, some 3 000 cycles per wave and batch,
// which the two waves of a SIMD run through together instead of one under the other's matrix instructions.
// Here the batches are cut at setup (k_bt_*: one descriptor of 20 ints per batch: representative, sizes, subdomains in
// the batch, subdomain of the inverse, the sixteen first dofs -- absent ones repeat the first), the control values of a
// batch are wave-uniform scalars, every gather is unconditional (columns beyond the list read the first dof: their entries
// of A are zero), and the entry of r the NEXT batch needs in a fragment register is requested as soon as the last product
// of this batch that reads the register has been issued -- between the matrix instructions, not behind them.  Products
// and summation order are those of k_apply_mfma: the same bits.
constexpr int BT_W = 20;    // ints per batch descriptor: rep | n + (nrow << 10) | subdomains | subdomain of the inverse | first dofs [16]
template <int RT, int KW, bool DBG = false>   // owned rows <= 16 RT, columns <= 16 KW; DBG: phase clocks of one wave (development)
__global__ __launch_bounds__(256, (RT <= 2 && KW <= 10) ? 3 : ((RT <= 4 && KW <= 12) ? 2 : 1)) void k_apply_bt(const int32_t* __restrict__ bt, const int32_t* __restrict__ sub_dofs,
                                                    const int64_t* __restrict__ inv_ptr, const double* __restrict__ inv,
                                                    const double* __restrict__ r, double* __restrict__ z, int32_t nbatch, int bspan) {
    __shared__ double part[4][RT][4][64];
    long long t_begin = 0, t_loop = 0;
    int n_reload = 0;
    if (DBG) t_begin = __builtin_readcyclecounter();
    const int tid = threadIdx.x, lane = tid & 63, lj = lane & 15, lk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous ranges (neighbouring boxes gather overlapping parts of r: one L2)
    const int nwg = gridDim.x, q_ = nwg >> 3, rem_ = nwg & 7, xcd_ = blockIdx.x & 7, within_ = blockIdx.x >> 3;
    const int wg = (xcd_ < rem_ ? xcd_ * (q_ + 1) : rem_ * (q_ + 1) + (xcd_ - rem_) * q_) + within_;
    int32_t b = wg * bspan;
    const int32_t b_end = min(nbatch, b + bspan);
    if (b >= b_end) return;
    const int4* __restrict__ bt4 = reinterpret_cast<const int4*>(bt);
    int4 h = bt4[(int64_t)b * (BT_W / 4)];
    int32_t a0 = bt[(int64_t)b * BT_W + 4 + lj];
    // (descriptors are read two batches ahead; past the end of the range the last one again: every load of the loop is issued
    // in every trip -- behind a branch the compiler's count of the loads in flight goes down by one per branch, and the last
    // product steps then waited for gathers issued a few hundred cycles before them)
    const int32_t bn_ = min(b + 1, b_end - 1);
    int4 hn = bt4[(int64_t)bn_ * (BT_W / 4)];
    int32_t a0n = bt[(int64_t)bn_ * BT_W + 4 + lj];
    int32_t so_rep = -1, cur = -1;
    // byte offsets of the representative's dof list in registers: this lane's KW columns (entries beyond the list: 0 -- they
    // meet zeros of A) and its RT output rows.  With 32-bit byte offsets on a uniform base the gathers and the stores are one
    // integer add each (r and z are far below 4 GB: 32-bit dof ids)
    uint32_t so_c[KW], so_o[RT];
    auto load_offsets = [&](int32_t rp_, int n_, int nrow_) {
        const int32_t* __restrict__ ref = sub_dofs + (int64_t)rp_ * NMAX;
        const int32_t r0 = ref[0];
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            const int c = 4 * (w + 4 * kk) + lk;
            so_c[kk] = (uint32_t)(ref[c < n_ ? c : 0] - r0) * 8u;
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int i = 16 * t + lk + 4 * w;
            so_o[t] = (uint32_t)(ref[i < nrow_ ? i : 0] - r0) * 8u;
        }
        so_rep = rp_;
    };
    const char* __restrict__ rb = reinterpret_cast<const char*>(r);
    char* __restrict__ zb = reinterpret_cast<char*>(z);
    constexpr uint32_t NO_ROW = 0xffffffffu;
    // the rows this lane writes at the end: result register w of tile t = row 16 t + lk + 4 w of subdomain lj
    auto out_rows = [&](const int4& hh, int32_t a0_, uint32_t (&o)[RT]) {
        const int32_t pk = __builtin_amdgcn_readfirstlane(hh.y), mb_ = __builtin_amdgcn_readfirstlane(hh.z);
        const int nrow_ = (pk >> 10) & 1023;
#pragma unroll
        for (int t = 0; t < RT; ++t) o[t] = (lj < mb_ && 16 * t + lk + 4 * w < nrow_) ? (uint32_t)a0_ * 8u + so_o[t] : NO_ROW;
    };
    double a[RT][KW], bv[KW];
    uint32_t od[RT], odn[RT];
    {
        const int32_t pk = __builtin_amdgcn_readfirstlane(h.y);
        load_offsets(__builtin_amdgcn_readfirstlane(h.x), pk & 1023, (pk >> 10) & 1023);
        // B fragments: lane (k, j) = entry 4 step + k of subdomain j's restriction of r
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) bv[kk] = *reinterpret_cast<const double*>(rb + ((uint32_t)a0 * 8u + so_c[kk]));
        out_rows(h, a0, od);
    }
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    int nbt = 0;
    if (DBG) t_loop = __builtin_readcyclecounter();
    for (;;) {
        long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (DBG) t0 = __builtin_readcyclecounter();
        const int32_t rp = __builtin_amdgcn_readfirstlane(h.x), packed = __builtin_amdgcn_readfirstlane(h.y);
        const int n = packed & 1023, nrow = (packed >> 10) & 1023;
        if (rp != cur) {    // (uniform) this lane's A fragments of the new inverse
            const double* __restrict__ src = inv + inv_ptr[__builtin_amdgcn_readfirstlane(h.w)];
            // (all loads first, unconditional on a clamped index, then the entries outside the slab to zero: with the
            // predicate in the load each of the RT KW loads waited for its own data)
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const int c = 4 * (w + 4 * kk) + lk;
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const int i = 16 * t + lj;
                    a[t][kk] = __builtin_nontemporal_load(src + ((c < n && i < nrow) ? c * nrow + i : 0));
                }
            }
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                const int c = 4 * (w + 4 * kk) + lk;
#pragma unroll
                for (int t = 0; t < RT; ++t)
                    if (!(c < n && 16 * t + lj < nrow)) a[t][kk] = 0.0;
            }
            cur = rp;
            if (DBG) ++n_reload;
        }
        const bool more = b + 1 < b_end;    // (uniform; the last trip prepares its own batch once more and drops it)
        {
            const int32_t rpn = __builtin_amdgcn_readfirstlane(hn.x);
            if (rpn != so_rep) {    // (uniform, rare) the next batch has another representative: its offsets
                const int32_t pkn = __builtin_amdgcn_readfirstlane(hn.y);
                load_offsets(rpn, pkn & 1023, (pkn >> 10) & 1023);
            }
        }
        out_rows(hn, a0n, odn);
        const uint32_t a0n8 = (uint32_t)a0n * 8u;
        const int32_t b2_ = min(b + 2, b_end - 1);
        const int4 h2 = bt4[(int64_t)b2_ * (BT_W / 4)];
        const int32_t a02 = bt[(int64_t)b2_ * BT_W + 4 + lj];
        if (DBG) t1 = __builtin_readcyclecounter();
        ap_d4 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = ap_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            // (scalar; only the last steps can lie beyond the list of a box that fills the kernel's shape, and a step beyond
            // it multiplies zeros of A -- the same bits with or without it --, so the first steps carry no test)
            if (kk + 2 < KW || 4 * (w + 4 * kk) < n) {
#pragma unroll
                for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][kk], bv[kk], acc[t], 0, 0, 0);
            }
            // this step's fragment is spent: the next batch's entry of r takes its place and flies while the remaining steps are
            // multiplied and the partial tiles exchanged and stored (unconditional: columns beyond the list read the first dof
            // against entries of A that are zero)
            bv[kk] = *reinterpret_cast<const double*>(rb + (a0n8 + so_c[kk]));
        }
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[w][t][q][lane] = acc[t][q];
        if (DBG) t2 = __builtin_readcyclecounter();
        __syncthreads();
        if (DBG) t3 = __builtin_readcyclecounter();
        double zs[RT];      // (all sums first: behind the predicate each tile's LDS reads and adds waited for the one before)
#pragma unroll
        for (int t = 0; t < RT; ++t)
            zs[t] = ((part[0][t][w][lane] + part[1][t][w][lane]) + part[2][t][w][lane]) + part[3][t][w][lane];
#pragma unroll
        for (int t = 0; t < RT; ++t)
            if (od[t] != NO_ROW) *reinterpret_cast<double*>(zb + od[t]) = zs[t];
        if (DBG) {
            t4 = __builtin_readcyclecounter();
            tk[0] += t1 - t0;
            tk[1] += t2 - t1;
            tk[2] += t3 - t2;
            tk[3] += t4 - t3;
            ++nbt;
        }
        if (!more) break;
#pragma unroll
        for (int t = 0; t < RT; ++t) od[t] = odn[t];
        h = hn;
        a0 = a0n;
        hn = h2;
        a0n = a02;
        ++b;
        __syncthreads();    // part is rewritten by the next batch (two buffers and one barrier per batch: measured, no gain)
        if (DBG) tk[4] += __builtin_readcyclecounter() - t4;
    }
    if (DBG && tid == 0 && (blockIdx.x % 31 == 0 || blockIdx.x + 8 >= gridDim.x)) {
        const long long t_end = __builtin_readcyclecounter();
        printf("[k_apply_bt] wg %4d (range %4d): begin %lld, prologue %lld, loop %lld ticks; %d batches, %d inverses; per batch: top %lld, products %lld, "
               "barrier %lld, sum + stores %lld, rotate + barrier %lld\n", (int)blockIdx.x, wg, t_begin & 0xffffff, t_loop - t_begin, t_end - t_loop, nbt,
               n_reload, tk[0] / nbt, tk[1] / nbt, tk[2] / nbt, tk[3] / nbt, tk[4] / max(nbt - 1, 1));
    }
}

// ---- the batch table of k_apply_bt (setup) ----
// a run = consecutive places of the apply order with the same representative and sizes (and not across `p_cut`, the first
// place of the subdomains with ghost dofs when the order is split); a batch = up to sixteen consecutive places of a run
__global__ void k_bt_flag(const int4* __restrict__ rec, int32_t n, int32_t p_cut, int32_t* __restrict__ start) {
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    bool s = p == 0 || p == p_cut;
    if (!s) {
        const int4 a = rec[p], b = rec[p - 1];
        s = a.y != b.y || a.z != b.z;
    }
    start[p] = s ? 1 : 0;
}
// run_first[run] = first place of the run (runid = exclusive scan of the flags, +1 at a start - 1 = the run of place p)
__global__ void k_bt_run_first(const int32_t* __restrict__ start, const int32_t* __restrict__ scan, int32_t n, int32_t nruns,
                               int32_t* __restrict__ run_first) {
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) run_first[nruns] = n;
    if (p < n && start[p]) run_first[scan[p]] = p;
}
__global__ void k_bt_batch_flag(const int32_t* __restrict__ start, const int32_t* __restrict__ scan, const int32_t* __restrict__ run_first,
                                int32_t n, int32_t* __restrict__ bflag) {
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t run = scan[p] + start[p] - 1;
    bflag[p] = ((p - run_first[run]) % AM_MB) == 0 ? 1 : 0;
}
__global__ void k_bt_fill(const int4* __restrict__ rec, const int32_t* __restrict__ start, const int32_t* __restrict__ scan,
                          const int32_t* __restrict__ run_first, const int32_t* __restrict__ bflag, const int32_t* __restrict__ bscan,
                          int32_t n, int32_t* __restrict__ bt) {
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || !bflag[p]) return;
    const int32_t run = scan[p] + start[p] - 1;
    const int32_t mb = min(AM_MB, run_first[run + 1] - p);
    const int4 h = rec[p];
    int32_t* d = bt + (int64_t)bscan[p] * BT_W;
    d[0] = h.y;
    d[1] = h.z & ((1 << 20) - 1);
    d[2] = mb;
    d[3] = h.x;
    for (int j = 0; j < AM_MB; ++j) d[4 + j] = j < mb ? rec[p + j].w : h.w;
}

// Warp-specialised form of the same batched product (round 4; option "apply_kind" 7 -- NOT the default: measured slower).
// A workgroup of EIGHT waves, one per CU, persistent over up to 1024 places: waves 0-3 multiply -- wave w owns row tile w of the
// shared inverse, its A fragments for all column steps in registers -- and waves 4-7 load: they gather the 16 subdomains'
// restrictions of r three batches ahead (two register sets in flight) and park them in LDS as B[column][subdomain]
// (double-buffered) together with the output row ids.  One barrier per batch, no partial tiles to add, the matrix-core loop is
// back-to-back MFMAs behind software-pipelined LDS reads.  Measured at 214^3 cells / 166 375 subdomains: 224 us against the
// 131 us of the K-split kernel above on the same box (33.9 against 46 us at 19 683 subdomains).  The counters say why
// (tools/pmc_apply.sh, profiles/r04_pmc_apply.txt): both kernels do the same 3.7 M MFMA-busy cycles per SE and fetch 2.8-4.2 M
// lines from L2 per apply -- r is re-fetched four to six times through the 16 KB L1s, 355-532 MB for the 80 MB vector -- and
// the L1s spend 38 % (K-split) / 53 % (this kernel) of the time with their miss queues full (TCP_PENDING_STALL_CYCLES): the
// apply is bound by outstanding L1 misses x L2/MALL latency per CU, and two independent workgroups per CU keep more of them
// in flight than one workgroup with four loader waves.  Kept as an option for A/B; what would help either kernel is fetching
// fewer lines (a workgroup owning a 3D block of boxes and staging the union of their overlap bricks once), not more overlap.
// Dof ids: first dof + the representative's offsets for conforming subdomains (the loader lanes keep the offsets of their
// columns in registers), the stored lists otherwise.
constexpr int WS_SPAN = 1024;     // most places of a workgroup of the warp-specialised apply kernel
template <int NK>    // column steps of 4: columns <= 4 NK; owned rows <= 64 (four row tiles)
__global__ __launch_bounds__(512, 1) void k_apply_ws(const int4* __restrict__ order, const int32_t* __restrict__ sub_dofs,
                                                     const int64_t* __restrict__ inv_ptr, const double* __restrict__ inv,
                                                     const double* __restrict__ r, double* __restrict__ z, int32_t nsub, int span) {
    constexpr int NC = 4 * NK, NG = NC / 16;    // columns, gathers per loader lane and batch
    __shared__ double Bs[2][NC][16];
    __shared__ int32_t ods[2][64][16];
    __shared__ int4 rec[WS_SPAN];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const bool loader = wv >= 4;
    const int nwg = gridDim.x, q_ = nwg >> 3, rem_ = nwg & 7, xcd_ = blockIdx.x & 7, within_ = blockIdx.x >> 3;
    const int wg = (xcd_ < rem_ ? xcd_ * (q_ + 1) : rem_ * (q_ + 1) + (xcd_ - rem_) * q_) + within_;
    const int32_t p0 = wg * span, cnt = min(span, nsub - p0);
    if (cnt <= 0) return;
    for (int i = tid; i < cnt; i += 512) rec[i] = order[p0 + i];
    __syncthreads();
    // a batch = up to 16 consecutive places with the same representative and sizes (one LDS read and a ballot per wave)
    auto batch_len = [&](int pos_) -> int {
        if (pos_ >= cnt) return 0;
        const int4 h0 = rec[pos_];
        const int4 hl = rec[min(pos_ + min(lane, AM_MB - 1), cnt - 1)];
        const uint64_t diff = __ballot(lane < AM_MB && (pos_ + lane >= cnt || hl.y != h0.y || hl.z != h0.z));
        return diff ? (int)__builtin_ctzll(diff) : AM_MB;
    };
    if (loader) {
        // batches are gathered THREE iterations before they are multiplied (two register sets in flight, parked one iteration
        // ahead): a batch of gathers takes longer to come back than a batch of matrix-core products takes
        const int lt = tid - 256, gj = lt & 15, gc = lt >> 4;   // subdomain gj of the batch, columns gc + 16 u
        int32_t so[NG];
        int32_t so_rep = -1;
        auto gather = [&](int pos_, int mb_, double (&g)[NG], int32_t (&id4)[4]) {
            const int4 h = rec[pos_ + min(gj, mb_ - 1)];
            const int n_ = h.z & 1023, nrow_ = (h.z >> 10) & 1023;
            const bool conf = (h.z >> 20) & 1;
            const int32_t* __restrict__ row = sub_dofs + (int64_t)h.x * NMAX;
            if (__builtin_amdgcn_readfirstlane(h.y) != so_rep) {        // (uniform: a batch has one representative)
                const int32_t* __restrict__ ref = sub_dofs + (int64_t)h.y * NMAX;
                const int32_t r0 = ref[0];
#pragma unroll
                for (int u = 0; u < NG; ++u) so[u] = gc + 16 * u < n_ ? ref[gc + 16 * u] - r0 : 0;
                so_rep = __builtin_amdgcn_readfirstlane(h.y);
            }
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int c = gc + 16 * u;
                const bool on = gj < mb_ && c < n_;
                const int32_t id = on ? (conf ? h.w + so[u] : row[c]) : 0;
                g[u] = on ? r[id] : 0.0;
                if (u < 4) id4[u] = (on && c < nrow_) ? id : -1;
            }
        };
        auto park = [&](int buf_, const double (&g)[NG], const int32_t (&id4)[4]) {
#pragma unroll
            for (int u = 0; u < NG; ++u) Bs[buf_][gc + 16 * u][gj] = g[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) ods[buf_][gc + 16 * u][gj] = id4[u];
        };
        double gA[NG], gB[NG];
        int32_t iA[4], iB[4];
        // batches 0, 1, 2: 0 parked now, 1 in set B, 2 in set A
        int pos = 0, mb = batch_len(0);
        gather(pos, mb, gA, iA);
        park(0, gA, iA);
        pos += mb;
        int mb1 = batch_len(pos);                   // batch i + 1 (in set B)
        if (mb1 > 0) gather(pos, mb1, gB, iB);
        pos += mb1;
        int mb2 = mb1 > 0 ? batch_len(pos) : 0;     // batch i + 2 (in set A)
        if (mb2 > 0) gather(pos, mb2, gA, iA);
        pos += mb2;
        __syncthreads();
        int buf = 0;
        for (;;) {
            // iteration i (even): park batch i + 1 from set B, request batch i + 3 into set B
            if (mb1 == 0) break;
            park(buf ^ 1, gB, iB);
            int mb3 = mb2 > 0 ? batch_len(pos) : 0;
            if (mb3 > 0) gather(pos, mb3, gB, iB);
            pos += mb3;
            __syncthreads();
            buf ^= 1;
            // iteration i + 1 (odd): park batch i + 2 from set A, request batch i + 4 into set A
            if (mb2 == 0) break;
            park(buf ^ 1, gA, iA);
            int mb4 = mb3 > 0 ? batch_len(pos) : 0;
            if (mb4 > 0) gather(pos, mb4, gA, iA);
            pos += mb4;
            __syncthreads();
            buf ^= 1;
            mb1 = mb3;
            mb2 = mb4;
        }
    } else {
        const int w = wv, lj = lane & 15, lk = lane >> 4;
        double a[NK];
        int32_t cur = -1;
        int pos_a = 0, mb_a = batch_len(0);
        int pos_b = mb_a, mb_b = batch_len(pos_b);
        __syncthreads();
        int buf = 0;
        for (;;) {
            const int4 h = rec[pos_a];
            const int n = h.z & 1023, nrow = (h.z >> 10) & 1023;
            if (h.y != cur) {       // (uniform) this wave's row tile of the new inverse
                const double* __restrict__ src = inv + inv_ptr[h.x];
                const int i = 16 * w + lj;
#pragma unroll
                for (int s = 0; s < NK; ++s) {
                    const int c = 4 * s + lk;
                    const bool on = c < n && i < nrow;
                    const double v = src[on ? c * nrow + i : 0];
                    a[s] = on ? v : 0.0;
                }
                cur = h.y;
            }
            if (16 * w < nrow) {    // (uniform over the wave)
                int32_t od[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) od[q] = ods[buf][16 * w + lk + 4 * q][lj];
                ap_d4 acc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = ap_d4{0.0, 0.0, 0.0, 0.0};
                // eight column steps at a time, the next eight B fragments requested before the current products (the scheduling
                // barriers keep the compiler from hoisting all NK reads to the front, which spilled 300 registers)
                double b[8], bn[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) b[u] = Bs[buf][4 * u + lk][lj];
#pragma unroll
                for (int s0 = 0; s0 < NK; s0 += 8) {
                    if (s0 + 8 < NK) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) bn[u] = Bs[buf][4 * (s0 + 8 + u) + lk][lj];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s0 + u], b[u], acc[u & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 8; ++u) b[u] = bn[u];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (od[q] >= 0) z[od[q]] = (acc[0][q] + acc[1][q]) + (acc[2][q] + acc[3][q]);
            }
            if (mb_b == 0) break;
            const int pos_c = pos_b + mb_b, mb_c = batch_len(pos_c);
            __syncthreads();
            buf ^= 1;
            pos_a = pos_b;
            mb_a = mb_b;
            pos_b = pos_c;
            mb_b = mb_c;
        }
    }
}

__global__ void k_count_mult(const int32_t* __restrict__ sub_n, const int32_t* __restrict__ sub_dofs, double* mult) {
    const int b = blockIdx.x;
    const int n = sub_n[b];
    for (int c = threadIdx.x; c < n; c += blockDim.x) atomicAdd(&mult[sub_dofs[(int64_t)b * NMAX + c]], 1.0);
}

__global__ void k_div(double* __restrict__ z, const double* __restrict__ m, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = z[i] / m[i];
}

}  // namespace

int bounding_box(fedd_ctx* c, int64_t n_nodes, double lo[3], double hi[3], double* d_scratch) {
    const int nblk = 128;     // (d_scratch: 768 doubles of the caller's; nullptr: the context's d_dtmp0)
    if (!d_scratch) {
        FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)nblk * 6, c->d_dtmp0.cap)));
        d_scratch = c->d_dtmp0.p;
    }
    hipLaunchKernelGGL(k_minmax, dim3(nblk), dim3(256), 0, c->stream, (const double*)c->d_xyz.p, (int32_t)n_nodes, c->dim,
                       d_scratch);
    std::vector<double> part((size_t)nblk * 6);
    FEDD_HIP(hipMemcpyAsync(part.data(), d_scratch, part.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    for (int d = 0; d < 3; ++d) {
        lo[d] = 0.0;
        hi[d] = 0.0;
        if (d >= c->dim) continue;
        double mn = 1e300, mx = -1e300;
        for (int k = 0; k < nblk; ++k) {
            mn = std::min(mn, part[(size_t)k * 6 + d]);
            mx = std::max(mx, part[(size_t)k * 6 + 3 + d]);
        }
        lo[d] = mn;
        hi[d] = mx;
    }
    return 0;
}

// bounding box of the owned nodes of ALL ranks and their number (min / max through the sum transport:
// every rank fills its own slots); every rank gets the same numbers
int global_box(fedd_ctx* c, int64_t n_own, double lo[3], double hi[3], double* n_global) {
    FEDD_TRY(bounding_box(c, n_own, lo, hi));
    *n_global = (double)n_own;
    if (c->nranks == 1) return 0;
    const int nr = c->nranks, len = nr * 7;
    std::vector<double> h((size_t)len, 0.0);
    for (int d = 0; d < 3; ++d) {
        h[(size_t)c->rank * 7 + d] = lo[d];
        h[(size_t)c->rank * 7 + 3 + d] = hi[d];
    }
    h[(size_t)c->rank * 7 + 6] = (double)n_own;
    FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)len, c->d_dtmp0.cap)));
    FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, h.data(), (size_t)len * sizeof(double), hipMemcpyHostToDevice, c->stream));
    FEDD_TRY(allreduce_sum(c, c->d_dtmp0.p, len));
    FEDD_HIP(hipMemcpyAsync(h.data(), c->d_dtmp0.p, (size_t)len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    *n_global = 0.0;
    for (int r = 0; r < nr; ++r) {
        // every rank sees every rank's count: all of them give up together
        FEDD_CHECK(h[(size_t)r * 7 + 6] > 0.0, "rank %d owns no nodes", r);
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], h[(size_t)r * 7 + d]);
            hi[d] = std::max(hi[d], h[(size_t)r * 7 + 3 + d]);
        }
        *n_global += h[(size_t)r * 7 + 6];
    }
    return 0;
}

// overlapping dof lists with the 1024-entry stride of the large-subdomain path (schwarz_big.hip): bins in
// d_bin_ptr / d_bin_nodes / d_node_bin -> d_sub_n / d_sub_nown / d_sub_dofs [nsub * 1024]
int schwarz_overlap_lists_big(fedd_ctx* c, int64_t nsub, int32_t* max_n, int32_t* max_own) {
    FEDD_TRY(c->d_sub_n.ensure((size_t)nsub));
    FEDD_TRY(c->d_sub_nown.ensure((size_t)nsub));
    FEDD_TRY(c->d_sub_dofs.ensure((size_t)nsub * SCHWARZ_NMAX_BIG));
    hipLaunchKernelGGL((k_sub_dofs<SCHWARZ_NMAX_BIG, 4096>), dim3((unsigned)nsub), dim3(256), 0, c->stream,
                       (const int32_t*)c->d_bin_ptr.p, (const int32_t*)c->d_bin_nodes.p, (const int32_t*)c->d_node_bin.p,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)c->d_rowptr.p,
                       (const int32_t*)c->d_colind.p, (int32_t)c->n_rows, (int32_t)c->n_rows_ext, c->ghost_overlap,
                       c->sw_overlap, c->d_sub_n.p, c->d_sub_nown.p, c->d_sub_dofs.p);
    FEDD_TRY(reduce_max_i32(c, c->d_sub_n.p, nsub, max_n));
    FEDD_TRY(reduce_max_i32(c, c->d_sub_nown.p, nsub, max_own));
    FEDD_HIP(hipGetLastError());
    return 0;
}

// slab offsets (elements) of every subdomain -> d_inv_ptr, total -> sw_inv_elems, d_inv sized
int schwarz_slab_offsets(fedd_ctx* c, int64_t nsub, int restricted) {
    FEDD_TRY(c->d_inv_ptr.ensure((size_t)nsub + 1));
    hipLaunchKernelGGL(k_slab_sizes, dim3((unsigned)((nsub + 255) / 256)), dim3(256), 0, c->stream,
                       (const int32_t*)c->d_sub_n.p, (const int32_t*)c->d_sub_nown.p, (int32_t)nsub, restricted, c->d_inv_ptr.p);
    int64_t total = 0;
    FEDD_TRY(exclusive_scan_i64(c, c->d_inv_ptr.p, c->d_inv_ptr.p, nsub, &total));
    c->sw_inv_elems = total;
    FEDD_TRY(c->d_inv.ensure((size_t)total));
    return 0;
}

int schwarz_setup(fedd_ctx* c) {
    if (schwarz_use_big(c)) return schwarz_setup_big(c);
    c->sw_big_active = false;
    c->have_coarse = false;
    ScopedTimer timer(c, FEDD_T_SCHWARZ_SETUP);
    const int32_t n_own = (int32_t)c->n_own;
    const int dim = c->dim, dofs = c->dofs;
    const int32_t n_rows = (int32_t)c->n_rows;
    const int32_t n_stored = (int32_t)c->n_rows_ext;   // rows the local matrices can read (owned + row ghosts)
    // (a failed check ahead of a collective would leave the other ranks waiting in it: with several ranks the
    // rank-local conditions are tested after the first all-reduce, on numbers every rank has)
    FEDD_CHECK(n_own > 0 || c->nranks > 1, "schwarz setup: no owned nodes");
    // ---- bounding box and number of the owned nodes: of all ranks, so that the lattice of boxes is the
    // one a single rank would lay over the whole mesh and a rank boundary only cuts the boxes it crosses
    // (box_kind 1: each rank's own bounding box; costs iterations) ----
    BinGeom gm;
    gm.dim = dim;
    double L[3] = {0, 0, 0}, bb_lo[3], bb_hi[3], n_nodes = (double)n_own;
    if (c->box_kind == 0) FEDD_TRY(global_box(c, n_own, bb_lo, bb_hi, &n_nodes));
    else FEDD_TRY(bounding_box(c, n_own, bb_lo, bb_hi));
    for (int d = 0; d < dim; ++d) {
        gm.lo[d] = bb_lo[d];
        L[d] = bb_hi[d] - bb_lo[d];
    }
    // regular grid of boxes with about sw_target nodes each (same formula as the oracle)
    double V = 1.0;
    for (int d = 0; d < dim; ++d) V *= (L[d] > 0 ? L[d] : 1.0);
    // default target: 27 nodes for scalar problems, 27 / dofs for node-interleaved vector problems
    // (the dense local solver takes 256 dofs including the overlap)
    const int target = c->sw_target > 0 ? c->sw_target : (c->merged ? 27 : std::max(1, 27 / std::max(1, dofs)));
    // If a subdomain comes out larger than the dense local solver takes (NMAX dofs with the overlap), the lattice is
    // refined (box edge x 0.85) and the lists are built again: every rank takes the same decision (several
    // ranks: the largest size is all-reduced), so the lattice stays one lattice.
    int64_t nraw = 1, nsub = 0;
    int32_t max_n = 0, max_own = 0;
    bool foreign = false;
    for (int attempt = 0;; ++attempt) {
        const double s = c->sw_scale * std::pow(0.85, attempt) * std::pow(V * (double)target / n_nodes, 1.0 / dim);
        nraw = 1;
        for (int d = 0; d < 3; ++d) {
            gm.g[d] = 1;
            gm.w[d] = 1.0;
            if (d >= dim) continue;
            const double Lp = L[d] > 0 ? L[d] : 1.0;
            int g = (int)std::ceil(Lp / s - 1e-9);
            if (g < 1 || !(L[d] > 0)) g = 1;
            if (g % 2 == 0) ++g;   // odd: the lattice has a centre box (see k_bin_id)
            gm.g[d] = g;
            gm.w[d] = Lp / g;
            nraw *= g;
        }
        FEDD_CHECK(nraw < ((int64_t)1 << 30), "schwarz setup: %lld boxes", (long long)nraw);
        // ---- dofs -> boxes (box of the carrying node), drop empty boxes, counting sort ----
        // with row ghosts (and "whole_boxes") the row-ghost dofs are binned too: a box that holds owned dofs also
        // lists the other ranks' dofs inside it, and k_sub_dofs builds the whole box when the stored rows reach
        foreign = c->whole_boxes && c->box_kind == 0 && n_stored > n_rows && !c->merged;
        const int32_t n_binned = foreign ? n_stored : n_rows;
        FEDD_TRY(c->d_itmp0.ensure((size_t)n_binned));     // raw box of each dof
        FEDD_TRY(c->d_itmp1.ensure((size_t)nraw + 1));     // raw counts
        FEDD_TRY(c->d_itmp2.ensure((size_t)nraw + 1));     // flags -> compact ids
        FEDD_HIP(hipMemsetAsync(c->d_itmp1.p, 0, ((size_t)nraw + 1) * sizeof(int32_t), c->stream));
        const dim3 gn((n_rows + 255) / 256), gb((unsigned)((nraw + 255) / 256)), blk(256);
        hipLaunchKernelGGL(k_bin_id, dim3((unsigned)((n_binned + 255) / 256)), blk, 0, c->stream, (const double*)c->d_xyz.p,
                           n_binned, n_rows, dofs, (const int32_t*)(c->merged ? c->d_dof_node.p : nullptr), gm, c->d_itmp0.p,
                           c->d_itmp1.p);
        hipLaunchKernelGGL(k_flag_nonempty, gb, blk, 0, c->stream, (const int32_t*)c->d_itmp1.p, (int32_t)nraw, c->d_itmp2.p);
        nsub = 0;
        FEDD_TRY(exclusive_scan_i32(c, c->d_itmp2.p, c->d_itmp2.p, nraw, &nsub));
        FEDD_CHECK(nsub > 0, "schwarz setup: no subdomain");
        c->sw_nsub = nsub;
        FEDD_TRY(c->d_bin_ptr.ensure((size_t)nsub + 1));
        FEDD_TRY(c->d_bin_nodes.ensure((size_t)n_rows));   // dofs grouped by box
        FEDD_TRY(c->d_node_bin.ensure((size_t)n_rows));    // compact box id of each dof
        FEDD_TRY(c->d_sub_n.ensure((size_t)nsub));
        FEDD_TRY(c->d_sub_nown.ensure((size_t)nsub));
        FEDD_TRY(c->d_sub_dofs.ensure((size_t)nsub * NMAX));
        hipLaunchKernelGGL(k_compact_counts, gb, blk, 0, c->stream, (const int32_t*)c->d_itmp1.p, (const int32_t*)c->d_itmp2.p,
                           (int32_t)nraw, c->d_bin_ptr.p);
        FEDD_TRY(exclusive_scan_i32(c, c->d_bin_ptr.p, c->d_bin_ptr.p, nsub, nullptr));
        // cursor = copy of bin_ptr (reuse the raw-count buffer)
        FEDD_HIP(hipMemcpyAsync(c->d_itmp1.p, c->d_bin_ptr.p, (size_t)nsub * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_fill_bins, gn, blk, 0, c->stream, (const int32_t*)c->d_itmp0.p, (const int32_t*)c->d_itmp2.p, n_rows,
                           c->d_itmp1.p, c->d_node_bin.p, c->d_bin_nodes.p);
        hipLaunchKernelGGL(k_sort_bins, dim3((unsigned)nsub), dim3(64), 0, c->stream, (const int32_t*)c->d_bin_ptr.p,
                           (int32_t)nsub, c->d_bin_nodes.p);
        // ---- foreign members of the boxes (row-ghost dofs), same counting sort ----
        if (foreign) {
            const int32_t nfor = n_stored - n_rows;
            const dim3 gf((unsigned)((nfor + 255) / 256));
            FEDD_TRY(c->d_fbin_ptr.ensure((size_t)nsub + 1));
            FEDD_TRY(c->d_fbin_nodes.ensure((size_t)nfor));
            FEDD_HIP(hipMemsetAsync(c->d_fbin_ptr.p, 0, ((size_t)nsub + 1) * sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(k_foreign_count, gf, blk, 0, c->stream, (const int32_t*)c->d_itmp0.p, (const int32_t*)c->d_itmp2.p,
                               n_rows, n_stored, c->d_fbin_ptr.p);
            FEDD_TRY(exclusive_scan_i32(c, c->d_fbin_ptr.p, c->d_fbin_ptr.p, nsub, nullptr));
            FEDD_HIP(hipMemcpyAsync(c->d_itmp1.p, c->d_fbin_ptr.p, (size_t)nsub * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(k_foreign_fill, gf, blk, 0, c->stream, (const int32_t*)c->d_itmp0.p, (const int32_t*)c->d_itmp2.p,
                               n_rows, n_stored, c->d_itmp1.p, c->d_fbin_nodes.p);
            hipLaunchKernelGGL(k_sort_bins, dim3((unsigned)nsub), dim3(64), 0, c->stream,
                               (const int32_t*)c->d_fbin_ptr.p, (int32_t)nsub, c->d_fbin_nodes.p);
        }
        // ---- overlapping dof lists ----
        hipLaunchKernelGGL((k_sub_dofs<NMAX, HS>), dim3((unsigned)nsub), blk, 0, c->stream, (const int32_t*)c->d_bin_ptr.p,
                           (const int32_t*)c->d_bin_nodes.p, (const int32_t*)c->d_node_bin.p,
                           (const int32_t*)(foreign ? c->d_fbin_ptr.p : nullptr), (const int32_t*)(foreign ? c->d_fbin_nodes.p : nullptr),
                           (const int32_t*)c->d_rowptr.p,
                           (const int32_t*)c->d_colind.p, n_rows, (int32_t)c->n_rows_ext, c->ghost_overlap, c->sw_overlap,
                           c->d_sub_n.p, c->d_sub_nown.p,
                           c->d_sub_dofs.p);
        max_n = 0;
        FEDD_TRY(reduce_max_i32(c, c->d_sub_n.p, nsub, &max_n));
        c->sw_max_size = max_n;
        max_own = 0;
        FEDD_TRY(reduce_max_i32(c, c->d_sub_nown.p, nsub, &max_own));
        c->sw_max_own = max_own;
        if (c->nranks > 1) {   // the largest subdomain of any rank (max through the sum transport: own slot per rank)
            std::vector<double> h((size_t)c->nranks, 0.0);
            h[(size_t)c->rank] = (double)max_n;
            FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>((size_t)c->nranks, c->d_dtmp0.cap)));
            FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            FEDD_TRY(allreduce_sum(c, c->d_dtmp0.p, c->nranks));
            FEDD_HIP(hipMemcpyAsync(h.data(), c->d_dtmp0.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            FEDD_HIP(hipStreamSynchronize(c->stream));
            double mx = 0.0;
            for (double v : h) mx = std::max(mx, v);
            c->sw_max_size_all = (int32_t)mx;
        } else {
            c->sw_max_size_all = max_n;
        }
        // with the default target the lattice is also refined past the register-tiled inversion classes (160 dofs):
        // the LDS variant beyond them is an order of magnitude slower than what the few extra iterations cost
        const int32_t limit = c->sw_target > 0 ? NMAX : 160;
        if (c->sw_max_size_all <= limit || attempt >= 8) break;
    }
    max_n = std::max(max_n, c->sw_max_size_all);
    const dim3 blk(256);
    FEDD_CHECK(max_n <= NMAX,
               "schwarz setup: an overlapping subdomain has %d dofs, the dense local solver takes at most %d; "
               "lower the target with fedd_schwarz_set_target (now %d nodes)", max_n, NMAX, target);
    const int restricted = c->sw_combine == FEDD_COMBINE_RESTRICTED ? 1 : 0;
    // merged block systems: dofs >= p_off are pressures and are pivoted after the velocities
    const int32_t p_off = c->merged ? (int32_t)c->merged_nA : INT32_MAX;
    // ---- equal local matrices share one slab (option "schwarz_dedupe", default on) ----
    const int32_t* sub_n_inv = c->d_sub_n.p;      // sizes as the inversion sees them: 0 = not a representative
    c->sw_nrep = nsub;
    if (c->sw_dedupe) {
        const dim3 gs((unsigned)((nsub + 255) / 256));
        FEDD_TRY(c->d_sw_rmax.ensure((size_t)n_stored));
        const int64_t tsize = 2 * nsub + 64;
        const bool by_rows = c->sw_fp_kind == 0;
        FEDD_TRY(c->d_sw_fp.ensure((size_t)(2 * nsub + 2 * tsize) + (by_rows ? 2 * (size_t)n_stored : 0)));
        uint64_t* row_hash = c->d_sw_fp.p + 2 * nsub + 2 * tsize;
        if (by_rows)
            hipLaunchKernelGGL(k_row_hash, dim3((unsigned)((n_stored + 16 * RH_R - 1) / (16 * RH_R))), blk, 0, c->stream, (const int32_t*)c->d_rowptr.p,
                               (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, n_stored, c->d_sw_rmax.p, row_hash);
        else
            hipLaunchKernelGGL(k_row_absmax, dim3((unsigned)((n_stored + 15) / 16)), blk, 0, c->stream, (const int32_t*)c->d_rowptr.p,
                               (const double*)c->d_val.p, n_stored, c->d_sw_rmax.p);
        FEDD_TRY(c->d_sw_rep.ensure((size_t)(3 * nsub + tsize + 4)));
        uint64_t* fp = c->d_sw_fp.p;
        uint64_t* tkey = fp + 2 * nsub;
        int32_t* rep = c->d_sw_rep.p;
        int32_t* n_inv = rep + nsub;
        int32_t* slot_of = n_inv + nsub;
        int32_t* tmin = slot_of + nsub;
        int32_t* n_rep = tmin + tsize;
        FEDD_HIP(hipMemsetAsync(tkey, 0, (size_t)(2 * tsize) * sizeof(uint64_t), c->stream));
        FEDD_HIP(hipMemsetAsync(tmin, 0x7f, (size_t)tsize * sizeof(int32_t), c->stream));
        FEDD_HIP(hipMemsetAsync(n_rep, 0, sizeof(int32_t), c->stream));
        if (by_rows)
            hipLaunchKernelGGL((k_sub_fingerprint_rows<NMAX>), dim3((unsigned)nsub), dim3(64), 0, c->stream, (const int32_t*)c->d_sub_n.p,
                               (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p, (const double*)c->d_sw_rmax.p,
                               (const uint64_t*)row_hash, n_stored, p_off, fp);
        else
            hipLaunchKernelGGL((k_sub_fingerprint<NMAX>), dim3((unsigned)nsub), dim3(64), 0, c->stream, (const int32_t*)c->d_sub_n.p,
                               (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p, (const int32_t*)c->d_rowptr.p,
                               (const int32_t*)c->d_colind.p, (const double*)c->d_val.p, (const double*)c->d_sw_rmax.p, n_stored, p_off, fp);
        hipLaunchKernelGGL(k_fp_insert, gs, blk, 0, c->stream, (const uint64_t*)fp, (int32_t)nsub, tkey, tmin, tsize, slot_of);
        hipLaunchKernelGGL(k_fp_resolve, gs, blk, 0, c->stream, (const int32_t*)slot_of, (const int32_t*)tmin,
                           (const int32_t*)c->d_sub_n.p, (int32_t)nsub, rep, n_inv, n_rep);
        sub_n_inv = n_inv;
        int32_t h_nrep = 0;
        FEDD_HIP(hipMemcpyAsync(&h_nrep, n_rep, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        c->sw_nrep = h_nrep;
    }
    // ---- the order the apply walks the subdomains in: sorted by representative (stable: lattice order within one); with
    // several ranks and option "halo_overlap" the subdomains without ghost dofs first ----
    c->sw_nint = -1;
    c->sw_nconf = 0;
    c->sw_nbatch = 0;
    c->sw_nbatch_int = 0;
    {
        const bool split = c->halo_overlap && restricted && (c->n_cols != c->n_rows || !c->halo.peers.empty());
        if (c->sw_dedupe || split) {
            const dim3 gs((unsigned)((nsub + 255) / 256));
            const int32_t* rep = c->sw_dedupe ? c->d_sw_rep.p : nullptr;
            FEDD_TRY(c->d_sw_order.ensure((size_t)(8 * nsub + 8)));
            int32_t* ord = c->d_sw_order.p;
            int32_t* keys_out = ord + nsub;
            int32_t* iota = ord + 2 * nsub;
            int32_t* keys_in = ord + 3 * nsub;      // (the records overwrite this part afterwards)
            int32_t* n_int = ord + 8 * nsub + 4;
            FEDD_HIP(hipMemsetAsync(n_int, 0, 2 * sizeof(int32_t), c->stream));     // interior subdomains | conforming ones
            hipLaunchKernelGGL(k_iota, gs, blk, 0, c->stream, iota, (int32_t)nsub);
            hipLaunchKernelGGL(k_order_key, gs, blk, 0, c->stream, (const int32_t*)c->d_sub_n.p, (const int32_t*)c->d_sub_dofs.p, rep,
                               (int32_t)nsub, (int32_t)n_rows, split ? 1 : 0, keys_in, n_int);
            {
                // stable sort by key (scan.hip; keys: representative [+ 2^30 for the subdomains with ghost dofs]); the result goes to `ord`
                int bits = 0;
                while (((int64_t)1 << bits) <= (int64_t)nsub) ++bits;
                if (split) bits = 31;
                int32_t* kk[2] = {keys_in, keys_out};
                int32_t* vv[2] = {iota, ord};
                int cur = 0;
                FEDD_TRY(radix_sort_pairs_i32(c, kk, vv, (int32_t)nsub, bits, &cur));
                if (cur == 0) FEDD_HIP(hipMemcpyAsync(ord, iota, (size_t)nsub * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
            }
            // records (subdomain, representative, columns, owned rows) in that order: one 16-byte load per place
            hipLaunchKernelGGL(k_pack_order, dim3((unsigned)(((int64_t)nsub * 16 + 255) / 256)), blk, 0, c->stream, (const int32_t*)ord, rep,
                               (const int32_t*)c->d_sub_n.p, (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p,
                               (int32_t)nsub, (int4*)(ord + 4 * nsub), n_int + 1);
            c->sw_order_off = 4 * (int64_t)nsub;
            int32_t h[2] = {0, 0};
            FEDD_HIP(hipMemcpyAsync(h, n_int, sizeof(h), hipMemcpyDeviceToHost, c->stream));
            FEDD_HIP(hipStreamSynchronize(c->stream));
            if (split) c->sw_nint = h[0];
            c->sw_nconf = h[1];
            // batch table of k_apply_bt: when every subdomain conforms (structured meshes) and the inverses are shared
            c->sw_nbatch = 0;
            c->sw_nbatch_int = 0;
            if (c->sw_dedupe && restricted && c->sw_nconf == nsub && nsub > 0 && c->apply_bt) {
                const int4* rec = (const int4*)(ord + 4 * nsub);
                FEDD_TRY(c->d_sw_btw.ensure((size_t)(5 * nsub + 8)));
                int32_t* start = c->d_sw_btw.p;
                int32_t* scan = start + nsub;
                int32_t* run_first = scan + nsub;       // [nsub + 1]
                int32_t* bflag = run_first + nsub + 1;
                int32_t* bscan = bflag + nsub;
                const int32_t p_cut = split ? (int32_t)c->sw_nint : -1;
                hipLaunchKernelGGL(k_bt_flag, gs, blk, 0, c->stream, rec, (int32_t)nsub, p_cut, start);
                int64_t nruns = 0, nbatch = 0;
                FEDD_TRY(exclusive_scan_i32(c, start, scan, nsub, &nruns));
                hipLaunchKernelGGL(k_bt_run_first, gs, blk, 0, c->stream, (const int32_t*)start, (const int32_t*)scan, (int32_t)nsub,
                                   (int32_t)nruns, run_first);
                hipLaunchKernelGGL(k_bt_batch_flag, gs, blk, 0, c->stream, (const int32_t*)start, (const int32_t*)scan,
                                   (const int32_t*)run_first, (int32_t)nsub, bflag);
                FEDD_TRY(exclusive_scan_i32(c, bflag, bscan, nsub, &nbatch));
                FEDD_TRY(c->d_sw_bt.ensure((size_t)(nbatch * BT_W + 8)));
                hipLaunchKernelGGL(k_bt_fill, gs, blk, 0, c->stream, rec, (const int32_t*)start, (const int32_t*)scan,
                                   (const int32_t*)run_first, (const int32_t*)bflag, (const int32_t*)bscan, (int32_t)nsub, c->d_sw_bt.p);
                c->sw_nbatch = nbatch;
                if (split && c->sw_nint > 0 && c->sw_nint < nsub) {
                    int32_t hb = 0;
                    FEDD_HIP(hipMemcpyAsync(&hb, bscan + c->sw_nint, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
                    FEDD_HIP(hipStreamSynchronize(c->stream));
                    c->sw_nbatch_int = hb;
                } else if (split) {
                    c->sw_nbatch_int = c->sw_nint >= nsub ? nbatch : 0;
                }
            }
        }
    }
    // ---- slab offsets ----
    FEDD_TRY(c->d_inv_ptr.ensure((size_t)nsub + 1));
    hipLaunchKernelGGL(k_slab_sizes, dim3((unsigned)((nsub + 255) / 256)), blk, 0, c->stream, sub_n_inv,
                       (const int32_t*)c->d_sub_nown.p, (int32_t)nsub, restricted, c->d_inv_ptr.p);
    int64_t total = 0;
    FEDD_TRY(exclusive_scan_i64(c, c->d_inv_ptr.p, c->d_inv_ptr.p, nsub, &total));
    c->sw_inv_elems = total;
    FEDD_TRY(c->d_inv.ensure((size_t)total + 16));
    if (c->sw_dedupe)
        hipLaunchKernelGGL(k_fp_share, dim3((unsigned)((nsub + 255) / 256)), blk, 0, c->stream, (const int32_t*)c->d_sw_rep.p,
                           (int32_t)nsub, c->d_inv_ptr.p);
    // ---- extract + invert ----
    FEDD_TRY(c->d_flags.ensure(16));
    int32_t* d_bad = c->d_flags.p + 1;
    FEDD_HIP(hipMemsetAsync(d_bad, 0, sizeof(int32_t), c->stream));
    // size classes of the register-tiled kernel (n <= 16 T); anything larger falls back below
    {
        // with shared inverses only the representatives are inverted: their list instead of one (idle) workgroup per
        // subdomain and size class (ten launches over 166 375 workgroups were 0.8 ms of early exits)
        const int32_t* rep_ids = nullptr;
        int64_t n_inv_wg = nsub;
        if (c->sw_dedupe && c->sw_nrep > 0 && c->sw_nrep < nsub) {
            FEDD_TRY(c->d_sw_replist.ensure((size_t)c->sw_nrep + 1));
            int32_t* cnt = c->d_flags.p + 8;
            FEDD_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t), c->stream));
            hipLaunchKernelGGL(k_rep_list, dim3((unsigned)((nsub + 255) / 256)), blk, 0, c->stream, (const int32_t*)c->d_sw_rep.p,
                               (int32_t)nsub, c->d_sw_replist.p, cnt);
            rep_ids = c->d_sw_replist.p;
            n_inv_wg = c->sw_nrep;
        }
        const dim3 grid((unsigned)n_inv_wg);
#define INV_REG1(T, TA, LO, HI, OWN_LE)                                                                        \
    if (max_n > (LO))                                                                                          \
        hipLaunchKernelGGL((k_invert_reg<T, TA>), grid, blk, 0, c->stream, sub_n_inv,                           \
                           (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p,                   \
                           (const int32_t*)c->d_rowptr.p, (const int32_t*)c->d_colind.p,                       \
                           (const double*)c->d_val.p, n_stored, restricted, (const int64_t*)c->d_inv_ptr.p,      \
                           c->d_inv.p, d_bad, (LO), (HI), p_off, (OWN_LE), rep_ids)
        // restricted combine on a plain system: boxes with at most 32 owned dofs take the variant
        // that drops finished overlap rows from the update, the others the generic one
        const bool rows_only = restricted && !c->merged && c->inv_kind != 2;
#define INV_REG(T, LO, HI)                     \
    if (rows_only) {                           \
        INV_REG1(T, 2, LO, HI, 0);             \
        if (max_own > 32) INV_REG1(T, 0, LO, HI, 32); \
    } else {                                   \
        INV_REG1(T, 0, LO, HI, 0);             \
    }
        if (!c->merged && c->inv_kind == 1) {
            // A/B alternative for plain systems up to 128 dofs: blocks of four pivots on the f64 matrix
            // cores (invert_mfma.hip); measured slower than the scalar-pivot kernel on MI355X
            FEDD_TRY(schwarz_invert_mfma(c, restricted, d_bad, max_n));
        } else {
            INV_REG(2, 0, 32);
            INV_REG(4, 32, 64);
            INV_REG(6, 64, 96);
            INV_REG(7, 96, 112);
            INV_REG(8, 112, 128);
        }
        INV_REG(9, 128, 144);
        INV_REG(10, 144, 160);
#undef INV_REG1
#undef INV_REG
    }
    // subdomains beyond the register-tiled classes (161 .. 256 dofs): batched blocked Gauss-Jordan on the f64 matrix
    // cores (dense.hip).  (Before: an LDS / global-memory scalar sweep at ~1 % of peak: 166 375 subdomains of <= 216 dofs
    // -- 64-node boxes on the 214^3 grid -- took 3.3 s, 27-node boxes with overlap 2, <= 343 -> 247 dofs, 23.7 s.)
    const int n_skip = 160;
    if (max_n > n_skip) FEDD_TRY(schwarz_dense_batched(c, nsub, NMAX, n_skip, p_off, restricted, max_n, d_bad, sub_n_inv));
    int32_t bad = 0;
    FEDD_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    if (c->nranks > 1) {   // one rank's failure is every rank's: nobody is left waiting in the next collective
        double flag = bad ? 1.0 : 0.0;
        FEDD_TRY(c->d_dtmp0.ensure(std::max<size_t>(1, c->d_dtmp0.cap)));
        FEDD_HIP(hipMemcpyAsync(c->d_dtmp0.p, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream));
        FEDD_TRY(allreduce_sum(c, c->d_dtmp0.p, 1));
        FEDD_HIP(hipMemcpyAsync(&flag, c->d_dtmp0.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        bad = flag > 0.0;
    }
    FEDD_CHECK(!bad, "schwarz setup: zero pivot in a local factorisation (matrix singular on a subdomain)");
    if (c->sw_combine == FEDD_COMBINE_AVERAGING) {
        FEDD_TRY(c->d_mult.ensure((size_t)c->n_cols));
        FEDD_HIP(hipMemsetAsync(c->d_mult.p, 0, (size_t)c->n_cols * sizeof(double), c->stream));
        hipLaunchKernelGGL(k_count_mult, dim3((unsigned)nsub), dim3(64), 0, c->stream, (const int32_t*)c->d_sub_n.p,
                           (const int32_t*)c->d_sub_dofs.p, c->d_mult.p);
    }
    FEDD_TRY(c->d_ycol.ensure((size_t)c->n_cols));
    FEDD_HIP(hipGetLastError());
    c->have_schwarz = true;
    timer.stop();
    if (c->sw_two_level) FEDD_TRY(coarse_setup(c));
    return 0;
}

// z_owned = M^-1 r_owned
int schwarz_apply(fedd_ctx* c, const double* d_r_owned, double* d_z_owned, bool r_has_tail) {
    if (c->sw_big_active) return schwarz_apply_big(c, d_r_owned, d_z_owned, r_has_tail);
    const double* r = d_r_owned;
    const bool need_halo = c->n_cols != c->n_rows || !c->halo.peers.empty();   // also a rank that only sends takes part
    // option "halo_overlap": the subdomains without ghost dofs (the first sw_nint places of the order records) are applied
    // while the ghost entries of r travel on a second stream; the others follow when they have arrived
    const bool overlap = need_halo && c->halo_overlap && c->sw_combine == FEDD_COMBINE_RESTRICTED && c->sw_nint > 0;
    double* rbuf = nullptr;
    if (need_halo) {
        if (r_has_tail) {   // the caller's buffer takes the ghost values behind its owned entries
            rbuf = const_cast<double*>(d_r_owned);
        } else {
            FEDD_HIP(hipMemcpyAsync(c->d_xcol.p, d_r_owned, (size_t)c->n_rows * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            rbuf = c->d_xcol.p;
            r = c->d_xcol.p;
        }
        if (!overlap) FEDD_TRY(halo_import(c, rbuf, c->dofs));
    }
    const dim3 blk(256);
    if (c->sw_combine == FEDD_COMBINE_RESTRICTED) {
        // flat streaming kernel while the product park of the largest slab fits 48 KB of LDS
        const size_t park = (((size_t)c->sw_max_size * (size_t)c->sw_max_own + 1) & ~(size_t)1) * sizeof(double);
        // most subdomains share their inverse (schwarz_dedupe found few distinct local matrices): matrix-core kernel
        // (a few thousand subdomains: their slabs stay in the caches and the flat kernel's shorter dependency chain wins)
        const bool shared = c->sw_dedupe && c->sw_nrep * 4 <= c->sw_nsub && c->sw_max_own <= 96 && c->apply_kind != 2 && c->apply_kind != 1 &&
                            (c->sw_nsub >= 4096 || c->apply_kind == 4 || c->apply_kind == 6);
        const int4* records = c->d_sw_order.p ? (const int4*)(c->d_sw_order.p + c->sw_order_off) : nullptr;
        // places [p0, p0 + count) of the order records (all subdomains: p0 = 0, count = sw_nsub)
        auto launch_range = [&](int64_t p0, int64_t count, bool permuted) {
            if (count <= 0) return;
            if (shared) {
                // ranges of whole 64-place chunks (measured on the 214^3 grid, 389017 subdomains: 64: 195 us, 128: 187, 256: 191, 512: 233)
                // (19683 subdomains, the share of one GPU of eight: 64: 40.8 us, 48: 37.0, 32: 34.9, 16: 36.2;
                //  166375 subdomains of 64 nodes, the headline: 32: 157 us, 48: 144, 64: 142.5, 96: 138, 128: 146)
                int span = c->apply_span > 0 ? c->apply_span
                                             : (count >= 256 * 1024 ? 128 : (count >= 128 * 1024 ? 96 : (count >= 48 * 1024 ? 64 : 32)));
                span = std::max(16, (span + 15) / 16 * 16);     // whole 16-place batches
                const int nwg = (int)((count + span - 1) / span);
                // the batch table (every subdomain conforms; "apply_kind" 6 keeps the chunk-record kernel): the places [p0, p0 +
                // count) are the batches [bt0, bt0 + btn) -- the table was cut at the split of the order
                const bool use_bt = c->sw_nbatch > 0 && c->apply_kind != 6 && c->apply_kind != 7 && c->apply_dbg <= 0 &&
                                    (p0 == 0 || p0 == c->sw_nint);
                const int64_t bt0 = p0 == 0 ? 0 : c->sw_nbatch_int;
                const int64_t btn = (p0 == 0 && count == c->sw_nsub) ? c->sw_nbatch
                                                                     : (p0 == 0 ? c->sw_nbatch_int : c->sw_nbatch - c->sw_nbatch_int);
                // one round of workgroups: the loop over the batches is the whole kernel (prologue 2 %), so every slot of the
                // chip -- 256 CUs x the workgroups the kernel's registers allow on one -- takes an equal share of the batches
                // (214^3 cells, 64-node boxes, 10 417 batches: 6 per workgroup 129.6 us, 12: 126.8, 21 = one round: 119.9,
                // 22: 123.8, 25: 136, 32: 162; 107^3 cells, 1 231 batches: 2: 30.1 us, 3 = one round: 25.1, 4: 33.6)
                const int occ_bt = c->sw_max_own <= 32 ? 3 : 2;      // (by the registers of the instantiations below: 134 / 153 and 179 ... 256)
                const int bspan = c->apply_span > 0 ? std::max(1, span / AM_MB)
                                                    : (int)std::max<int64_t>(1, (btn + 256 * occ_bt - 1) / (256 * occ_bt));
                const int nwg_bt = (int)((btn + bspan - 1) / bspan);
                if (use_bt && btn <= 0) return;
#define APPLY_MFMA(RT, KW)                                                                                                   \
    do {                                                                                                                     \
        if (use_bt)                                                                                                          \
            hipLaunchKernelGGL((k_apply_bt<RT, KW>), dim3((unsigned)nwg_bt), blk, 0, c->stream,                               \
                               (const int32_t*)c->d_sw_bt.p + (int64_t)bt0 * BT_W, (const int32_t*)c->d_sub_dofs.p,          \
                               (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, d_z_owned, (int32_t)btn, bspan); \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_apply_mfma<RT, KW>), dim3((unsigned)nwg), blk, 0, c->stream, records + p0,                  \
                               (const int32_t*)c->d_sub_dofs.p, (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, \
                               d_z_owned, (int32_t)count, span);                                                             \
    } while (0)
                // (row tiles, column steps per wave) by the largest subdomain: fewer steps = fewer registers = more waves
                const int64_t mx = c->sw_max_size;
                // 33 ... 64 owned rows: the row tiles split over the waves, B through LDS (k_apply_ms); "apply_kind" 6 = K-split
                if (c->apply_kind == 7 && c->sw_max_own > 32 && c->sw_max_own <= 64 && mx <= 256) {
                    // one workgroup of eight waves per CU: whole rounds of 256 workgroups, at most WS_SPAN places each
                    int span_ws = c->apply_span;
                    if (span_ws <= 0) {
                        const int64_t rounds = (count + 256 * WS_SPAN - 1) / (256 * WS_SPAN);
                        span_ws = (int)((count + 256 * rounds - 1) / (256 * rounds));
                    }
                    span_ws = std::min(WS_SPAN, std::max(16, (span_ws + 15) / 16 * 16));
                    const int nwg_ws = (int)((count + span_ws - 1) / span_ws);
#define APPLY_WS(NK)                                                                                                          \
    hipLaunchKernelGGL((k_apply_ws<NK>), dim3((unsigned)nwg_ws), dim3(512), 0, c->stream, records + p0, (const int32_t*)c->d_sub_dofs.p, \
                       (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, d_z_owned, (int32_t)count, span_ws)
                    if (mx <= 160) APPLY_WS(40);
                    else if (mx <= 192) APPLY_WS(48);
                    else APPLY_WS(64);
#undef APPLY_WS
                    return;
                }
                if (c->sw_max_own <= 32) {
                    if (mx <= 160) APPLY_MFMA(2, 10);
                    else APPLY_MFMA(2, 16);
                } else if (c->sw_max_own <= 64) {
                    if (mx <= 160) APPLY_MFMA(4, 10);
                    else if (mx <= 192 && c->apply_dbg > 0) {
#define APPLY_ABL(A)                                                                                                          \
    case A:                                                                                                                  \
        hipLaunchKernelGGL((k_apply_mfma<4, 12, A>), dim3((unsigned)nwg), blk, 0, c->stream, records + p0,                    \
                           (const int32_t*)c->d_sub_dofs.p, (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r,     \
                           d_z_owned, (int32_t)count, span);                                                                 \
        break;
                        switch (c->apply_dbg) {     // (development: what bounds the kernel; results are wrong by design)
                            APPLY_ABL(1) APPLY_ABL(3) APPLY_ABL(4) APPLY_ABL(7) APPLY_ABL(23) APPLY_ABL(32) APPLY_ABL(39) APPLY_ABL(55)
                            default: break;
                        }
#undef APPLY_ABL
                    } else if (mx <= 192 && use_bt && c->apply_dbg == -1) {     // (development: phase clocks of one wave)
                        hipLaunchKernelGGL((k_apply_bt<4, 12, true>), dim3((unsigned)nwg_bt), blk, 0, c->stream,
                                           (const int32_t*)c->d_sw_bt.p + (int64_t)bt0 * BT_W, (const int32_t*)c->d_sub_dofs.p,
                                           (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, d_z_owned, (int32_t)btn, bspan);
                    } else if (mx <= 192) APPLY_MFMA(4, 12);
                    else APPLY_MFMA(4, 16);
                } else {
                    APPLY_MFMA(6, 16);
                }
#undef APPLY_MFMA
                return;
            }
            const dim3 grid((unsigned)count);
            const int4* perm = permuted ? records : nullptr;
            if ((c->apply_kind == 0 || c->apply_kind == 2 || c->apply_kind == 4) && park <= 48 * 1024) {   // 2 = flat without the compact LDS layout (A/B)
                if (c->sw_max_size <= 128 && c->apply_kind != 2)
                    hipLaunchKernelGGL(k_apply_flat<true>, grid, blk, park, c->stream, (const int32_t*)c->d_sub_n.p,
                                       (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p,
                                       (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, d_z_owned,
                                       (c->sw_max_size + 63) & ~63, perm, (int32_t)p0);
                else
                    hipLaunchKernelGGL(k_apply_flat<false>, grid, blk, park, c->stream, (const int32_t*)c->d_sub_n.p,
                                       (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p,
                                       (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, d_z_owned,
                                       (c->sw_max_size + 63) & ~63, perm, (int32_t)p0);
            } else {
                hipLaunchKernelGGL(k_apply<true>, grid, blk, 0, c->stream, (const int32_t*)c->d_sub_n.p,
                                   (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p,
                                   (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, d_z_owned, perm, (int32_t)p0);
            }
        };
        if (!overlap) {
            ScopedTimer t(c, FEDD_T_SCHWARZ_APPLY);
            launch_range(0, c->sw_nsub, false);
            t.stop();
        } else {
            if (!c->stream2) {
                FEDD_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                FEDD_HIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                FEDD_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            hipStream_t main_stream = c->stream;
            FEDD_HIP(hipEventRecord(c->ev_fork, main_stream));       // the owned part of r is complete here
            FEDD_HIP(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            ScopedTimer t(c, FEDD_T_SCHWARZ_APPLY);                  // (includes whatever of the import stays exposed)
            launch_range(0, c->sw_nint, true);
            c->stream = c->stream2;                                  // the import -- pack, send / receive, unpack -- on stream 2
            const int rc = halo_import(c, rbuf, c->dofs);
            c->stream = main_stream;
            if (rc) return rc;
            FEDD_HIP(hipEventRecord(c->ev_join, c->stream2));
            FEDD_HIP(hipStreamWaitEvent(main_stream, c->ev_join, 0));
            launch_range(c->sw_nint, c->sw_nsub - c->sw_nint, true);
            t.stop();
        }
    } else {
        // Several ranks: what a rank's subdomains contribute to its ghost dofs is dropped (their rows
        // are not stored here, so those entries of the local solutions are not Schwarz corrections),
        // and the multiplicity counts this rank's subdomains only; no exchange is needed.
        double* z = c->d_ycol.p;
        FEDD_HIP(hipMemsetAsync(z, 0, (size_t)c->n_cols * sizeof(double), c->stream));
        {
            ScopedTimer t(c, FEDD_T_SCHWARZ_APPLY);
            hipLaunchKernelGGL(k_apply<false>, dim3((unsigned)c->sw_nsub), blk, 0, c->stream, (const int32_t*)c->d_sub_n.p,
                               (const int32_t*)c->d_sub_nown.p, (const int32_t*)c->d_sub_dofs.p,
                               (const int64_t*)c->d_inv_ptr.p, (const double*)c->d_inv.p, r, z, (const int4*)nullptr, 0);
            t.stop();
        }
        if (c->sw_combine == FEDD_COMBINE_AVERAGING)
            hipLaunchKernelGGL(k_div, dim3((unsigned)((c->n_rows + 255) / 256)), blk, 0, c->stream, z,
                               (const double*)c->d_mult.p, c->n_rows);
        FEDD_HIP(hipMemcpyAsync(d_z_owned, z, (size_t)c->n_rows * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->have_coarse) FEDD_TRY(coarse_apply_add(c, d_r_owned, d_z_owned));
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd
