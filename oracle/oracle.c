/* oracle.c -- CPU restatement of the 3D P1 Laplace hot path in plain C (+OpenMP).
 * TEST INFRASTRUCTURE ONLY: used by tests/ (validated against oracle/fedd_oracle.py) and by
 * bench.py's cpu_baseline leg (kind "port").  Never linked into or called from the product path.
 * PARITY STATUS: "parity unpinned" -- the reference has no golden vectors and cannot be built
 * here (see fedd_oracle.py header); pinned by the analytic known-answer tests through the numpy
 * oracle it is checked against.
 *
 * Follows, loop for loop where the reference code exists:
 *   MeshStructured::buildMesh3D P1          feddlib/core/Mesh/MeshStructured_def.hpp:703-806
 *   setStructuredMeshFlags(1) 3D            feddlib/core/Mesh/MeshStructured_def.hpp:3136-3167
 *   FE::assemblyLaplace                     feddlib/core/FE/FE_def.hpp:604-667
 *   FE::buildTransformation / applyBTinv    feddlib/core/FE/FE_def.hpp:5342-5357 / 83-96
 *   SmallMatrix::computeInverse             feddlib/core/General/SmallMatrix.hpp:306-357
 *   Matrix::insertGlobalValues/fillComplete feddlib/core/LinearAlgebra/Matrix_def.hpp:88-92,192-199
 *   FE::assemblyRHS                         feddlib/core/FE/FE_def.hpp:4694-4766
 *   BCBuilder::setSystem / setRHS           feddlib/core/General/BCBuilder_def.hpp:589-707, 93-170
 * and, for the Trilinos half that is not in the tree, the published algorithms with the
 * reference's parameters (laplace/parametersSolver.xml:5-15, parametersPrec.xml:10-61):
 * right-preconditioned restarted GMRES with classical Gram-Schmidt + DGKS re-orthogonalisation,
 * one-level restricted additive Schwarz, overlap 1, exact local solves, subdomains = the
 * normative node boxes of fedd_oracle.schwarz_bins.
 * Threading: the element loop runs over z-slabs of cells in two colours (even/odd slabs touch
 * disjoint node layers), the stand-in for the reference's one-MPI-rank-per-block parallelism. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return (double)clock() / CLOCKS_PER_SEC;
#endif
}

/* ---- dynamic-insert CSR rows (what Tpetra's dynamic profile does before fillComplete) ---- */
typedef struct { int n, cap; int64_t* col; double* val; } Row;

static void row_insert(Row* r, int nent, const int64_t* cols, const double* vals) {
    if (r->n + nent > r->cap) {
        r->cap = (r->n + nent) * 2 + 8;
        r->col = (int64_t*)realloc(r->col, (size_t)r->cap * sizeof(int64_t));
        r->val = (double*)realloc(r->val, (size_t)r->cap * sizeof(double));
    }
    memcpy(r->col + r->n, cols, (size_t)nent * sizeof(int64_t));
    memcpy(r->val + r->n, vals, (size_t)nent * sizeof(double));
    r->n += nent;
}

/* fillComplete: sort by column, sum duplicates, keep structural zeros */
static int row_compress(Row* r) {
    for (int i = 1; i < r->n; ++i) { /* insertion sort: rows are short */
        int64_t c = r->col[i]; double v = r->val[i]; int j = i - 1;
        while (j >= 0 && r->col[j] > c) { r->col[j + 1] = r->col[j]; r->val[j + 1] = r->val[j]; --j; }
        r->col[j + 1] = c; r->val[j + 1] = v;
    }
    int m = 0;
    for (int i = 0; i < r->n; ++i) {
        if (m > 0 && r->col[m - 1] == r->col[i]) r->val[m - 1] += r->val[i];
        else { r->col[m] = r->col[i]; r->val[m] = r->val[i]; ++m; }
    }
    r->n = m;
    return m;
}

static const int KUHN[6][4][3] = {
    {{1,0,0},{0,0,0},{1,0,1},{1,1,1}}, {{0,0,1},{0,0,0},{1,0,1},{1,1,1}}, {{1,0,0},{0,0,0},{1,1,0},{1,1,1}},
    {{0,0,0},{0,1,0},{1,1,0},{1,1,1}}, {{0,0,0},{0,1,0},{0,1,1},{1,1,1}}, {{0,0,0},{0,0,1},{0,1,1},{1,1,1}}};

typedef struct {
    int M; int64_t n, nnz;
    double* xyz; int* flag;
    int64_t* rowptr; int32_t* col; double* val; double* rhs;
} Problem;

static void problem_free(Problem* p) { free(p->xyz); free(p->flag); free(p->rowptr); free(p->col); free(p->val); free(p->rhs); }

/* mesh + assembly + rhs + Dirichlet; times[0]=mesh, [1]=assembly(matrix+rhs+fillComplete), [2]=bc */
static void build_problem(int M, Problem* P, double* times) {
    const int n1 = M + 1;
    const int64_t n = (int64_t)n1 * n1 * n1;
    const double eps = 2.220446049250313e-16, h = 1.0 / (M * 1), H = 1.0; /* length/(M*N), length/N with N = 1 */
    double t0 = now();
    double* xyz = (double*)malloc((size_t)n * 3 * sizeof(double));
    int* flag = (int*)calloc((size_t)n, sizeof(int));
    #pragma omp parallel for schedule(static)
    for (int t = 0; t < n1; ++t)
        for (int s = 0; s < n1; ++s)
            for (int r = 0; r < n1; ++r) {
                int64_t id = r + (int64_t)n1 * (s + (int64_t)n1 * t);
                double p[3] = {r * h + 0 * H, s * h + 0 * H, t * h + 0 * H};
                for (int d = 0; d < 3; ++d) if (p[d] < eps && p[d] > -eps) p[d] = 0.0;
                for (int d = 0; d < 3; ++d) xyz[id * 3 + d] = p[d];
                int f = 0;
                for (int d = 0; d < 3; ++d) if (p[d] > 1.0 - eps || p[d] < 0.0 + eps) f = 1;
                const double tol = 1e-12; /* setStructuredMeshFlags(1) */
                if (p[0] < tol) f = 2;
                int in = p[0] > tol;
                if (in && p[2] < tol) f = 1;
                if (in && p[2] > 1.0 - tol) f = 1;
                if (in && p[1] < tol) f = 1;
                if (in && p[1] > 1.0 - tol) f = 1;
                if (p[0] > 1.0 - tol && p[1] > tol && p[1] < 1.0 - tol && p[2] > tol && p[2] < 1.0 - tol) f = 3;
                flag[id] = f;
            }
    times[0] = now() - t0;

    t0 = now();
    Row* rows = (Row*)calloc((size_t)n, sizeof(Row));
    double* rhs = (double*)calloc((size_t)n, sizeof(double));
    /* reference tables for P1, degree 1: one point, w = 1/6, dPhi = (-1,-1,-1),(1,0,0),(0,1,0),(0,0,1), phi = 1/4 */
    const double w = 1.0 / 6.0;
    const double dPhi[4][3] = {{-1,-1,-1},{1,0,0},{0,1,0},{0,0,1}};
    const double phi[4] = {0.25, 0.25, 0.25, 0.25};
    const double fval = 1.0; /* oneFunc */
    for (int colour = 0; colour < 2; ++colour) {
        #pragma omp parallel for schedule(dynamic, 1)
        for (int t = colour; t < M; t += 2)
            for (int s = 0; s < M; ++s)
                for (int r = 0; r < M; ++r)
                    for (int k = 0; k < 6; ++k) {
                        int64_t nd[4];
                        for (int v = 0; v < 4; ++v)
                            nd[v] = (r + KUHN[k][v][0]) + (int64_t)n1 * ((s + KUHN[k][v][1]) + (int64_t)n1 * (t + KUHN[k][v][2]));
                        double B[3][3], Bi[3][3];
                        for (int j = 0; j < 3; ++j)
                            for (int i = 0; i < 3; ++i) B[i][j] = xyz[nd[j + 1] * 3 + i] - xyz[nd[0] * 3 + i];
                        double det = B[0][0]*B[1][1]*B[2][2] + B[0][1]*B[1][2]*B[2][0] + B[0][2]*B[1][0]*B[2][1]
                                   - B[2][0]*B[1][1]*B[0][2] - B[2][1]*B[1][2]*B[0][0] - B[2][2]*B[1][0]*B[0][1];
                        Bi[0][0] = (B[1][1]*B[2][2] - B[1][2]*B[2][1]) / det; Bi[0][1] = (B[0][2]*B[2][1] - B[0][1]*B[2][2]) / det;
                        Bi[0][2] = (B[0][1]*B[1][2] - B[0][2]*B[1][1]) / det; Bi[1][0] = (B[1][2]*B[2][0] - B[1][0]*B[2][2]) / det;
                        Bi[1][1] = (B[0][0]*B[2][2] - B[0][2]*B[2][0]) / det; Bi[1][2] = (B[0][2]*B[1][0] - B[0][0]*B[1][2]) / det;
                        Bi[2][0] = (B[1][0]*B[2][1] - B[1][1]*B[2][0]) / det; Bi[2][1] = (B[0][1]*B[2][0] - B[0][0]*B[2][1]) / det;
                        Bi[2][2] = (B[0][0]*B[1][1] - B[0][1]*B[1][0]) / det;
                        double absdet = fabs(det), G[4][3];
                        for (int i = 0; i < 4; ++i)
                            for (int d1 = 0; d1 < 3; ++d1) {
                                G[i][d1] = 0.0;
                                for (int d2 = 0; d2 < 3; ++d2) G[i][d1] += dPhi[i][d2] * Bi[d2][d1];
                            }
                        for (int i = 0; i < 4; ++i) {
                            double value[4];
                            for (int j = 0; j < 4; ++j) {
                                value[j] = 0.0;
                                for (int d = 0; d < 3; ++d) value[j] += w * G[i][d] * G[j][d];
                                value[j] *= absdet;
                            }
                            row_insert(&rows[nd[i]], 4, nd, value);           /* insertGlobalValues */
                            double v = w * phi[i];                            /* assemblyRHS, deg 1 */
                            v *= absdet * fval;
                            rhs[nd[i]] += v;
                        }
                    }
    }
    int64_t* rowptr = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
    #pragma omp parallel for schedule(static, 1024)
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] = row_compress(&rows[i]);        /* fillComplete */
    rowptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
    const int64_t nnz = rowptr[n];
    int32_t* col = (int32_t*)malloc((size_t)nnz * sizeof(int32_t));
    double* val = (double*)malloc((size_t)nnz * sizeof(double));
    #pragma omp parallel for schedule(static, 1024)
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < rows[i].n; ++k) { col[rowptr[i] + k] = (int32_t)rows[i].col[k]; val[rowptr[i] + k] = rows[i].val[k]; }
        free(rows[i].col); free(rows[i].val);
    }
    free(rows);
    times[1] = now() - t0;

    t0 = now();
    #pragma omp parallel for schedule(static, 1024)
    for (int64_t i = 0; i < n; ++i)
        if (flag[i] == 1 || flag[i] == 2 || flag[i] == 3) {                         /* setLocalRowOne + setRHS */
            for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) val[p] = col[p] == i ? 1.0 : 0.0;
            rhs[i] = 0.0;
        }
    times[2] = now() - t0;
    P->M = M; P->n = n; P->nnz = nnz; P->xyz = xyz; P->flag = flag; P->rowptr = rowptr; P->col = col; P->val = val; P->rhs = rhs;
}

/* ---- restricted additive Schwarz ---- */
typedef struct { int n, no; int32_t* dof; double* inv; /* [no][n] row-major: owned rows of A_i^-1 */ } Sub;
typedef struct { int64_t nsub; Sub* s; int max_n; } Ras;

static int cmp_i32(const void* a, const void* b) { int32_t x = *(const int32_t*)a, y = *(const int32_t*)b; return (x > y) - (x < y); }

static void ras_setup(const Problem* P, int target, Ras* R) {
    const int64_t n = P->n;
    /* normative boxes (fedd_oracle.schwarz_bins): bounding box of the nodes, g_d = ceil(L_d/s - 1e-9) made odd,
       boundary nodes to the centre side */
    double lo[3] = {1e300,1e300,1e300}, hi[3] = {-1e300,-1e300,-1e300};
    for (int64_t i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) { double v = P->xyz[i*3+d]; if (v < lo[d]) lo[d] = v; if (v > hi[d]) hi[d] = v; }
    double L[3], V = 1.0; for (int d = 0; d < 3; ++d) { L[d] = hi[d] - lo[d]; V *= L[d] > 0 ? L[d] : 1.0; }
    double s = pow(V * target / (double)n, 1.0 / 3.0);
    int g[3]; double wd[3]; int64_t nraw = 1;
    for (int d = 0; d < 3; ++d) { double Lp = L[d] > 0 ? L[d] : 1.0; g[d] = (int)ceil(Lp / s - 1e-9); if (g[d] < 1 || !(L[d] > 0)) g[d] = 1; if (g[d] % 2 == 0) ++g[d]; wd[d] = Lp / g[d]; nraw *= g[d]; }
    int32_t* bin = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    int32_t* cnt = (int32_t*)calloc((size_t)nraw + 1, sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) {
        int64_t b = 0, mul = 1;
        for (int d = 0; d < 3; ++d) { double t = (P->xyz[i*3+d] - lo[d]) / wd[d], kb = floor(t + 0.5); int ix = (int)floor(t); if (fabs(t - kb) <= 1e-9) ix = 2 * (int)kb <= g[d] - 1 ? (int)kb : (int)kb - 1; if (ix > g[d]-1) ix = g[d]-1; if (ix < 0) ix = 0; b += mul * ix; mul *= g[d]; }
        bin[i] = (int32_t)b; cnt[b + 1]++;
    }
    int32_t* cid = (int32_t*)malloc((size_t)nraw * sizeof(int32_t)); int64_t nsub = 0;
    for (int64_t b = 0; b < nraw; ++b) cid[b] = cnt[b + 1] > 0 ? (int32_t)nsub++ : -1;
    int32_t* ptr = (int32_t*)calloc((size_t)nsub + 1, sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) { bin[i] = cid[bin[i]]; ptr[bin[i] + 1]++; }
    for (int64_t b = 0; b < nsub; ++b) ptr[b + 1] += ptr[b];
    int32_t* nodes = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    int32_t* cur = (int32_t*)malloc((size_t)nsub * sizeof(int32_t)); memcpy(cur, ptr, (size_t)nsub * sizeof(int32_t));
    for (int64_t i = 0; i < n; ++i) nodes[cur[bin[i]]++] = (int32_t)i;       /* ascending within a box */
    free(cur); free(cnt); free(cid);
    R->nsub = nsub; R->s = (Sub*)calloc((size_t)nsub, sizeof(Sub)); int max_n = 0;
    #pragma omp parallel for schedule(dynamic, 16) reduction(max:max_n)
    for (int64_t b = 0; b < nsub; ++b) {
        const int no = ptr[b + 1] - ptr[b];
        const int32_t* own = nodes + ptr[b];
        int cap = 64, ne = 0; int32_t* ext = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
        for (int k = 0; k < no; ++k)                                          /* one graph layer */
            for (int64_t p = P->rowptr[own[k]]; p < P->rowptr[own[k] + 1]; ++p) {
                int32_t c = P->col[p];
                if (bin[c] == b) continue;
                if (ne == cap) { cap *= 2; ext = (int32_t*)realloc(ext, (size_t)cap * sizeof(int32_t)); }
                ext[ne++] = c;
            }
        qsort(ext, (size_t)ne, sizeof(int32_t), cmp_i32);
        int m = 0; for (int k = 0; k < ne; ++k) if (m == 0 || ext[m - 1] != ext[k]) ext[m++] = ext[k];
        const int nn = no + m;
        int32_t* dof = (int32_t*)malloc((size_t)nn * sizeof(int32_t));
        memcpy(dof, own, (size_t)no * sizeof(int32_t)); memcpy(dof + no, ext, (size_t)m * sizeof(int32_t)); free(ext);
        double* A = (double*)calloc((size_t)nn * nn, sizeof(double));
        for (int r = 0; r < nn; ++r)
            for (int64_t p = P->rowptr[dof[r]]; p < P->rowptr[dof[r] + 1]; ++p) {
                int32_t c = P->col[p];
                int32_t* f = (int32_t*)bsearch(&c, dof, (size_t)no, sizeof(int32_t), cmp_i32);
                if (!f) f = (int32_t*)bsearch(&c, dof + no, (size_t)m, sizeof(int32_t), cmp_i32);
                if (f) A[(size_t)r * nn + (f - dof)] = P->val[p];
            }
        for (int k = 0; k < nn; ++k) {                                       /* Gauss-Jordan, no pivoting */
            const double pinv = 1.0 / A[(size_t)k * nn + k];
            for (int j = 0; j < nn; ++j) A[(size_t)k * nn + j] *= pinv;
            A[(size_t)k * nn + k] = pinv;
            for (int i = 0; i < nn; ++i) {
                if (i == k) continue;
                const double f = A[(size_t)i * nn + k];
                A[(size_t)i * nn + k] = 0.0;
                for (int j = 0; j < nn; ++j) A[(size_t)i * nn + j] -= f * A[(size_t)k * nn + j];
            }
        }
        double* inv = (double*)malloc((size_t)no * nn * sizeof(double));
        memcpy(inv, A, (size_t)no * nn * sizeof(double)); free(A);
        R->s[b].n = nn; R->s[b].no = no; R->s[b].dof = dof; R->s[b].inv = inv;
        if (nn > max_n) max_n = nn;
    }
    R->max_n = max_n;
    free(bin); free(ptr); free(nodes);
}

static void ras_apply(const Ras* R, const double* r, double* z) {
    #pragma omp parallel for schedule(dynamic, 64)
    for (int64_t b = 0; b < R->nsub; ++b) {
        const Sub* s = &R->s[b];
        double rs[512];
        for (int c = 0; c < s->n; ++c) rs[c] = r[s->dof[c]];
        for (int i = 0; i < s->no; ++i) {
            double acc = 0.0; const double* row = s->inv + (size_t)i * s->n;
            for (int c = 0; c < s->n; ++c) acc += row[c] * rs[c];
            z[s->dof[i]] = acc;
        }
    }
}

static void ras_free(Ras* R) { for (int64_t b = 0; b < R->nsub; ++b) { free(R->s[b].dof); free(R->s[b].inv); } free(R->s); }

static void spmv(const Problem* P, const double* x, double* y) {
    #pragma omp parallel for schedule(static, 2048)
    for (int64_t i = 0; i < P->n; ++i) {
        double s = 0.0;
        for (int64_t p = P->rowptr[i]; p < P->rowptr[i + 1]; ++p) s += P->val[p] * x[P->col[p]];
        y[i] = s;
    }
}

static double dot(const double* a, const double* b, int64_t n) {
    double s = 0.0;
    #pragma omp parallel for reduction(+:s) schedule(static)
    for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* h[c] = V_c . w for c <= j, one sweep over row blocks (memory-bound like the GPU multi-dot) */
static void multidot(const double* V, const double* w, int64_t n, int ncols, double* h) {
    for (int c = 0; c < ncols; ++c) h[c] = 0.0;
    #pragma omp parallel
    {
        double loc[1024];
        for (int c = 0; c < ncols; ++c) loc[c] = 0.0;
        #pragma omp for schedule(static) nowait
        for (int64_t blk = 0; blk < (n + 2047) / 2048; ++blk) {
            const int64_t i0 = blk * 2048, i1 = i0 + 2048 < n ? i0 + 2048 : n;
            for (int c = 0; c < ncols; ++c) {
                const double* v = V + (size_t)c * n; double s = 0.0;
                for (int64_t i = i0; i < i1; ++i) s += v[i] * w[i];
                loc[c] += s;
            }
        }
        #pragma omp critical
        for (int c = 0; c < ncols; ++c) h[c] += loc[c];
    }
}

/* right-preconditioned GMRES(m), CGS + DGKS second pass, Givens; returns iterations */
static double g_t_op = 0.0, g_t_ortho = 0.0;   /* seconds inside the operator (RAS + SpMV) / inside Gram-Schmidt of the last gmres() */
static int gmres(const Problem* P, const Ras* R, const double* b, double* x, double rtol, int max_it, int restart, double* relres_out) {
    g_t_op = g_t_ortho = 0.0;
    const int64_t n = P->n; const int m = restart < max_it ? restart : max_it;
    double* V = (double*)malloc((size_t)(m + 1) * n * sizeof(double));
    double* w = (double*)malloc((size_t)n * sizeof(double)), *z = (double*)malloc((size_t)n * sizeof(double)), *r = (double*)malloc((size_t)n * sizeof(double));
    double* H = (double*)calloc((size_t)(m + 1) * m, sizeof(double)), *cs = (double*)calloc(m, sizeof(double)), *sn = (double*)calloc(m, sizeof(double));
    double* g = (double*)calloc(m + 1, sizeof(double)), *h = (double*)calloc(m + 2, sizeof(double)), *h2 = (double*)calloc(m + 2, sizeof(double)), *y = (double*)calloc(m, sizeof(double));
    memset(x, 0, (size_t)n * sizeof(double)); memcpy(r, b, (size_t)n * sizeof(double));
    const double beta0 = sqrt(dot(r, r, n)); int its = 0, done = 0; double relres = 1.0;
    while (!done && its < max_it && beta0 > 0) {
        const double beta = sqrt(dot(r, r, n));
        #pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) V[i] = r[i] / beta;
        memset(g, 0, (size_t)(m + 1) * sizeof(double)); g[0] = beta; int k = 0;
        for (int j = 0; j < m && its < max_it; ++j) {
            const double* vj = V + (size_t)j * n;
            double tt = now();
            if (R) { ras_apply(R, vj, z); spmv(P, z, w); } else spmv(P, vj, w);
            g_t_op += now() - tt; tt = now();
            const double n0 = dot(w, w, n);
            multidot(V, w, n, j + 1, h);
            #pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) { double v = w[i]; for (int c = 0; c <= j; ++c) v -= h[c] * V[(size_t)c * n + i]; w[i] = v; }
            double n1 = dot(w, w, n);
            if (n1 < 0.5 * n0) {                                             /* DGKS */
                multidot(V, w, n, j + 1, h2);
                #pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < n; ++i) { double v = w[i]; for (int c = 0; c <= j; ++c) v -= h2[c] * V[(size_t)c * n + i]; w[i] = v; }
                for (int c = 0; c <= j; ++c) h[c] += h2[c];
                n1 = dot(w, w, n);
            }
            g_t_ortho += now() - tt;
            const double hn = sqrt(n1); double* Hj = H + (size_t)j * (m + 1);
            for (int c = 0; c <= j; ++c) Hj[c] = h[c];
            Hj[j + 1] = hn;
            for (int i = 0; i < j; ++i) { double t = cs[i]*Hj[i] + sn[i]*Hj[i+1]; Hj[i+1] = -sn[i]*Hj[i] + cs[i]*Hj[i+1]; Hj[i] = t; }
            const double d = hypot(Hj[j], Hj[j + 1]); cs[j] = Hj[j] / d; sn[j] = Hj[j + 1] / d; Hj[j] = d; Hj[j + 1] = 0.0;
            g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
            if (hn > 0) {
                #pragma omp parallel for schedule(static)
                for (int64_t i = 0; i < n; ++i) V[(size_t)(j + 1) * n + i] = w[i] / hn;
            }
            ++its; k = j + 1; relres = fabs(g[j + 1]) / beta0;
            if (relres <= rtol || !(hn > 0)) { done = 1; break; }
        }
        for (int i = k - 1; i >= 0; --i) { double s = g[i]; for (int c = i + 1; c < k; ++c) s -= H[(size_t)c * (m + 1) + i] * y[c]; y[i] = s / H[(size_t)i * (m + 1) + i]; }
        #pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) { double v = 0.0; for (int c = 0; c < k; ++c) v += y[c] * V[(size_t)c * n + i]; w[i] = v; }
        if (R) { ras_apply(R, w, z); } else memcpy(z, w, (size_t)n * sizeof(double));
        #pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) x[i] += z[i];
        if (!done && its < max_it) { spmv(P, x, r);
            #pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < n; ++i) r[i] = b[i] - r[i]; }
    }
    free(V); free(w); free(z); free(r); free(H); free(cs); free(sn); free(g); free(h); free(h2); free(y);
    *relres_out = relres; return its;
}

/* whole driver: feddlib/problems/tests/laplace/main.cpp:199-208 with 3D / P1 / structured / H/h = M.
 * times[0..6] = mesh, assemble, bc, prec setup, gmres, of which operator, of which Gram-Schmidt (seconds).  x (nullable) [(M+1)^3].
 * csr outputs (nullable): rowptr[n+1], col[nnz], val[nnz] of the Dirichlet-modified matrix; rhs[n]. */
int oracle_laplace3d(int M, int target, double rtol, int restart, int max_it, int use_prec, double* times, int* its,
                     double* relres, int* threads, int64_t* nnz_out, int64_t* nsub_out, int* max_n_out, double* x,
                     int64_t* rowptr, int32_t* col, double* val, double* rhs) {
    Problem P; Ras R; memset(&R, 0, sizeof(R));
#ifdef _OPENMP
    if (*threads > 0) omp_set_num_threads(*threads);
#endif
    build_problem(M, &P, times);
    double t0 = now();
    if (use_prec) ras_setup(&P, target, &R);
    times[3] = now() - t0;
    double* xs = x ? x : (double*)malloc((size_t)P.n * sizeof(double));
    t0 = now();
    *its = gmres(&P, use_prec ? &R : NULL, P.rhs, xs, rtol, max_it, restart, relres);
    times[4] = now() - t0;
    times[5] = g_t_op; times[6] = g_t_ortho;   /* split of times[4]: operator applications / Gram-Schmidt (the rest: basis scaling, update of x) */
#ifdef _OPENMP
    *threads = omp_get_max_threads();
#else
    *threads = 1;
#endif
    if (nnz_out) *nnz_out = P.nnz;
    if (nsub_out) *nsub_out = R.nsub;
    if (max_n_out) *max_n_out = R.max_n;
    if (rowptr) memcpy(rowptr, P.rowptr, (size_t)(P.n + 1) * sizeof(int64_t));
    if (col) memcpy(col, P.col, (size_t)P.nnz * sizeof(int32_t));
    if (val) memcpy(val, P.val, (size_t)P.nnz * sizeof(double));
    if (rhs) memcpy(rhs, P.rhs, (size_t)P.n * sizeof(double));
    if (!x) free(xs);
    if (use_prec) ras_free(&R);
    problem_free(&P);
    return 0;
}
