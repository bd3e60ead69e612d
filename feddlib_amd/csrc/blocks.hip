// Block systems on the device: stored blocks and their merge into one monolithic CSR matrix.
//
// Replaces BlockMatrix::merge / mergeBlockNew and BlockMap::merge
// (feddlib/core/LinearAlgebra/BlockMatrix_def.hpp:119-148, 212-287; BlockMap_def.hpp:55-80), which
// re-insert every row of every block into a new Tpetra matrix with the global ids of block k
// shifted by the cumulated (maxAllGlobalIndex + 1) of the blocks before it.  Here: row counts ->
// scan -> one fill kernel; column ids of the second block column are shifted by the first block's
// column count, so concatenated rows stay sorted.
// Layout of the 2 x 2 system (Stokes_def.hpp:47-138):  [ A  B^T ; B  C ],  C optional.
#include "fedd_internal.hpp"
#include <algorithm>

namespace fedd {
namespace {

__global__ void k_scale(double* __restrict__ v, int64_t n, double a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= a;
}

struct CsrView {
    const int32_t* rowptr;
    const int32_t* colind;
    const double* val;
};

__global__ void k_merge_count(CsrView A, CsrView BT, CsrView B, CsrView C, int32_t nA, int32_t nB,
                              int32_t* __restrict__ cnt) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nA + nB) return;
    int32_t n = 0;
    if (r < nA) {
        n = A.rowptr[r + 1] - A.rowptr[r];
        if (BT.rowptr) n += BT.rowptr[r + 1] - BT.rowptr[r];
    } else {
        const int32_t i = r - nA;
        if (B.rowptr) n += B.rowptr[i + 1] - B.rowptr[i];
        if (C.rowptr) n += C.rowptr[i + 1] - C.rowptr[i];
    }
    cnt[r] = n;
}

__global__ void k_merge_fill(CsrView A, CsrView BT, CsrView B, CsrView C, int32_t nA, int32_t nB, int32_t col_off,
                             const int32_t* __restrict__ rowptr, int32_t* __restrict__ colind,
                             double* __restrict__ val) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nA + nB) return;
    int32_t p = rowptr[r];
    const CsrView& L = r < nA ? A : B;   // block in the first block column
    const CsrView& R = r < nA ? BT : C;  // block in the second block column
    const int32_t i = r < nA ? r : r - nA;
    if (L.rowptr)
        for (int32_t q = L.rowptr[i]; q < L.rowptr[i + 1]; ++q, ++p) {
            colind[p] = L.colind[q];
            val[p] = L.val[q];
        }
    if (R.rowptr)
        for (int32_t q = R.rowptr[i]; q < R.rowptr[i + 1]; ++q, ++p) {
            colind[p] = R.colind[q] + col_off;
            val[p] = R.val[q];
        }
}

__global__ void k_dof_node(int32_t nA, int32_t nB, int dofsA, int32_t* __restrict__ dof_node) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nA + nB) return;
    dof_node[r] = r < nA ? r / dofsA : r - nA;  // pressure dof i sits on vertex node i (P1 nodes come first)
}

CsrView view(const DevCsr* m) {
    CsrView v{nullptr, nullptr, nullptr};
    if (m && m->valid) {
        v.rowptr = m->rowptr.p;
        v.colind = m->colind.p;
        v.val = m->val.p;
    }
    return v;
}

}  // namespace

int matrix_store(fedd_ctx* c, int slot) {
    DevCsr& m = c->aux[slot];
    FEDD_TRY(m.rowptr.ensure((size_t)c->n_rows + 1));
    FEDD_TRY(m.colind.ensure((size_t)c->nnz));
    FEDD_TRY(m.val.ensure((size_t)c->nnz));
    FEDD_HIP(hipMemcpyAsync(m.rowptr.p, c->d_rowptr.p, ((size_t)c->n_rows + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    FEDD_HIP(hipMemcpyAsync(m.colind.p, c->d_colind.p, (size_t)c->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
    FEDD_HIP(hipMemcpyAsync(m.val.p, c->d_val.p, (size_t)c->nnz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    m.n_rows = c->n_rows;
    m.n_cols = c->n_cols;
    m.nnz = c->nnz;
    m.max_row_nnz = c->max_row_nnz;
    m.valid = true;
    return 0;
}

int matrix_scale(fedd_ctx* c, int slot, double alpha) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    double* v = slot < 0 ? c->d_val.p : c->aux[slot].val.p;
    const int64_t n = slot < 0 ? c->nnz : c->aux[slot].nnz;
    if (n > 0) hipLaunchKernelGGL(k_scale, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, v, n, alpha);
    FEDD_HIP(hipGetLastError());
    if (slot < 0) c->have_schwarz = false;
    return 0;
}

int block_merge(fedd_ctx* c, int slot_a, int slot_bt, int slot_b, int slot_c) {
    c->cs_valid = false;   // the solver's compacted SpMV stream follows the matrix values
    FEDD_CHECK(c->nranks == 1, "block merge: one rank only for now");
    const DevCsr* A = &c->aux[slot_a];
    const DevCsr* BT = slot_bt >= 0 ? &c->aux[slot_bt] : nullptr;
    const DevCsr* B = slot_b >= 0 ? &c->aux[slot_b] : nullptr;
    const DevCsr* C = slot_c >= 0 ? &c->aux[slot_c] : nullptr;
    FEDD_CHECK(A->valid, "block merge: block (0,0) is empty");
    FEDD_CHECK((B && B->valid) || (BT && BT->valid), "block merge: needs at least one off-diagonal block");
    const int64_t nA = A->n_rows;
    const int64_t nB = B && B->valid ? B->n_rows : BT->n_cols;
    FEDD_CHECK(!BT || !BT->valid || (BT->n_rows == nA && BT->n_cols == nB), "block merge: B^T has the wrong shape");
    FEDD_CHECK(!B || !B->valid || B->n_cols == A->n_cols, "block merge: B has the wrong shape");
    FEDD_CHECK(!C || !C->valid || (C->n_rows == nB && C->n_cols == nB), "block merge: C has the wrong shape");
    const int64_t n = nA + nB;
    FEDD_CHECK(n < ((int64_t)1 << 31), "block merge: too many rows");
    const int dofsA = (int)(nA / std::max<int64_t>(1, c->n_own));
    FEDD_TRY(c->d_rowptr.ensure((size_t)n + 1));
    const CsrView vA = view(A), vBT = view(BT), vB = view(B), vC = view(C);
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_merge_count, grid, blk, 0, c->stream, vA, vBT, vB, vC, (int32_t)nA, (int32_t)nB, c->d_rowptr.p);
    int32_t mx = 0;
    FEDD_TRY(reduce_max_i32(c, c->d_rowptr.p, n, &mx));
    int64_t nnz = 0;
    FEDD_TRY(exclusive_scan_i32(c, c->d_rowptr.p, c->d_rowptr.p, n, &nnz));
    FEDD_TRY(c->d_colind.ensure((size_t)nnz));
    FEDD_TRY(c->d_val.ensure((size_t)nnz));
    hipLaunchKernelGGL(k_merge_fill, grid, blk, 0, c->stream, vA, vBT, vB, vC, (int32_t)nA, (int32_t)nB, (int32_t)A->n_cols,
                       (const int32_t*)c->d_rowptr.p, c->d_colind.p, c->d_val.p);
    FEDD_TRY(c->d_dof_node.ensure((size_t)n));
    hipLaunchKernelGGL(k_dof_node, grid, blk, 0, c->stream, (int32_t)nA, (int32_t)nB, dofsA, c->d_dof_node.p);
    c->n_rows = n;
    c->n_rows_ext = n;
    c->n_cols = n;
    c->nnz = nnz;
    c->nnz_ext = nnz;
    c->max_row_nnz = mx;
    c->merged = true;
    c->merged_nA = nA;
    c->merged_dofsA = dofsA;
    c->dofs = 1;
    c->block_mode = FEDD_BLOCK_SCALAR;
    FEDD_TRY(c->d_rhs.ensure((size_t)n));
    FEDD_TRY(c->d_x.ensure((size_t)n));
    FEDD_TRY(c->d_xcol.ensure((size_t)n));
    FEDD_TRY(c->d_isdir.ensure((size_t)n));
    FEDD_HIP(hipMemsetAsync(c->d_rhs.p, 0, (size_t)n * sizeof(double), c->stream));
    FEDD_HIP(hipMemsetAsync(c->d_x.p, 0, (size_t)n * sizeof(double), c->stream));
    FEDD_HIP(hipMemsetAsync(c->d_isdir.p, 0, (size_t)n * sizeof(int32_t), c->stream));
    FEDD_HIP(hipGetLastError());
    c->have_schwarz = false;
    c->spmv_rows_ready = false;
    return 0;
}

}  // namespace fedd
