// Halo exchange between the per-GPU subdomains: plan construction (host logic, transport
// agnostic so that it can be exercised on CPU ranks over gloo) and the device-side import
// (pack kernel -> grouped ncclSend/ncclRecv over xGMI -> unpack kernel).
//
// Replaces the Tpetra Import the reference reaches through Xpetra for every SpMV
// (feddlib/core/LinearAlgebra/Matrix_def.hpp:245-254) and FROSch's restriction import;
// FEDDLib's own Import/Export call sites are feddlib/core/LinearAlgebra/MultiVector_def.hpp:258-330.
#include "fedd_internal.hpp"
#include <rccl/rccl.h>
#include <algorithm>
#include <cmath>
#include <unordered_map>

namespace fedd {
namespace {

__global__ void k_pack(const double* __restrict__ x, const int32_t* __restrict__ lid, int64_t n, int dofs,
                       double* __restrict__ buf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dofs) return;
    const int64_t k = i / dofs;
    const int d = (int)(i - k * dofs);
    buf[i] = x[(int64_t)lid[k] * dofs + d];
}

__global__ void k_unpack(double* __restrict__ x, const int32_t* __restrict__ lid, int64_t n, int dofs,
                         const double* __restrict__ buf) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dofs) return;
    const int64_t k = i / dofs;
    const int d = (int)(i - k * dofs);
    x[(int64_t)lid[k] * dofs + d] = buf[i];
}

}  // namespace

// x[ghost dofs] <- owners' values
int halo_import(fedd_ctx* c, double* d_xcol, int dofs) {
    if (c->nranks == 1) return 0;
    HaloPlan& h = c->halo;
    // a rank without ghost nodes may still have to send; only a rank without peers has nothing to do
    if (h.ready && h.peers.empty()) return 0;
    FEDD_CHECK(h.ready && (c->comm || c->cb_exchange),
               "halo import: no exchange plan / transport; call fedd_halo_exchange_setup after fedd_mesh_set");
    const int64_t ns = (int64_t)h.send_lid.size(), nr = (int64_t)h.recv_lid.size();
    ScopedTimer timer(c, FEDD_T_HALO);
    if (ns > 0)
        hipLaunchKernelGGL(k_pack, dim3((unsigned)((ns * dofs + 255) / 256)), dim3(256), 0, c->stream, (const double*)d_xcol,
                           (const int32_t*)h.d_send_lid.p, ns, dofs, h.d_send_buf.p);
    if (c->cb_exchange) {  // host-staged transport (functional tests): same kernels, caller moves the bytes
        c->h_send.resize((size_t)ns * dofs + 1);
        c->h_recv.resize((size_t)nr * dofs + 1);
        if (ns > 0) FEDD_HIP(hipMemcpyAsync(c->h_send.data(), h.d_send_buf.p, (size_t)ns * dofs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        const int rc = c->cb_exchange(c->cb_user, (int)h.peers.size(), h.peers.data(), h.send_ptr.data(), c->h_send.data(),
                                      h.recv_ptr.data(), c->h_recv.data(), dofs);
        FEDD_CHECK(rc == 0, "halo import: the host exchange callback failed (%d)", rc);
        if (nr > 0) {
            FEDD_HIP(hipMemcpyAsync(h.d_recv_buf.p, c->h_recv.data(), (size_t)nr * dofs * sizeof(double), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_unpack, dim3((unsigned)((nr * dofs + 255) / 256)), dim3(256), 0, c->stream, d_xcol,
                               (const int32_t*)h.d_recv_lid.p, nr, dofs, (const double*)h.d_recv_buf.p);
            FEDD_HIP(hipStreamSynchronize(c->stream));
        }
        return 0;
    }
    ncclComm_t comm = (ncclComm_t)c->comm;
    // every call of the group is checked; the group is closed in any case (an open group would swallow the
    // next collective) and the first failure is the one reported
    ncclResult_t bad = ncclGroupStart();
    for (size_t k = 0; k < h.peers.size() && bad == ncclSuccess; ++k) {
        const int64_t s0 = h.send_ptr[k] * dofs, s1 = h.send_ptr[k + 1] * dofs;
        const int64_t r0 = h.recv_ptr[k] * dofs, r1 = h.recv_ptr[k + 1] * dofs;
        if (s1 > s0) bad = ncclSend(h.d_send_buf.p + s0, (size_t)(s1 - s0), ncclDouble, h.peers[k], comm, c->stream);
        if (r1 > r0 && bad == ncclSuccess)
            bad = ncclRecv(h.d_recv_buf.p + r0, (size_t)(r1 - r0), ncclDouble, h.peers[k], comm, c->stream);
    }
    const ncclResult_t r = ncclGroupEnd();
    FEDD_CHECK(bad == ncclSuccess, "halo import: %s", ncclGetErrorString(bad));
    FEDD_CHECK(r == ncclSuccess, "halo import: %s", ncclGetErrorString(r));
    if (nr > 0)
        hipLaunchKernelGGL(k_unpack, dim3((unsigned)((nr * dofs + 255) / 256)), dim3(256), 0, c->stream, d_xcol,
                           (const int32_t*)h.d_recv_lid.p, nr, dofs, (const double*)h.d_recv_buf.p);
    FEDD_HIP(hipGetLastError());
    return 0;
}

}  // namespace fedd

using namespace fedd;

// Step 1 (host): owners of the ghost nodes -> per-owner request lists (ascending global id).
extern "C" int fedd_halo_set_owners(fedd_ctx* c, int64_t n_rep, const int64_t* gid_rep, const int32_t* owner_rep) {
    FEDD_CHECK(c && c->n_node > 0, "fedd_halo_set_owners: call fedd_mesh_set first");
    FEDD_CHECK(n_rep == 0 || (gid_rep && owner_rep), "fedd_halo_set_owners: null array");
    HaloPlan& h = c->halo;
    h.reset();
    const int64_t ng = c->n_node - c->n_own;
    std::unordered_map<int64_t, int32_t> ghost;  // gid -> ghost node id
    ghost.reserve((size_t)ng * 2);
    for (int64_t k = 0; k < ng; ++k) ghost.emplace(c->h_node_gid[c->n_own + k], (int32_t)(c->n_own + k));
    std::vector<int32_t> owner((size_t)ng, -1);
    for (int64_t i = 0; i < n_rep; ++i) {
        auto it = ghost.find(gid_rep[i]);
        if (it == ghost.end()) continue;
        FEDD_CHECK(owner_rep[i] >= 0 && owner_rep[i] < c->nranks && owner_rep[i] != c->rank,
                   "fedd_halo_set_owners: ghost node %lld has owner %d", (long long)gid_rep[i], owner_rep[i]);
        owner[it->second - c->n_own] = owner_rep[i];
    }
    // with row ghosts declared (fedd_mesh_set_rows) only they are imported: the owned rows and the Schwarz
    // subdomains reach no further, the outer ghost layer exists for its elements only
    const int64_t ng_req = c->n_rowg > 0 ? c->n_rowg : ng;
    std::vector<std::pair<int32_t, int64_t>> order;  // (owner, gid)
    order.reserve((size_t)ng_req);
    for (int64_t k = 0; k < ng_req; ++k) {
        FEDD_CHECK(owner[k] >= 0, "fedd_halo_set_owners: no owner given for ghost node %lld", (long long)c->h_node_gid[c->n_own + k]);
        order.emplace_back(owner[k], c->h_node_gid[c->n_own + k]);
    }
    std::sort(order.begin(), order.end());
    h.req_count.assign((size_t)c->nranks, 0);
    h.req_gid.clear();
    h.recv_lid.clear();
    for (auto& pr : order) {
        h.req_count[pr.first] += 1;
        h.req_gid.push_back(pr.second);
        h.recv_lid.push_back(ghost[pr.second]);
    }
    return 0;
}

extern "C" int fedd_halo_requests_sizes(fedd_ctx* c, int64_t* count_to_rank) {
    FEDD_CHECK(c && count_to_rank, "fedd_halo_requests_sizes: null");
    FEDD_CHECK((int)c->halo.req_count.size() == c->nranks, "fedd_halo_requests_sizes: call fedd_halo_set_owners first");
    std::copy(c->halo.req_count.begin(), c->halo.req_count.end(), count_to_rank);
    return 0;
}

extern "C" int fedd_halo_requests_get(fedd_ctx* c, int64_t* gids) {
    FEDD_CHECK(c && (gids || c->halo.req_gid.empty()), "fedd_halo_requests_get: null");
    std::copy(c->halo.req_gid.begin(), c->halo.req_gid.end(), gids);
    return 0;
}

// Step 2 (host): what the other ranks asked of me -> send lists; finalises the plan.
extern "C" int fedd_halo_requests_set(fedd_ctx* c, const int64_t* count_from_rank, const int64_t* gids) {
    FEDD_CHECK(c && count_from_rank, "fedd_halo_requests_set: null");
    HaloPlan& h = c->halo;
    FEDD_CHECK((int)h.req_count.size() == c->nranks, "fedd_halo_requests_set: call fedd_halo_set_owners first");
    std::unordered_map<int64_t, int32_t> own;
    own.reserve((size_t)c->n_own * 2);
    for (int64_t i = 0; i < c->n_own; ++i) own.emplace(c->h_node_gid[i], (int32_t)i);
    h.peers.clear();
    h.send_ptr.assign(1, 0);
    h.recv_ptr.assign(1, 0);
    h.send_lid.clear();
    int64_t off = 0, roff = 0;
    for (int p = 0; p < c->nranks; ++p) {
        const int64_t ns = count_from_rank[p], nr = h.req_count[p];
        if (ns > 0 || nr > 0) {
            FEDD_CHECK(p != c->rank, "fedd_halo_requests_set: a rank cannot request from itself");
            h.peers.push_back(p);
            for (int64_t k = 0; k < ns; ++k) {
                auto it = own.find(gids[off + k]);
                FEDD_CHECK(it != own.end(), "halo plan: rank %d asked rank %d for node %lld which it does not own", p, c->rank,
                           (long long)gids[off + k]);
                h.send_lid.push_back(it->second);
            }
            roff += nr;
            h.send_ptr.push_back((int64_t)h.send_lid.size());
            h.recv_ptr.push_back(roff);
        }
        off += ns;
    }
    if (c->device >= 0) {
        FEDD_HIP(hipSetDevice(c->device));
        FEDD_TRY(h.d_send_lid.ensure(h.send_lid.size()));
        FEDD_TRY(h.d_recv_lid.ensure(h.recv_lid.size()));
        FEDD_TRY(h.d_send_buf.ensure(h.send_lid.size() * MAX_DOFS));
        FEDD_TRY(h.d_recv_buf.ensure(h.recv_lid.size() * MAX_DOFS));
        if (!h.send_lid.empty())
            FEDD_HIP(hipMemcpy(h.d_send_lid.p, h.send_lid.data(), h.send_lid.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        if (!h.recv_lid.empty())
            FEDD_HIP(hipMemcpy(h.d_recv_lid.p, h.recv_lid.data(), h.recv_lid.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    h.ready = true;
    return 0;
}

// Both steps over RCCL (all-gather of the request counts, then grouped send/recv of the lists).
extern "C" int fedd_halo_exchange_setup(fedd_ctx* c) {
    FEDD_CHECK(c && c->device >= 0, "fedd_halo_exchange_setup needs a GPU context");
    if (c->nranks == 1) {
        c->halo.ready = true;
        return 0;
    }
    FEDD_CHECK(c->comm, "fedd_halo_exchange_setup: no communicator");
    HaloPlan& h = c->halo;
    const int R = c->nranks;
    FEDD_CHECK((int)h.req_count.size() == R, "fedd_halo_exchange_setup: call fedd_halo_set_owners first");
    FEDD_HIP(hipSetDevice(c->device));
    ncclComm_t comm = (ncclComm_t)c->comm;
    DevBuf<int64_t> d_cnt, d_all, d_req, d_in;
    FEDD_TRY(d_cnt.ensure((size_t)R));
    FEDD_TRY(d_all.ensure((size_t)R * R));
    FEDD_HIP(hipMemcpy(d_cnt.p, h.req_count.data(), (size_t)R * sizeof(int64_t), hipMemcpyHostToDevice));
    ncclResult_t r = ncclAllGather(d_cnt.p, d_all.p, (size_t)R, ncclInt64, comm, c->stream);
    FEDD_CHECK(r == ncclSuccess, "halo setup all-gather: %s", ncclGetErrorString(r));
    std::vector<int64_t> all((size_t)R * R);
    FEDD_HIP(hipMemcpyAsync(all.data(), d_all.p, all.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    std::vector<int64_t> from((size_t)R);
    int64_t nin = 0;
    for (int p = 0; p < R; ++p) {
        from[p] = all[(size_t)p * R + c->rank];  // what rank p asks of me
        nin += from[p];
    }
    FEDD_TRY(d_req.ensure(h.req_gid.size()));
    FEDD_TRY(d_in.ensure((size_t)nin));
    if (!h.req_gid.empty())
        FEDD_HIP(hipMemcpy(d_req.p, h.req_gid.data(), h.req_gid.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    ncclResult_t bad = ncclGroupStart();
    int64_t so = 0, ro = 0;
    for (int p = 0; p < R && bad == ncclSuccess; ++p) {
        if (h.req_count[p] > 0) bad = ncclSend(d_req.p + so, (size_t)h.req_count[p], ncclInt64, p, comm, c->stream);
        if (from[p] > 0 && bad == ncclSuccess) bad = ncclRecv(d_in.p + ro, (size_t)from[p], ncclInt64, p, comm, c->stream);
        so += h.req_count[p];
        ro += from[p];
    }
    r = ncclGroupEnd();
    FEDD_CHECK(bad == ncclSuccess, "halo setup exchange: %s", ncclGetErrorString(bad));
    FEDD_CHECK(r == ncclSuccess, "halo setup exchange: %s", ncclGetErrorString(r));
    std::vector<int64_t> in((size_t)nin);
    if (nin) FEDD_HIP(hipMemcpyAsync(in.data(), d_in.p, (size_t)nin * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    int rc = fedd_halo_requests_set(c, from.data(), in.data());
    d_cnt.release();
    d_all.release();
    d_req.release();
    d_in.release();
    return rc;
}

// One-rank RCCL self-test: the call shapes of this library's communication steps -- ncclCommInitRank, the grouped
// ncclSend / ncclRecv of halo_import (to and from the only rank there is), ncclAllReduce(sum, f64) in place as in
// allreduce_sum, ncclAllGather(int64) as in fedd_halo_exchange_setup -- on the context's stream, checked against the
// expected values.  Development boxes have one GPU and RCCL refuses several ranks on one device, so this is the part
// of the RCCL path that can run there; the multi-rank logic around it runs over the host-staged transport.
extern "C" int fedd_rccl_selftest(fedd_ctx* c, int n, double* max_abs_err) {
    FEDD_CHECK(c && c->device >= 0, "fedd_rccl_selftest needs a GPU context");
    FEDD_CHECK(n > 0 && max_abs_err, "fedd_rccl_selftest: n %d", n);
    FEDD_HIP(hipSetDevice(c->device));
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    FEDD_CHECK(r == ncclSuccess, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    ncclComm_t comm;
    r = ncclCommInitRank(&comm, 1, id, 0);
    FEDD_CHECK(r == ncclSuccess, "ncclCommInitRank(1 rank): %s", ncclGetErrorString(r));
    DevBuf<double> a, b;
    DevBuf<int64_t> g0, g1;
    int rc = 0;
    std::vector<double> h((size_t)n), out((size_t)n);
    for (int i = 0; i < n; ++i) h[(size_t)i] = 0.5 + 1e-3 * i;
    std::vector<int64_t> hg = {7, 11}, og(2, 0);
    double err = 0.0;
    do {
        if ((rc = a.ensure((size_t)n)) || (rc = b.ensure((size_t)n)) || (rc = g0.ensure(2)) || (rc = g1.ensure(2))) break;
        if (hipMemcpyAsync(a.p, h.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = 1; set_error("selftest: upload"); break; }
        // grouped send / receive (halo_import's shape)
        ncclResult_t bad = ncclGroupStart();
        if (bad == ncclSuccess) bad = ncclSend(a.p, (size_t)n, ncclDouble, 0, comm, c->stream);
        if (bad == ncclSuccess) bad = ncclRecv(b.p, (size_t)n, ncclDouble, 0, comm, c->stream);
        r = ncclGroupEnd();
        if (bad != ncclSuccess || r != ncclSuccess) { rc = 1; set_error("selftest send/recv: %s", ncclGetErrorString(bad != ncclSuccess ? bad : r)); break; }
        // in-place all-reduce (allreduce_sum's shape): one rank -> unchanged
        r = ncclAllReduce(b.p, b.p, (size_t)n, ncclDouble, ncclSum, comm, c->stream);
        if (r != ncclSuccess) { rc = 1; set_error("selftest all-reduce: %s", ncclGetErrorString(r)); break; }
        if (hipMemcpyAsync(g0.p, hg.data(), 16, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = 1; set_error("selftest: upload"); break; }
        r = ncclAllGather(g0.p, g1.p, 2, ncclInt64, comm, c->stream);
        if (r != ncclSuccess) { rc = 1; set_error("selftest all-gather: %s", ncclGetErrorString(r)); break; }
        if (hipMemcpyAsync(out.data(), b.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(og.data(), g1.p, 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { rc = 1; set_error("selftest: download"); break; }
        for (int i = 0; i < n; ++i) err = std::max(err, std::abs(out[(size_t)i] - h[(size_t)i]));
        if (og[0] != 7 || og[1] != 11) err = std::max(err, 1.0);
    } while (false);
    ncclCommDestroy(comm);
    *max_abs_err = err;
    return rc;
}

// The collectives of the N > 1 path on the context's OWN communicator, in the shapes the solver uses them: an in-place
// all-reduce (allreduce_sum), a grouped send / receive with every other rank (halo_import with all peers: the 2 x 2 x 2
// decomposition's centre block) and a ring shift.  bench.py runs it as a pre-flight right after the contexts exist, under a
// watchdog: a misuse of RCCL shows here within seconds and with a message.
extern "C" int fedd_comm_selftest(fedd_ctx* c, int n, double* max_abs_err) {
    FEDD_CHECK(c && c->device >= 0, "fedd_comm_selftest needs a GPU context");
    FEDD_CHECK(n > 0 && max_abs_err, "fedd_comm_selftest: n %d", n);
    *max_abs_err = 0.0;
    if (c->nranks == 1) return 0;
    FEDD_CHECK(c->comm || c->cb_allreduce, "fedd_comm_selftest: no transport (context created without an RCCL id and without host callbacks)");
    FEDD_HIP(hipSetDevice(c->device));
    const int N = c->nranks, me = c->rank;
    double err = 0.0;
    // (1) all-reduce through the solver's own entry
    DevBuf<double> a, sb, rb;
    FEDD_TRY(a.ensure((size_t)n));
    std::vector<double> h((size_t)n), out((size_t)n);
    for (int i = 0; i < n; ++i) h[(size_t)i] = (me + 1) * (1.0 + 1e-3 * i);
    FEDD_HIP(hipMemcpyAsync(a.p, h.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    FEDD_TRY(allreduce_sum(c, a.p, n));
    FEDD_HIP(hipMemcpyAsync(out.data(), a.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    FEDD_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) err = std::max(err, std::abs(out[(size_t)i] - 0.5 * N * (N + 1) * (1.0 + 1e-3 * i)));
    if (c->comm) {
        // (2) grouped send / receive with every other rank, n doubles each: rank p gets (me, p, i) encoded
        ncclComm_t comm = (ncclComm_t)c->comm;
        FEDD_TRY(sb.ensure((size_t)n * N));
        FEDD_TRY(rb.ensure((size_t)n * N));
        std::vector<double> hs((size_t)n * N), hr((size_t)n * N, -1.0);
        for (int p = 0; p < N; ++p)
            for (int i = 0; i < n; ++i) hs[(size_t)p * n + i] = 1000.0 * me + p + 1e-6 * i;
        FEDD_HIP(hipMemcpyAsync(sb.p, hs.data(), hs.size() * 8, hipMemcpyHostToDevice, c->stream));
        FEDD_HIP(hipMemsetAsync(rb.p, 0, (size_t)n * N * 8, c->stream));
        ncclResult_t bad = ncclGroupStart();
        for (int p = 0; p < N && bad == ncclSuccess; ++p) {
            if (p == me) continue;
            bad = ncclSend(sb.p + (size_t)p * n, (size_t)n, ncclDouble, p, comm, c->stream);
            if (bad == ncclSuccess) bad = ncclRecv(rb.p + (size_t)p * n, (size_t)n, ncclDouble, p, comm, c->stream);
        }
        ncclResult_t r = ncclGroupEnd();
        FEDD_CHECK(bad == ncclSuccess && r == ncclSuccess, "fedd_comm_selftest: grouped send/recv: %s", ncclGetErrorString(bad != ncclSuccess ? bad : r));
        FEDD_HIP(hipMemcpyAsync(hr.data(), rb.p, hr.size() * 8, hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        for (int p = 0; p < N; ++p)
            if (p != me)
                for (int i = 0; i < n; ++i) err = std::max(err, std::abs(hr[(size_t)p * n + i] - (1000.0 * p + me + 1e-6 * i)));
        // (3) ring shift
        const int to = (me + 1) % N, from = (me + N - 1) % N;
        bad = ncclGroupStart();
        if (bad == ncclSuccess) bad = ncclSend(sb.p, (size_t)n, ncclDouble, to, comm, c->stream);
        if (bad == ncclSuccess) bad = ncclRecv(rb.p, (size_t)n, ncclDouble, from, comm, c->stream);
        r = ncclGroupEnd();
        FEDD_CHECK(bad == ncclSuccess && r == ncclSuccess, "fedd_comm_selftest: ring shift: %s", ncclGetErrorString(bad != ncclSuccess ? bad : r));
        FEDD_HIP(hipMemcpyAsync(hr.data(), rb.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        FEDD_HIP(hipStreamSynchronize(c->stream));
        for (int i = 0; i < n; ++i) err = std::max(err, std::abs(hr[(size_t)i] - (1000.0 * from + 0 + 1e-6 * i)));
    }
    *max_abs_err = err;
    return 0;
}

// owner rank of a structured-grid node under the lowest-rank rule (mesh_structured.cpp)
extern "C" int fedd_mesh_structured_owner(int dim, const int* decomp, const int* cells, int64_t n, const int64_t* gid,
                                          int32_t* owner_rank) {
    FEDD_CHECK(dim == 2 || dim == 3, "fedd_mesh_structured_owner: dimension must be 2 or 3");
    int64_t P[3] = {1, 1, 1};
    int N[3] = {1, 1, 1}, M[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d) {
        N[d] = decomp[d];
        M[d] = cells[d];
        P[d] = (int64_t)N[d] * M[d] + 1;
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t g = gid[i];
        const int64_t cc[3] = {g % P[0], (g / P[0]) % P[1], g / (P[0] * P[1])};
        int64_t b[3];
        for (int d = 0; d < 3; ++d) b[d] = std::max<int64_t>(0, (cc[d] + M[d] - 1) / M[d] - 1);
        owner_rank[i] = (int32_t)(b[0] + N[0] * (b[1] + (int64_t)N[1] * b[2]));
    }
    return 0;
}
