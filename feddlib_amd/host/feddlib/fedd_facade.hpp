// Host-side mirror of the reference's operator surface for the assembly + solve hot path,
// namespace FEDD, same class / method names, argument meaning and error behaviour (exceptions),
// sitting on top of the C ABI of include/fedd_hip.h.  Header-only, C++17, no Trilinos needed
// (Teuchos_shim.hpp).  Each class cites the reference declaration it mirrors.
//
//   Map                 feddlib/core/LinearAlgebra/Map_decl.hpp
//   MultiVector         feddlib/core/LinearAlgebra/MultiVector_decl.hpp
//   Matrix              feddlib/core/LinearAlgebra/Matrix_decl.hpp
//   BlockMatrix / BlockMultiVector   feddlib/core/LinearAlgebra/Block*_decl.hpp
//   Elements / FiniteElement         feddlib/core/FE/Elements.hpp, FiniteElement.hpp
//   Domain              feddlib/core/FE/Domain_decl.hpp:66-203
//   FE                  feddlib/core/FE/FE_decl.hpp:40-488
//   BCBuilder           feddlib/core/General/BCBuilder_decl.hpp:36-84
//   Problem             feddlib/problems/abstract/Problem_decl.hpp:38-229
//   Laplace / LinElas / Stokes   feddlib/problems/specific/{Laplace,LinElas,Stokes}_decl.hpp
//   MeshPartitioner     feddlib/core/Mesh/MeshPartitioner_decl.hpp (one rank: read)
//   ExporterParaView    feddlib/core/General/ExporterParaView_decl.hpp (XDMF + raw binary)
//   LinearSolver        feddlib/problems/Solver/LinearSolver_decl.hpp
//
// Differences that are deliberate: SoA mesh storage behind the same accessors; the matrix lives on
// the GPU (host copy pulled lazily by the row-view accessors); one process = one GPU = one rank.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <iostream>
#include <fstream>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/fedd_hip.h"
#include "../Teuchos_shim.hpp"

typedef double default_sc;
typedef int default_lo;
typedef long long default_go;
struct default_no { };
typedef unsigned UN;

namespace FEDD {

typedef std::vector<double> vec_dbl_Type;
typedef std::vector<int> vec_int_Type;
typedef std::vector<std::vector<double>> vec2D_dbl_Type;
typedef Teuchos::RCP<vec_dbl_Type> vec_dbl_ptr_Type;
typedef Teuchos::RCP<vec_int_Type> vec_int_ptr_Type;
typedef Teuchos::RCP<vec2D_dbl_Type> vec2D_dbl_ptr_Type;
typedef Teuchos::RCP<Teuchos::ParameterList> ParameterListPtr_Type;
// feddlib/core/FEDDCore.hpp:115-118 (boost::function there)
typedef std::function<void(double* x, double* res, double* parameters)> RhsFunc_Type;
typedef std::function<void(double* x, double* res, double t, const double* parameters)> BC_func_Type;

inline void feddCheck(int rc, const char* what) {
    TEUCHOS_TEST_FOR_EXCEPTION(rc != 0, std::runtime_error, what << ": " << fedd_last_error());
}

// One GPU context per Domain (mesh resident on the device); shared by the objects built on it.
struct DeviceContext {
    fedd_ctx* ctx = nullptr;
    long generation = 0;      // bumped by every assembly into the context
    Teuchos::RCP<const Teuchos::Comm<int> > comm;
    explicit DeviceContext(int device) { feddCheck(fedd_ctx_create(&ctx, device, nullptr, 0, 1), "fedd_ctx_create"); }
    // One context per rank of the communicator (the reference: one MPI rank per subdomain block).  Ranks = processes:
    // GPU = LOCAL_RANK (else rank), RCCL communicator from the 128-byte id that rank 0 makes and the communicator
    // broadcasts.  Ranks = threads of one process (Teuchos::runAsRanks): one GPU, the library's host-callback transport.
    explicit DeviceContext(const Teuchos::RCP<const Teuchos::Comm<int> >& c) : comm(c) {
        const int rank = c->getRank(), size = c->getSize();
        const char* lr = std::getenv("LOCAL_RANK");
        if (size == 1) {
            feddCheck(fedd_ctx_create(&ctx, lr ? std::atoi(lr) : 0, nullptr, 0, 1), "fedd_ctx_create");
        } else if (c->backend()->inProcess()) {
            feddCheck(fedd_ctx_create(&ctx, lr ? std::atoi(lr) : 0, nullptr, rank, size), "fedd_ctx_create");
            feddCheck(fedd_comm_set_host_callbacks(ctx, &DeviceContext::exchangeCb, &DeviceContext::allreduceCb, this), "fedd_comm_set_host_callbacks");
        } else {
            unsigned char id[128] = {0};
            if (rank == 0) feddCheck(fedd_nccl_unique_id(id), "fedd_nccl_unique_id");
            c->broadcast(0, sizeof(id), id);
            feddCheck(fedd_ctx_create(&ctx, lr ? std::atoi(lr) : rank, id, rank, size), "fedd_ctx_create");
        }
    }
    // after fedd_mesh_set* and fedd_halo_set_owners: the exchange plan of the ghost imports
    void finishHaloPlan() {
        const int size = comm.is_null() ? 1 : comm->getSize();
        if (size == 1) return;
        if (!comm->backend()->inProcess()) {
            feddCheck(fedd_halo_exchange_setup(ctx), "fedd_halo_exchange_setup");
            return;
        }
        // transport-agnostic form: every rank learns what the others ask of it (an all-to-all of the gid lists, here
        // through two all-gathers: counts, then the lists padded to the longest)
        const int rank = comm->getRank();
        std::vector<int64_t> cnt(size, 0);
        feddCheck(fedd_halo_requests_sizes(ctx, cnt.data()), "fedd_halo_requests_sizes");
        int64_t mine = 0;
        for (int64_t v : cnt) mine += v;
        std::vector<int64_t> gids((size_t)std::max<int64_t>(mine, 1));
        feddCheck(fedd_halo_requests_get(ctx, gids.data()), "fedd_halo_requests_get");
        std::vector<char> allc;
        comm->gatherAll(cnt.data(), size * sizeof(int64_t), allc);
        const int64_t* ac = (const int64_t*)allc.data();          // ac[p * size + q] = what rank p asks of rank q
        int64_t longest = 1;
        for (int p = 0; p < size; ++p) {
            int64_t t = 0;
            for (int q = 0; q < size; ++q) t += ac[(size_t)p * size + q];
            longest = std::max(longest, t);
        }
        gids.resize((size_t)longest, 0);
        std::vector<char> allg;
        comm->gatherAll(gids.data(), (size_t)longest * sizeof(int64_t), allg);
        const int64_t* ag = (const int64_t*)allg.data();
        std::vector<int64_t> fromMe(size, 0), lists;
        for (int p = 0; p < size; ++p) {
            int64_t off = 0;
            for (int q = 0; q < rank; ++q) off += ac[(size_t)p * size + q];
            fromMe[p] = ac[(size_t)p * size + rank];
            lists.insert(lists.end(), ag + (size_t)p * longest + off, ag + (size_t)p * longest + off + fromMe[p]);
        }
        if (lists.empty()) lists.push_back(0);
        feddCheck(fedd_halo_requests_set(ctx, fromMe.data(), lists.data()), "fedd_halo_requests_set");
    }
    ~DeviceContext() { if (ctx) fedd_ctx_destroy(ctx); }
private:
    static int exchangeCb(void* user, int n_peers, const int32_t* peers, const int64_t* send_ptr, const double* send_buf,
                          const int64_t* recv_ptr, double* recv_buf, int dofs) {
        DeviceContext* self = (DeviceContext*)user;
        try {
            const int rank = self->comm->getRank();
            Teuchos::CommBackend* be = self->comm->backend();
            for (int k = 0; k < n_peers; ++k)
                if (send_ptr[k + 1] > send_ptr[k])
                    be->send(rank, peers[k], send_buf + send_ptr[k] * dofs, (size_t)((send_ptr[k + 1] - send_ptr[k]) * dofs));
            for (int k = 0; k < n_peers; ++k)
                if (recv_ptr[k + 1] > recv_ptr[k])
                    be->recv(peers[k], rank, recv_buf + recv_ptr[k] * dofs, (size_t)((recv_ptr[k + 1] - recv_ptr[k]) * dofs));
            return 0;
        } catch (const std::exception& e) {
            std::cerr << "facade exchange callback: " << e.what() << std::endl;
            return 1;
        }
    }
    static int allreduceCb(void* user, double* buf, int n) {
        DeviceContext* self = (DeviceContext*)user;
        try {
            self->comm->sumAll(buf, n);
            return 0;
        } catch (const std::exception& e) {
            std::cerr << "facade allreduce callback: " << e.what() << std::endl;
            return 1;
        }
    }
public:
    DeviceContext(const DeviceContext&) = delete;
    DeviceContext& operator=(const DeviceContext&) = delete;
};
typedef Teuchos::RCP<DeviceContext> DeviceContextPtr;

template <class LO = default_lo, class GO = default_go, class NO = default_no>
class Map {
public:
    typedef Teuchos::Comm<int> Comm_Type;
    typedef Teuchos::RCP<const Comm_Type> CommConstPtr_Type;
    // nGlobal = number of global ids over all ranks (-1: this rank's largest id + 1, right for one rank)
    Map(const std::vector<GO>& gids, CommConstPtr_Type comm, GO nGlobal = -1) : gids_(gids), comm_(comm) {
        for (size_t i = 0; i < gids_.size(); ++i) maxGid_ = std::max(maxGid_, gids_[i]);
        if (nGlobal >= 0) maxGid_ = nGlobal - 1;
    }
    LO getNodeNumElements() const { return (LO)gids_.size(); }
    GO getGlobalNumElements() const { return maxGid_ + 1; }
    GO getGlobalElement(LO i) const { return gids_.at(i); }
    GO getMaxAllGlobalIndex() const { return maxGid_; }
    LO getLocalElement(GO g) const {     // hash lookup, the table is built on first use
        if (lookup_.empty() && !gids_.empty()) {
            lookup_.reserve(gids_.size() * 2);
            for (size_t i = 0; i < gids_.size(); ++i) lookup_.emplace(gids_[i], (LO)i);
        }
        auto it = lookup_.find(g);
        return it == lookup_.end() ? (LO)-1 : it->second;
    }
    CommConstPtr_Type getComm() const { return comm_; }
    // vector-field map: dof = dim*node + d   (Map_def.hpp:95-107)
    Teuchos::RCP<Map> buildVecFieldMap(UN dofs) const {
        std::vector<GO> g(gids_.size() * dofs);
        for (size_t i = 0; i < gids_.size(); ++i)
            for (UN d = 0; d < dofs; ++d) g[i * dofs + d] = dofs * gids_[i] + d;
        return Teuchos::rcp(new Map(g, comm_, (GO)dofs * (maxGid_ + 1)));
    }
    const std::vector<GO>& gids() const { return gids_; }
private:
    std::vector<GO> gids_;
    mutable std::unordered_map<GO, LO> lookup_;
    GO maxGid_ = -1;
    CommConstPtr_Type comm_;
};

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class MultiVector {
public:
    typedef Map<LO, GO, NO> Map_Type;
    typedef Teuchos::RCP<const Map_Type> MapConstPtr_Type;
    MultiVector(MapConstPtr_Type map, UN nmbVectors = 1) : map_(map), data_(nmbVectors, std::vector<SC>(map->getNodeNumElements(), 0.)) {}
    MapConstPtr_Type getMap() const { return map_; }
    UN getNumVectors() const { return (UN)data_.size(); }
    size_t getLocalLength() const { return data_[0].size(); }
    Teuchos::ArrayRCP<SC> getDataNonConst(UN i) { return Teuchos::ArrayRCP<SC>(data_.at(i).data(), data_[i].size()); }
    Teuchos::ArrayRCP<const SC> getData(UN i) const { return Teuchos::ArrayRCP<const SC>(data_.at(i).data(), data_[i].size()); }
    void putScalar(const SC& v) { for (auto& c : data_) std::fill(c.begin(), c.end(), v); }
    void update(const SC& alpha, const MultiVector& A, const SC& beta) {
        for (size_t j = 0; j < data_.size(); ++j)
            for (size_t i = 0; i < data_[j].size(); ++i) data_[j][i] = alpha * A.data_[j][i] + beta * data_[j][i];
    }
    SC norm2(UN j = 0) const {      // over all ranks (the map is a unique map: every entry counted once)
        double s = 0;
        for (SC v : data_.at(j)) s += v * v;
        if (!map_->getComm().is_null()) map_->getComm()->sumAll(&s, 1);
        return std::sqrt(s);
    }
    std::vector<SC>& raw(UN j = 0) { return data_.at(j); }
    const std::vector<SC>& raw(UN j = 0) const { return data_.at(j); }
private:
    MapConstPtr_Type map_;
    std::vector<std::vector<SC>> data_;
};

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class Matrix {
public:
    typedef Map<LO, GO, NO> Map_Type;
    typedef Teuchos::RCP<Map_Type> MapPtr_Type;
    typedef Teuchos::RCP<const Map_Type> MapConstPtr_Type;
    typedef MultiVector<SC, LO, GO, NO> MultiVector_Type;
    // row map = unique (vec-field) map, second argument = allocation hint only (Matrix_def.hpp:46-51)
    Matrix(MapConstPtr_Type map, LO numEntries = 0) : map_(map) { (void)numEntries; }
    MapConstPtr_Type getMap(std::string = "row") const { return map_; }
    bool isFillComplete() const { return filled_; }
    void fillComplete() { filled_ = true; }
    void resumeFill() { filled_ = false; }
    // binding to the device matrix the assembly produced
    void bind(DeviceContextPtr dev, int dofs) { dev_ = dev; gen_ = dev->generation; dofs_ = dofs; hostValid_ = false; filled_ = true; }
    // block of a mixed problem held in a numbered slot beside the system matrix (fedd_matrix_store / fedd_assemble_div)
    void bindSlot(DeviceContextPtr dev, int slot) { dev_ = dev; gen_ = dev->generation; slot_ = slot; hostValid_ = false; filled_ = true; }
    int slot() const { return slot_; }
    // Matrix::scale (used by Stokes::assemble, Stokes_def.hpp:83-85)
    void scale(const SC& alpha) {
        TEUCHOS_TEST_FOR_EXCEPTION(dev_.is_null(), std::runtime_error, "Matrix::scale: no assembled data");
        feddCheck(fedd_matrix_scale(dev_->ctx, slot_, alpha), "fedd_matrix_scale");
        hostValid_ = false;
    }
    void fillComplete(MapConstPtr_Type, MapConstPtr_Type) { filled_ = true; }
    bool isResident() const { return !dev_.is_null() && (slot_ >= 0 || gen_ == dev_->generation); }
    DeviceContextPtr device() const { return dev_; }
    void invalidateHost() { hostValid_ = false; }
    GO getGlobalNumEntries() { pull(); return (GO)val_.size(); }
    LO getNumEntriesInLocalRow(LO row) { pull(); return (LO)(rowptr_.at(row + 1) - rowptr_.at(row)); }
    // Matrix::getLocalRowView (local column indices; colGid() maps them to global dof ids)
    void getLocalRowView(LO row, Teuchos::ArrayView<const LO>& indices, Teuchos::ArrayView<const SC>& values) {
        pull();
        const int64_t b = rowptr_.at(row), e = rowptr_.at(row + 1);
        indices = Teuchos::ArrayView<const LO>(col_.data() + b, (size_t)(e - b));
        values = Teuchos::ArrayView<const SC>(val_.data() + b, (size_t)(e - b));
    }
    GO colGid(LO localCol) { pull(); return (GO)colGid_.at(localCol); }
    // y = A x on owned rows (Matrix::apply, Matrix_def.hpp:245-254)
    void apply(const MultiVector_Type& X, MultiVector_Type& Y) {
        TEUCHOS_TEST_FOR_EXCEPTION(!isResident(), std::runtime_error, "Matrix::apply: matrix is not the one resident on the device");
        feddCheck(fedd_spmv(dev_->ctx, X.raw().data(), Y.raw().data()), "fedd_spmv");
    }
    void print() {
        pull();
        for (size_t r = 0; r + 1 < rowptr_.size(); ++r)
            for (int64_t p = rowptr_[r]; p < rowptr_[r + 1]; ++p)
                std::cout << map_->getGlobalElement((LO)r) << " " << colGid_[col_[p]] << " " << val_[p] << "\n";
    }
private:
    void pull() {
        if (hostValid_) return;
        TEUCHOS_TEST_FOR_EXCEPTION(!isResident(), std::runtime_error, "Matrix: no assembled data");
        int64_t nr, nc, nnz;
        if (slot_ >= 0) {      // a stored block: local column ids of the block's own column space
            feddCheck(fedd_matrix_sizes(dev_->ctx, slot_, &nr, &nc, &nnz), "fedd_matrix_sizes");
            rowptr_.resize(nr + 1); col_.resize(nnz); val_.resize(nnz); colGid_.resize(nc);
            feddCheck(fedd_matrix_get(dev_->ctx, slot_, rowptr_.data(), col_.data(), val_.data()), "fedd_matrix_get");
            for (int64_t i = 0; i < nc; ++i) colGid_[i] = i;
            hostValid_ = true;
            return;
        }
        feddCheck(fedd_csr_sizes(dev_->ctx, &nr, &nc, &nnz), "fedd_csr_sizes");
        rowptr_.resize(nr + 1); col_.resize(nnz); val_.resize(nnz); colGid_.resize(nc);
        feddCheck(fedd_csr_get(dev_->ctx, rowptr_.data(), col_.data(), val_.data(), colGid_.data()), "fedd_csr_get");
        hostValid_ = true;
    }
    MapConstPtr_Type map_;
    DeviceContextPtr dev_;
    long gen_ = -1;
    int dofs_ = 1, slot_ = -1;
    bool filled_ = false, hostValid_ = false;
    std::vector<int64_t> rowptr_, colGid_;
    std::vector<int32_t> col_;
    std::vector<double> val_;
};

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class BlockMatrix {
public:
    typedef Matrix<SC, LO, GO, NO> Matrix_Type;
    typedef Teuchos::RCP<Matrix_Type> MatrixPtr_Type;
    BlockMatrix(UN size) : n_(size), blocks_(size * size) {}
    // BlockMatrix::merge (BlockMatrix_def.hpp:119-148) happens on the device (fedd_block_merge); the flag says that the
    // device's system matrix is the merged [A B^T; B C] of this block matrix
    void setMerged(DeviceContextPtr dev) { mergedDev_ = dev; }
    DeviceContextPtr mergedDevice() const { return mergedDev_; }
    UN size() const { return n_; }
    void addBlock(const MatrixPtr_Type& m, UN i, UN j) { blocks_.at(i * n_ + j) = m; }
    bool blockExists(UN i, UN j) const { return !blocks_.at(i * n_ + j).is_null(); }
    MatrixPtr_Type getBlock(UN i, UN j) const {
        TEUCHOS_TEST_FOR_EXCEPTION(!blockExists(i, j), std::runtime_error, "Block does not exist.");
        return blocks_[i * n_ + j];
    }
private:
    UN n_;
    std::vector<MatrixPtr_Type> blocks_;
    DeviceContextPtr mergedDev_;
};

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class BlockMultiVector {
public:
    typedef MultiVector<SC, LO, GO, NO> MultiVector_Type;
    typedef Teuchos::RCP<MultiVector_Type> MultiVectorPtr_Type;
    BlockMultiVector(UN size) : blocks_(size) {}
    UN size() const { return (UN)blocks_.size(); }
    UN getNumVectors() const { return blocks_.empty() || blocks_[0].is_null() ? 0 : blocks_[0]->getNumVectors(); }
    void addBlock(const MultiVectorPtr_Type& mv, UN i) { blocks_.at(i) = mv; }
    Teuchos::RCP<const MultiVector_Type> getBlock(UN i) const { return blocks_.at(i); }
    MultiVectorPtr_Type getBlockNonConst(UN i) { return blocks_.at(i); }
    void putScalar(const SC& v) { for (auto& b : blocks_) if (!b.is_null()) b->putScalar(v); }
    void update(const SC& a, const BlockMultiVector& A, const SC& b) {
        for (size_t i = 0; i < blocks_.size(); ++i) blocks_[i]->update(a, *A.blocks_[i], b);
    }
private:
    std::vector<MultiVectorPtr_Type> blocks_;
};

// FiniteElement / Elements: same accessors over a flat connectivity array (FiniteElement.hpp:17-100)
class FiniteElement {
public:
    FiniteElement(const int32_t* nodes, int nen, int flag) : nodes_(nodes, nodes + nen), flag_(flag) {}
    const vec_int_Type& getVectorNodeList() const { return nodes_; }
    int getNode(int i) const { return nodes_.at(i); }
    int getFlag() const { return flag_; }
    int size() const { return (int)nodes_.size(); }
private:
    vec_int_Type nodes_;
    int flag_;
};
class Elements {
public:
    Elements(const std::vector<int32_t>& conn, int nen) : conn_(&conn), nen_(nen) {}
    UN numberElements() const { return (UN)(conn_->size() / nen_); }
    FiniteElement getElement(UN T) const { return FiniteElement(conn_->data() + (size_t)T * nen_, nen_, 0); }
    int nodesPerElement() const { return nen_; }
private:
    const std::vector<int32_t>* conn_;
    int nen_;
};
typedef Teuchos::RCP<Elements> ElementsPtr_Type;

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class Domain {
public:
    typedef Map<LO, GO, NO> Map_Type;
    typedef Teuchos::RCP<const Map_Type> MapConstPtr_Type;
    typedef Teuchos::Comm<int> Comm_Type;
    typedef Teuchos::RCP<const Comm_Type> CommConstPtr_Type;

    Domain(CommConstPtr_Type comm, int dimension = 0) : comm_(comm), dim_(dimension) {}
    Domain(vec_dbl_Type coor, double l, double h, CommConstPtr_Type comm) : comm_(comm), coorRec(coor), length(l), height(h) {}
    Domain(vec_dbl_Type coor, double l, double w, double h, CommConstPtr_Type comm) : comm_(comm), coorRec(coor), length(l), width(w), height(h) {}

    // Domain::buildMesh (Domain_def.hpp:201-265): structured square / cube, P1
    void buildMesh(int flagsOption, std::string meshType, int dim, std::string FEType, int N, int M, int numProcsCoarseSolve = 0) {
        TEUCHOS_TEST_FOR_EXCEPTION(meshType != "Square", std::logic_error,
                                   "Select valid mesh. Structured types are 'structured' and 'structured_bfs'; only 'Square' (rectangle/box) is built here.");
        TEUCHOS_TEST_FOR_EXCEPTION(dim != 2 && dim != 3, std::logic_error, "Select valid mesh dimension. 2 or 3 dimensional meshes can be constructed.");
        TEUCHOS_TEST_FOR_EXCEPTION(FEType != "P1", std::logic_error, "Wrong FE-Type: the structured generator of this build makes P1 meshes.");
        TEUCHOS_TEST_FOR_EXCEPTION(!(M >= 1), std::logic_error, "H/h is to small.");
        TEUCHOS_TEST_FOR_EXCEPTION(numProcsCoarseSolve != 0, std::logic_error, "Mpi Ranks Coarse is not supported.");
        dim_ = dim; FEType_ = FEType; n_ = N; m_ = M; flagsOption_ = flagsOption;
        const int rank = comm_->getRank(), size = comm_->getSize();
        int nblocks = 1;
        for (int d = 0; d < dim; ++d) nblocks *= N;
        TEUCHOS_TEST_FOR_EXCEPTION(nblocks != size, std::logic_error, "Domain::buildMesh: " << N << "^" << dim << " subdomain blocks need as many ranks, the communicator has " << size);
        int dec[3] = {N, N, N}, cel[3] = {M, M, M};
        // Several ranks: the rank's block (the reference's repeated map) plus four layers of ghost elements, which
        // complete the rows of the ghost nodes within three layers (row ghosts): the Schwarz boxes a rank boundary
        // crosses are then built whole on both sides and the preconditioner does not depend on the split
        const int ghosts = size > 1 ? 4 : 0;
        int64_t ne, nr, nu, ng;
        feddCheck(fedd_mesh_structured_sizes(dim, dec, cel, rank, ghosts, &ne, &nr, &nu, &ng), "fedd_mesh_structured_sizes");
        conn_.resize(ne * (dim + 1)); xyz_.resize(nr * dim); flagRep_.resize(nr); flagUni_.resize(nu);
        std::vector<int64_t> grep(nr), guni(nu), rowGhost;
        std::vector<int32_t> rowGhostFlag;
        double org[3] = {0, 0, 0}, sz[3] = {length, dim == 2 ? height : width, height};
        for (size_t d = 0; d < coorRec.size() && d < 3; ++d) org[d] = coorRec[d];
        feddCheck(fedd_mesh_structured_build(dim, dec, cel, rank, org, sz, flagsOption, ghosts, conn_.data(), xyz_.data(), grep.data(),
                                             flagRep_.data(), guni.data(), flagUni_.data()), "fedd_mesh_structured_build");
        if (ghosts >= 2) {
            int64_t nrg = 0;
            feddCheck(fedd_mesh_structured_row_ghosts(dim, dec, cel, rank, ghosts, org, sz, flagsOption, &nrg, nullptr, nullptr), "fedd_mesh_structured_row_ghosts");
            rowGhost.resize((size_t)nrg); rowGhostFlag.resize((size_t)nrg);
            if (nrg) feddCheck(fedd_mesh_structured_row_ghosts(dim, dec, cel, rank, ghosts, org, sz, flagsOption, &nrg, rowGhost.data(), rowGhostFlag.data()), "fedd_mesh_structured_row_ghosts");
        }
        finishMesh(grep, guni, dim + 1, ng, &rowGhost, &rowGhostFlag);
        if (size > 1) {     // owner of every repeated node (lowest rank whose block holds it), then the import plan
            std::vector<int32_t> owner(grep.size());
            feddCheck(fedd_mesh_structured_owner(dim, dec, cel, (int64_t)grep.size(), grep.data(), owner.data()), "fedd_mesh_structured_owner");
            feddCheck(fedd_halo_set_owners(dev_->ctx, (int64_t)grep.size(), grep.data(), owner.data()), "fedd_halo_set_owners");
            dev_->finishHaloPlan();
        }
    }
    // generic entry for externally built (e.g. unstructured, P2) meshes in the reference's data model
    void setMesh(int dim, std::string FEType, int nen, const std::vector<int32_t>& conn, const std::vector<double>& xyz,
                 const std::vector<int64_t>& gidRep, const std::vector<int64_t>& gidUni, const std::vector<int32_t>& flagUni) {
        dim_ = dim; FEType_ = FEType; conn_ = conn; xyz_ = xyz; flagUni_ = flagUni; flagRep_.assign(gidRep.size(), 0);
        int64_t ng = 0;
        for (auto g : gidRep) ng = std::max<int64_t>(ng, g + 1);
        finishMesh(gidRep, gidUni, nen, ng);
    }

    // what MeshPartitioner::readAndPartition leaves in the domain on one rank (MeshPartitioner_def.hpp:224-530,
    // MeshUnstructured::readMeshSize / readMeshEntity): INRIA .mesh file, repeated = unique = identity numbering
    // Several ranks: every rank reads the file (as in the reference), partitions the elements with the library's
    // deterministic coordinate bisection (in place of METIS_PartMeshDual, :324) -- the same partition on every rank
    // without communication -- and keeps its part plus two layers of ghost elements (row ghosts within one layer).
    void readMeshFile(const std::string& file, int dim, std::string FEType, int volumeID) {
        if (comm_->getSize() > 1) {
            readAndPartitionMeshFile(file, dim, FEType, volumeID);
            return;
        }
        int64_t nv, ne, ns;
        feddCheck(fedd_mesh_read_sizes(file.c_str(), dim, &nv, &ne, &ns), "fedd_mesh_read_sizes");
        std::vector<double> xyz(nv * dim);
        std::vector<int32_t> vflag(nv), conn(ne * (dim + 1)), eflag(ne);
        surf_.assign(ns * dim, 0); surfFlag_.assign(ns, 0);
        feddCheck(fedd_mesh_read(file.c_str(), dim, xyz.data(), vflag.data(), conn.data(), eflag.data(), surf_.data(), surfFlag_.data()), "fedd_mesh_read");
        std::vector<int64_t> gid(nv);
        for (int64_t i = 0; i < nv; ++i) gid[i] = i;
        volumeID_ = volumeID;
        setMesh(dim, FEType, dim + 1, conn, xyz, gid, gid, vflag);
        flagRep_ = vflag;
    }
    void readAndPartitionMeshFile(const std::string& file, int dim, std::string FEType, int volumeID) {
        TEUCHOS_TEST_FOR_EXCEPTION(FEType != "P1", std::logic_error, "Domain::readMeshFile on several ranks: P1 meshes");
        const int rank = comm_->getRank(), size = comm_->getSize(), nen = dim + 1, layers = 2;
        int64_t nv, ne, ns;
        feddCheck(fedd_mesh_read_sizes(file.c_str(), dim, &nv, &ne, &ns), "fedd_mesh_read_sizes");
        std::vector<double> xyz(nv * dim);
        std::vector<int32_t> vflag(nv), conn(ne * nen), eflag(ne), surf(ns * dim), sflag(ns), part(ne);
        feddCheck(fedd_mesh_read(file.c_str(), dim, xyz.data(), vflag.data(), conn.data(), eflag.data(), surf.data(), sflag.data()), "fedd_mesh_read");
        feddCheck(fedd_mesh_partition(dim, nen, ne, conn.data(), nv, xyz.data(), size, part.data()), "fedd_mesh_partition");
        int64_t nel, nr, nu, nrg;
        feddCheck(fedd_mesh_partition_sizes(nen, ne, conn.data(), nv, part.data(), size, rank, layers, &nel, &nr, &nu, &nrg), "fedd_mesh_partition_sizes");
        conn_.assign(nel * nen, 0); xyz_.assign(nr * dim, 0.); flagRep_.assign(nr, 0); flagUni_.assign(nu, 0);
        std::vector<int64_t> grep(nr), guni(nu), rowGhost(nrg), egid(nel);
        std::vector<int32_t> owner(nr), rowGhostFlag(nrg);
        feddCheck(fedd_mesh_partition_extract(dim, nen, ne, conn.data(), nv, xyz.data(), vflag.data(), part.data(), size, rank, layers,
                                              conn_.data(), xyz_.data(), grep.data(), flagRep_.data(), owner.data(), guni.data(), flagUni_.data(),
                                              rowGhost.data(), rowGhostFlag.data(), egid.data()), "fedd_mesh_partition_extract");
        dim_ = dim; FEType_ = FEType; volumeID_ = volumeID;
        finishMesh(grep, guni, nen, nv, &rowGhost, &rowGhostFlag);
        feddCheck(fedd_halo_set_owners(dev_->ctx, (int64_t)grep.size(), grep.data(), owner.data()), "fedd_halo_set_owners");
        dev_->finishHaloPlan();
    }
    // Domain::buildP2ofP1Domain (Domain_def.hpp -> MeshUnstructured::buildP2ofP1MeshEdge, MeshUnstructured_def.hpp:129-410)
    void buildP2ofP1Domain(const Teuchos::RCP<Domain>& domainP1) {
        TEUCHOS_TEST_FOR_EXCEPTION(comm_->getSize() > 1, std::logic_error, "Domain::buildP2ofP1Domain: one rank in this build");
        const int dim = (int)domainP1->getDimension();
        const int64_t ne = domainP1->getNumElements(), nv = domainP1->getNumPoints("Unique");
        int64_t ned = 0;
        feddCheck(fedd_mesh_p2_sizes(dim, ne, domainP1->conn_.data(), &ned), "fedd_mesh_p2_sizes");
        const int nen2 = dim == 3 ? 10 : 6;
        std::vector<int32_t> conn2(ne * nen2), flag2(nv + ned);
        std::vector<double> xyz2((nv + ned) * dim);
        feddCheck(fedd_mesh_p2_build(dim, nv, ne, domainP1->conn_.data(), domainP1->xyz_.data(), domainP1->flagRep_.data(),
                                     (int64_t)domainP1->surfFlag_.size(), domainP1->surf_.data(), domainP1->surfFlag_.data(),
                                     domainP1->volumeID_, conn2.data(), xyz2.data(), flag2.data()), "fedd_mesh_p2_build");
        std::vector<int64_t> gid(nv + ned);
        for (size_t i = 0; i < gid.size(); ++i) gid[i] = (int64_t)i;
        setMesh(dim, "P2", nen2, conn2, xyz2, gid, gid, flag2);
        flagRep_ = flag2;
        nP1_ = nv;
    }
    int64_t numberOfP1Nodes() const { return nP1_; }      // P2-of-P1 domain: its first nodes are the P1 nodes

    LO getApproxEntriesPerRow() const {      // Domain_def.hpp:176-198 (allocation hint only)
        if (dim_ == 2) return FEType_ == "P1" ? 20 : 30;
        return FEType_ == "P1" ? 50 : (FEType_ == "P2" ? 80 : 100);
    }
    UN getDimension() const { return (UN)dim_; }
    std::string getFEType() const { return FEType_; }
    CommConstPtr_Type getComm() const { return comm_; }
    MapConstPtr_Type getMapUnique() const { return mapUnique_; }
    MapConstPtr_Type getMapRepeated() const { return mapRepeated_; }
    MapConstPtr_Type getMapVecFieldUnique() const { return mapUnique_->buildVecFieldMap(dim_); }
    MapConstPtr_Type getMapVecFieldRepeated() const { return mapRepeated_->buildVecFieldMap(dim_); }
    vec2D_dbl_ptr_Type getPointsRepeated() const { return points(xyz_, (size_t)mapRepeated_->getNodeNumElements(), nullptr); }
    vec2D_dbl_ptr_Type getPointsUnique() const { return points(xyz_, (size_t)mapUnique_->getNodeNumElements(), &uniOfRep_); }
    vec_int_ptr_Type getBCFlagUnique() const { return Teuchos::rcp(new vec_int_Type(flagUni_.begin(), flagUni_.end())); }
    vec_int_ptr_Type getBCFlagRepeated() const { return Teuchos::rcp(new vec_int_Type(flagRep_.begin(), flagRep_.end())); }
    ElementsPtr_Type getElementsC() const { return Teuchos::rcp(new Elements(conn_, nen_)); }
    LO getNumElements() const { return (LO)(conn_.size() / nen_); }
    GO getNumElementsGlobal() const { return (GO)(conn_.size() / nen_); }
    LO getNumPoints(std::string type = "Unique") const { return type == "Unique" ? mapUnique_->getNodeNumElements() : mapRepeated_->getNodeNumElements(); }
    void info() const {
        if (comm_->getRank() == 0)
            std::cout << "\t### Domain: dim " << dim_ << ", FE " << FEType_ << ", elements " << getNumElements() << ", nodes "
                      << mapUnique_->getGlobalNumElements() << " ###" << std::endl;
    }
    // Domain::getMesh (Domain_decl.hpp): the facade's Domain holds the mesh arrays itself, so "the mesh" is a view of it
    Teuchos::RCP<const Domain> getMesh() const { return Teuchos::RCP<const Domain>(std::shared_ptr<const Domain>(this, [](const Domain*) {})); }
    const std::vector<int32_t>& connectivity() const { return conn_; }
    // facade internals
    DeviceContextPtr device() const { return dev_; }
    int nodesPerElement() const { return nen_; }
    const std::vector<double>& xyzRepeated() const { return xyz_; }
    const std::vector<int32_t>& uniqueLocalOfRepeated() const { return uniOfRep_; }
    // several ranks: nodes of other ranks whose rows this rank also builds (fedd_mesh_set_rows); their device node id is
    // number of unique nodes + position here
    const std::vector<int32_t>& rowGhostRepeatedIds() const { return rowGhostRep_; }
    const std::vector<int32_t>& rowGhostFlags() const { return rowGhostFlag_; }

private:
    void finishMesh(const std::vector<int64_t>& grep, const std::vector<int64_t>& guni, int nen, int64_t nGlobal,
                    const std::vector<int64_t>* rowGhost = nullptr, const std::vector<int32_t>* rowGhostFlag = nullptr) {
        nen_ = nen;
        std::vector<GO> gr(grep.begin(), grep.end()), gu(guni.begin(), guni.end());
        mapRepeated_ = Teuchos::rcp(new Map_Type(gr, comm_, (GO)nGlobal));
        mapUnique_ = Teuchos::rcp(new Map_Type(gu, comm_, (GO)nGlobal));
        // repeated-local id of every unique node (for getPointsUnique)
        uniOfRep_.assign(guni.size(), -1);
        std::vector<std::pair<int64_t, int32_t>> s(grep.size());
        for (size_t i = 0; i < grep.size(); ++i) s[i] = {grep[i], (int32_t)i};
        std::sort(s.begin(), s.end());
        for (size_t i = 0; i < guni.size(); ++i) {
            auto it = std::lower_bound(s.begin(), s.end(), std::make_pair(guni[i], (int32_t)-1));
            TEUCHOS_TEST_FOR_EXCEPTION(it == s.end() || it->first != guni[i], std::runtime_error, "unique node missing from repeated map");
            uniOfRep_[i] = it->second;
        }
        dev_ = Teuchos::rcp(new DeviceContext(comm_));
        rowGhostRep_.clear();
        rowGhostFlag_.clear();
        if (rowGhost) {     // their repeated-local ids (coordinates) and flags: the Dirichlet treatment covers their rows too
            for (size_t k = 0; k < rowGhost->size(); ++k) {
                auto it = std::lower_bound(s.begin(), s.end(), std::make_pair((*rowGhost)[k], (int32_t)-1));
                TEUCHOS_TEST_FOR_EXCEPTION(it == s.end() || it->first != (*rowGhost)[k], std::runtime_error, "row ghost missing from repeated map");
                rowGhostRep_.push_back(it->second);
                rowGhostFlag_.push_back((*rowGhostFlag)[k]);
            }
        }
        if (rowGhost && !rowGhost->empty())
            feddCheck(fedd_mesh_set_rows(dev_->ctx, dim_, nen_, (int64_t)(conn_.size() / nen_), conn_.data(), (int64_t)grep.size(), xyz_.data(),
                                         grep.data(), (int64_t)guni.size(), guni.data(), flagUni_.data(), (int64_t)rowGhost->size(),
                                         rowGhost->data(), rowGhostFlag->data()), "fedd_mesh_set_rows");
        else
            feddCheck(fedd_mesh_set(dev_->ctx, dim_, nen_, (int64_t)(conn_.size() / nen_), conn_.data(), (int64_t)grep.size(), xyz_.data(),
                                    grep.data(), (int64_t)guni.size(), guni.data(), flagUni_.data()), "fedd_mesh_set");
    }
    vec2D_dbl_ptr_Type points(const std::vector<double>& xyz, size_t n, const std::vector<int32_t>* idx) const {
        vec2D_dbl_ptr_Type p = Teuchos::rcp(new vec2D_dbl_Type(n, vec_dbl_Type(dim_, 0.)));
        for (size_t i = 0; i < n; ++i) {
            const size_t src = idx ? (size_t)(*idx)[i] : i;
            for (int d = 0; d < dim_; ++d) (*p)[i][d] = xyz[src * dim_ + d];
        }
        return p;
    }
    CommConstPtr_Type comm_;
    vec_dbl_Type coorRec;
    double length = 1., width = 1., height = 1.;
    int dim_ = 0, n_ = 0, m_ = 0, flagsOption_ = 0, nen_ = 0;
    std::string FEType_;
    std::vector<int32_t> conn_, flagRep_, flagUni_, uniOfRep_, surf_, surfFlag_, rowGhostRep_, rowGhostFlag_;
    std::vector<double> xyz_;
    int volumeID_ = 10;
    int64_t nP1_ = 0;
    Teuchos::RCP<Map_Type> mapRepeated_, mapUnique_;
    DeviceContextPtr dev_;
};

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class FE {
public:
    typedef Domain<SC, LO, GO, NO> Domain_Type;
    typedef Teuchos::RCP<const Domain_Type> DomainConstPtr_Type;
    typedef Matrix<SC, LO, GO, NO> Matrix_Type;
    typedef Teuchos::RCP<Matrix_Type> MatrixPtr_Type;
    typedef MultiVector<SC, LO, GO, NO> MultiVector_Type;
    typedef Teuchos::RCP<MultiVector_Type> MultiVectorPtr_Type;

    FE(bool saveAssembly = false) { (void)saveAssembly; }
    void addFE(DomainConstPtr_Type domain) { domainVec_.push_back(domain); }
    void doSetZeros(double eps = 10 * 2.220446049250313e-16) { setZeros_ = true; myeps_ = eps; }

    // FE_def.hpp:6932-6953
    UN checkFE(int dim, std::string FEType) const {
        for (UN i = 0; i < domainVec_.size(); ++i)
            if ((int)domainVec_[i]->getDimension() == dim && domainVec_[i]->getFEType() == FEType) return i;
        TEUCHOS_TEST_FOR_EXCEPTION(true, std::runtime_error, "Wrong FEType or dimension. No assembly possible.");
        return 0;
    }
    void assemblyLaplace(int dim, std::string FEType, int degree, MatrixPtr_Type& A, bool callFillComplete = true, int FELocExternal = -1) {
        TEUCHOS_TEST_FOR_EXCEPTION(FEType == "P0", std::logic_error, "Not implemented for P0");
        (void)degree;
        assembleInto(FELocExternal < 0 ? checkFE(dim, FEType) : (UN)FELocExternal, 1, FEDD_BLOCK_SCALAR, FEDD_FORM_LAPLACE, nullptr, A, callFillComplete);
    }
    void assemblyLaplaceVecField(int dim, std::string FEType, int degree, MatrixPtr_Type& A, bool callFillComplete = true) {
        TEUCHOS_TEST_FOR_EXCEPTION(FEType == "P1-disc" || FEType == "P0", std::logic_error, "Not implemented for P0 or P1-disc");
        (void)degree;
        // doSetZeros (FE_def.hpp:74-79, 719-721): element contributions below myeps_ are dropped before they are added
        feddCheck(fedd_set_option(domainVec_.at(checkFE(dim, FEType))->device()->ctx, "asm_zero_eps", setZeros_ ? myeps_ : 0.), "fedd_set_option");
        assembleInto(checkFE(dim, FEType), dim, FEDD_BLOCK_DIAG, FEDD_FORM_LAPLACE_VEC, nullptr, A, callFillComplete);
    }
    void assemblyMass(int dim, std::string FEType, std::string fieldType, MatrixPtr_Type& A, bool callFillComplete = true) {
        TEUCHOS_TEST_FOR_EXCEPTION(FEType == "P0", std::logic_error, "Not implemented for P0");
        if (fieldType == "Scalar") assembleInto(checkFE(dim, FEType), 1, FEDD_BLOCK_SCALAR, FEDD_FORM_MASS, nullptr, A, callFillComplete);
        else if (fieldType == "Vector") assembleInto(checkFE(dim, FEType), dim, FEDD_BLOCK_DIAG, FEDD_FORM_MASS_VEC, nullptr, A, callFillComplete);
        else TEUCHOS_TEST_FOR_EXCEPTION(true, std::logic_error, "Specify valid vieldType for assembly of mass matrix.");
    }
    // FE::assemblyBDStabilization (FE_def.hpp:2151-2220): the Bochev-Dohrmann pressure block of P1/P1 Stokes
    void assemblyBDStabilization(int dim, std::string FEType, MatrixPtr_Type& A, bool callFillComplete = true) {
        TEUCHOS_TEST_FOR_EXCEPTION(FEType != "P1", std::logic_error, "Only implemented for P1. Q1 is equivalent but we need to adjust scaling for the reference element.");
        assembleInto(checkFE(dim, FEType), 1, FEDD_BLOCK_SCALAR, FEDD_FORM_BDSTAB, nullptr, A, callFillComplete);
    }
    void assemblyLinElasXDim(int dim, std::string FEType, MatrixPtr_Type& A, double lambda, double mu, bool callFillComplete = true) {
        TEUCHOS_TEST_FOR_EXCEPTION(FEType == "P0", std::logic_error, "Not implemented for P0");
        const double p[2] = {lambda, mu};
        assembleInto(checkFE(dim, FEType), dim, FEDD_BLOCK_FULL, FEDD_FORM_LINELAS, p, A, callFillComplete);
    }
    // FE::assemblyDivAndDivT (FE_def.hpp:1932-2057): velocity = FEType1 on the first domain, pressure = P1 on the vertices,
    // which are the first nodes of a P2-of-P1 velocity domain.  B and B^T land in slots 1 and 2 of the velocity domain's
    // device context (the system slot is scratch for the node pattern), unscaled.
    void assemblyDivAndDivT(int dim, std::string FEType1, std::string FEType2, int degree, MatrixPtr_Type& Bmat, MatrixPtr_Type& BTmat,
                            Teuchos::RCP<const Map<LO, GO, NO>> map1, Teuchos::RCP<const Map<LO, GO, NO>> map2, bool callFillComplete = true) {
        (void)degree; (void)map1; (void)callFillComplete;
        TEUCHOS_TEST_FOR_EXCEPTION(FEType2 != "P1", std::logic_error, "assemblyDivAndDivT: the pressure space of this build is P1");
        const UN loc1 = checkFE(dim, FEType1);
        auto dom = domainVec_.at(loc1);
        const int64_t n_p = (int64_t)map2->getNodeNumElements();
        feddCheck(fedd_set_option(dom->device()->ctx, "asm_zero_eps", setZeros_ ? myeps_ : 0.), "fedd_set_option");   // (FE_def.hpp:2002-2004, 2032-2034)
        feddCheck(fedd_assemble_div(dom->device()->ctx, n_p, 1, 2), "fedd_assemble_div");
        dom->device()->generation++;
        Bmat->bindSlot(dom->device(), 1);
        BTmat->bindSlot(dom->device(), 2);
    }
    // FE::assemblyRHS (FE_def.hpp:4694-4766): f evaluated once (constant); a lives on the REPEATED
    // map in the reference and is then export-added; here the owned entries are produced directly,
    // `a` must live on the unique (vec-field) map.
    void assemblyRHS(int dim, std::string FEType, MultiVectorPtr_Type a, std::string fieldType, RhsFunc_Type func, std::vector<SC>& funcParameter) {
        TEUCHOS_TEST_FOR_EXCEPTION(FEType == "P0", std::logic_error, "Not implemented for P0");
        TEUCHOS_TEST_FOR_EXCEPTION(a.is_null(), std::runtime_error, "MultiVector in assemblyConstRHS is null.");
        TEUCHOS_TEST_FOR_EXCEPTION(a->getNumVectors() > 1, std::logic_error, "Implement for numberMV > 1 .");
        const UN loc = checkFE(dim, FEType);
        const int dofs = fieldType == "Scalar" ? 1 : (fieldType == "Vector" ? dim : 0);
        TEUCHOS_TEST_FOR_EXCEPTION(dofs == 0, std::logic_error, "Invalid field type.");
        const int degFunc = (int)(funcParameter[funcParameter.size() - 1] + 1.e-14);
        double x = 0., f[3] = {0, 0, 0};
        func(&x, f, funcParameter.data());            // "for now just const!" (FE_def.hpp:4731-4736)
        fedd_ctx* ctx = domainVec_[loc]->device()->ctx;
        int64_t nr = 0, nc = 0, nnz = 0;
        if (fedd_csr_sizes(ctx, &nr, &nc, &nnz) != 0 || nr != (int64_t)a->getLocalLength()) {
            int64_t dummy;                             // rhs before any matrix: build the matching pattern
            feddCheck(fedd_pattern_build(ctx, dofs, dofs == 1 ? FEDD_BLOCK_SCALAR : FEDD_BLOCK_DIAG, &dummy), "fedd_pattern_build");
            domainVec_[loc]->device()->generation++;
        }
        feddCheck(fedd_assemble_rhs(ctx, dofs, f, degFunc), "fedd_assemble_rhs");
        feddCheck(fedd_rhs_get(ctx, a->raw().data()), "fedd_rhs_get");
    }
private:
    void assembleInto(UN loc, int dofs, int mode, int form, const double* params, MatrixPtr_Type& A, bool callFillComplete) {
        TEUCHOS_TEST_FOR_EXCEPTION(A.is_null(), std::runtime_error, "Matrix is null.");
        auto dom = domainVec_.at(loc);
        TEUCHOS_TEST_FOR_EXCEPTION((size_t)A->getMap()->getNodeNumElements() != (size_t)dom->getMapUnique()->getNodeNumElements() * dofs,
                                   std::logic_error, "Matrix row map does not match the unique map of the domain.");
        fedd_ctx* ctx = dom->device()->ctx;
        int64_t nnz;
        feddCheck(fedd_pattern_build(ctx, dofs, mode, &nnz), "fedd_pattern_build");
        feddCheck(fedd_assemble(ctx, form, params), "fedd_assemble");
        dom->device()->generation++;
        A->bind(dom->device(), dofs);
        if (!callFillComplete) A->resumeFill();
    }
    std::vector<DomainConstPtr_Type> domainVec_;
    bool setZeros_ = false;
    double myeps_ = 0.;
};

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class BCBuilder {
public:
    typedef Domain<SC, LO, GO, NO> Domain_Type;
    typedef Teuchos::RCP<Domain_Type> DomainPtr_Type;
    typedef Teuchos::RCP<const Domain_Type> DomainConstPtr_Type;
    typedef BlockMatrix<SC, LO, GO, NO> BlockMatrix_Type;
    typedef Teuchos::RCP<BlockMatrix_Type> BlockMatrixPtr_Type;
    typedef BlockMultiVector<SC, LO, GO, NO> BlockMultiVector_Type;
    typedef Teuchos::RCP<BlockMultiVector_Type> BlockMultiVectorPtr_Type;

    BCBuilder() {}
    void addBC(BC_func_Type funcBC, int flag, int block, const DomainPtr_Type& domain, std::string type, int dofs) {
        vec_dbl_Type dummy(1, 0.);
        addBC(funcBC, flag, block, domain, type, dofs, dummy);
    }
    void addBC(BC_func_Type funcBC, int flag, int block, const DomainPtr_Type& domain, std::string type, int dofs, vec_dbl_Type& parameter_vec) {
        vecBC_func_.push_back(funcBC); vecFlag_.push_back(flag); vecBlockID_.push_back(block); vecDomain_.push_back(domain);
        vecBCType_.push_back(type); vecDofs_.push_back(dofs); vecBC_Parameters_.push_back(parameter_vec);
    }
    bool findFlag(LO flag, int block, int& loc) const {   // BCBuilder_def.hpp findFlag
        for (size_t i = 0; i < vecFlag_.size(); ++i)
            if (vecFlag_[i] == flag && vecBlockID_[i] == block) { loc = (int)i; return true; }
        return false;
    }
    bool blockHasDirichletBC(int block) const { int l; return blockHasDirichletBC(block, l); }
    bool blockHasDirichletBC(int block, int& loc) const {
        for (size_t i = 0; i < vecBlockID_.size(); ++i)
            if (vecBlockID_[i] == block && vecBCType_[i].compare(0, 9, "Dirichlet") == 0) { loc = (int)i; return true; }
        return false;
    }
    // BCBuilder::set = setSystem + setRHS (BCBuilder_def.hpp:84-90).  The device call does both for
    // the diagonal block; the host evaluates the user function at every flagged unique node.
    void set(const BlockMatrixPtr_Type& blockMatrix, const BlockMultiVectorPtr_Type& blockMV, double t = 0.) const {
        TEUCHOS_TEST_FOR_EXCEPTION(blockMV->getNumVectors() > 1, std::runtime_error, "BCBuilder::setRHS() only for getNumVectors == 1.");
        for (UN block = 0; block < blockMatrix->size(); ++block) {
            int loc0;
            if (!blockHasDirichletBC((int)block, loc0)) continue;
            if (blockMatrix->size() > 1) {
                // merged block system on the device: unit row on the merged row, i.e. setLocalRowOne on the diagonal block and
                // setLocalRowZero on the off-diagonal blocks (BCBuilder_def.hpp:653-707), rhs <- boundary value
                setMerged(blockMatrix, blockMV, (int)block, t);
                continue;
            }
            auto A = blockMatrix->getBlock(block, block);
            TEUCHOS_TEST_FOR_EXCEPTION(!A->isResident(), std::runtime_error, "BCBuilder: the matrix block is not resident on the device");
            auto dom = vecDomain_.at(loc0);
            const int dofs = vecDofs_.at(loc0), dim = (int)dom->getDimension();
            vec_int_ptr_Type flags = dom->getBCFlagUnique();
            vec2D_dbl_ptr_Type pts = dom->getPointsUnique();
            std::vector<int32_t> nodes, mask;
            std::vector<double> values;
            vec_dbl_Type result(dofs, 0.), point(dim, 0.);
            // owned nodes, then (several ranks) the row ghosts: their rows exist on this rank and get the same treatment
            const auto& rgRep = dom->rowGhostRepeatedIds();
            const auto& rgFlag = dom->rowGhostFlags();
            const auto& xyzRep = dom->xyzRepeated();
            const size_t nOwned = flags->size();
            for (size_t i = 0; i < nOwned + rgRep.size(); ++i) {
                int loc;
                const int flag = i < nOwned ? (*flags)[i] : rgFlag[i - nOwned];
                if (!findFlag(flag, (int)block, loc)) continue;
                const std::string& ty = vecBCType_[loc];
                if (ty.compare(0, 9, "Dirichlet") != 0) continue;
                for (int d = 0; d < dim; ++d) point[d] = i < nOwned ? (*pts)[i][d] : xyzRep[(size_t)rgRep[i - nOwned] * dim + d];
                for (int d = 0; d < dofs; ++d) result[d] = d < dim ? point[d] : 0.;   // :136-138
                vecBC_func_[loc](point.data(), result.data(), t, vecBC_Parameters_[loc].data());
                nodes.push_back((int32_t)i);
                for (int d = 0; d < dofs; ++d) {
                    const bool on = ty == "Dirichlet" || (ty == "Dirichlet_X" && d == 0) || (ty == "Dirichlet_Y" && d == 1) ||
                                    (ty == "Dirichlet_Z" && d == 2) || (ty == "Dirichlet_X_Y" && d != 2) ||
                                    (ty == "Dirichlet_X_Z" && d != 1) || (ty == "Dirichlet_Y_Z" && d != 0);
                    mask.push_back(on ? 1 : 0);
                    values.push_back(result[d]);
                }
            }
            fedd_ctx* ctx = A->device()->ctx;
            // push the current rhs, apply rows + rhs on the device, pull the rhs back
            feddCheck(fedd_rhs_set(ctx, blockMV->getBlockNonConst(block)->raw().data()), "fedd_rhs_set");
            feddCheck(fedd_dirichlet_nodes(ctx, (int64_t)nodes.size(), nodes.data(), mask.data(), values.data()), "fedd_dirichlet_nodes");
            feddCheck(fedd_rhs_get(ctx, blockMV->getBlockNonConst(block)->raw().data()), "fedd_rhs_get");
            A->invalidateHost();
        }
    }
private:
    void setMerged(const BlockMatrixPtr_Type& blockMatrix, const BlockMultiVectorPtr_Type& blockMV, int block, double t) const {
        auto dev = blockMatrix->mergedDevice();
        TEUCHOS_TEST_FOR_EXCEPTION(dev.is_null(), std::runtime_error, "BCBuilder: the block system has not been merged on the device");
        int64_t rowOffset = 0;
        for (int b = 0; b < block; ++b) rowOffset += (int64_t)blockMV->getBlock(b)->getLocalLength();
        std::vector<int32_t> rows;
        std::vector<double> values;
        for (size_t k = 0; k < vecFlag_.size(); ++k) {
            if (vecBlockID_[k] != block || vecBCType_[k].compare(0, 9, "Dirichlet") != 0) continue;
            auto dom = vecDomain_[k];
            const int dofs = vecDofs_[k], dim = (int)dom->getDimension();
            vec_int_ptr_Type flags = dom->getBCFlagUnique();
            vec2D_dbl_ptr_Type pts = dom->getPointsUnique();
            const std::string& ty = vecBCType_[k];
            vec_dbl_Type result(dofs, 0.), point(dim, 0.);
            for (size_t i = 0; i < flags->size(); ++i) {
                if ((*flags)[i] != vecFlag_[k]) continue;
                for (int d = 0; d < dim; ++d) point[d] = (*pts)[i][d];
                for (int d = 0; d < dofs; ++d) result[d] = d < dim ? (*pts)[i][d] : 0.;
                vecBC_func_[k](point.data(), result.data(), t, vecBC_Parameters_[k].data());
                for (int d = 0; d < dofs; ++d) {
                    const bool on = ty == "Dirichlet" || (ty == "Dirichlet_X" && d == 0) || (ty == "Dirichlet_Y" && d == 1) ||
                                    (ty == "Dirichlet_Z" && d == 2) || (ty == "Dirichlet_X_Y" && d != 2) ||
                                    (ty == "Dirichlet_X_Z" && d != 1) || (ty == "Dirichlet_Y_Z" && d != 0);
                    if (!on) continue;
                    rows.push_back((int32_t)(rowOffset + (int64_t)i * dofs + d));
                    values.push_back(result[d]);
                }
            }
        }
        // the merged rhs = the blocks' right-hand sides one after the other
        std::vector<double> rhs;
        for (UN b = 0; b < blockMV->size(); ++b) rhs.insert(rhs.end(), blockMV->getBlock(b)->raw().begin(), blockMV->getBlock(b)->raw().end());
        feddCheck(fedd_rhs_set(dev->ctx, rhs.data()), "fedd_rhs_set");
        feddCheck(fedd_dirichlet_rows(dev->ctx, (int64_t)rows.size(), rows.data(), values.data()), "fedd_dirichlet_rows");
        feddCheck(fedd_rhs_get(dev->ctx, rhs.data()), "fedd_rhs_get");
        size_t off = 0;
        for (UN b = 0; b < blockMV->size(); ++b) {
            auto& dst = blockMV->getBlockNonConst(b)->raw();
            std::copy(rhs.begin() + off, rhs.begin() + off + dst.size(), dst.begin());
            off += dst.size();
        }
    }
    std::vector<BC_func_Type> vecBC_func_;
    std::vector<int> vecFlag_, vecBlockID_, vecDofs_;
    std::vector<DomainPtr_Type> vecDomain_;
    std::vector<std::string> vecBCType_;
    std::vector<vec_dbl_Type> vecBC_Parameters_;
};

template <class SC, class LO, class GO, class NO>
class Problem;

// PreconditionerOperator (feddlib/problems/Solver/PreconditionerOperator_decl.hpp:24-125): the reference's base class for a
// preconditioner that the user hands to the iterative solver through Problem::setPreconditionerThyraFromLinOp
// (Problem_def.hpp:397-399, Preconditioner_def.hpp:96-102) -- there a Thyra::LinearOpBase with a pure applyImpl, here the
// same contract on the facade's MultiVector:  Y = alpha * M^-1 X + beta * Y  on the entries of the unique map.
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class PreconditionerOperator {
public:
    typedef MultiVector<SC, LO, GO, NO> MultiVector_Type;
    virtual ~PreconditionerOperator() {}
    void apply(const MultiVector_Type& X, MultiVector_Type& Y, SC alpha = 1., SC beta = 0.) const { applyImpl(X, Y, alpha, beta); }
    virtual std::string description() const { return "FEDD::PreconditionerOperator"; }
protected:
    virtual void applyImpl(const MultiVector_Type& X, MultiVector_Type& Y, const SC alpha, const SC beta) const = 0;
};

// the library's Schwarz preconditioner behind that interface.  Every application crosses the host
// boundary (fedd_schwarz_apply takes and returns host vectors); the resident path is LinearSolver's default.
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class DeviceSchwarzOperator : public PreconditionerOperator<SC, LO, GO, NO> {
public:
    typedef MultiVector<SC, LO, GO, NO> MultiVector_Type;
    // the matrix resident in `dev` must be final (assembled, boundary conditions set); parameters as in parametersPrec.xml
    DeviceSchwarzOperator(const DeviceContextPtr& dev, int overlap = 1, const std::string& combine = "Restricted", bool twoLevel = false,
                          int coarseKind = FEDD_COARSE_GDSW)
        : dev_(dev) {
        const int cmb = combine == "Averaging" ? FEDD_COMBINE_AVERAGING : (combine == "Full" ? FEDD_COMBINE_FULL : FEDD_COMBINE_RESTRICTED);
        feddCheck(fedd_schwarz_setup(dev_->ctx, overlap, cmb, twoLevel ? 1 : 0, twoLevel ? coarseKind : 0), "fedd_schwarz_setup");
    }
    std::string description() const override { return "FEDD::DeviceSchwarzOperator (overlapping Schwarz on the GPU)"; }
protected:
    void applyImpl(const MultiVector_Type& X, MultiVector_Type& Y, const SC alpha, const SC beta) const override {
        z_.resize(X.raw().size());
        feddCheck(fedd_schwarz_apply(dev_->ctx, X.raw().data(), z_.data()), "fedd_schwarz_apply");
        auto& y = Y.raw();
        for (size_t i = 0; i < y.size(); ++i) y[i] = alpha * z_[i] + (beta == 0. ? 0. : beta * y[i]);
    }
private:
    DeviceContextPtr dev_;
    mutable std::vector<SC> z_;
};

// LinearSolver::solve -> solveMonolithic (LinearSolver_def.hpp:23-135): GMRES + one-level Schwarz on
// the device, configured from the same ParameterList entries the reference hands to Stratimikos.
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class LinearSolver {
public:
    typedef Problem<SC, LO, GO, NO> Problem_Type;
    typedef BlockMultiVector<SC, LO, GO, NO> BlockMultiVector_Type;
    typedef Teuchos::RCP<BlockMultiVector_Type> BlockMultiVectorPtr_Type;
    int solve(Problem_Type* problem, BlockMultiVectorPtr_Type rhs, std::string type = "Monolithic");
    double lastRelativeResidual = 0.;
private:
    typedef MultiVector<SC, LO, GO, NO> MV;
    static double dot(const MV& a, const MV& b) {
        double s = 0.;
        const auto& x = a.raw();
        const auto& y = b.raw();
        for (size_t i = 0; i < x.size(); ++i) s += x[i] * y[i];
        if (!a.getMap()->getComm().is_null()) a.getMap()->getComm()->sumAll(&s, 1);
        return s;
    }
    static int hostGmres(Matrix<SC, LO, GO, NO>& A, const PreconditionerOperator<SC, LO, GO, NO>& M, const MV& b, MV& x, double tol,
                         int maxIt, int m, double& rel) {
        auto map = b.getMap();
        const double bnorm = b.norm2();
        x.putScalar(0.);
        rel = 0.;
        if (bnorm == 0.) return 0;
        MV r(map), z(map), w(map);
        r.update(1., b, 0.);
        double beta = bnorm;
        int its = 0;
        while (its < maxIt) {
            std::vector<MV> V;
            V.reserve((size_t)m + 1);
            V.emplace_back(map);
            V[0].update(1. / beta, r, 0.);
            std::vector<std::vector<double>> H((size_t)m, std::vector<double>((size_t)m + 1, 0.));
            std::vector<double> cs((size_t)m, 0.), sn((size_t)m, 0.), g((size_t)m + 1, 0.);
            g[0] = beta;
            int k = 0;
            for (; k < m && its < maxIt; ++k) {
                M.apply(V[k], z);
                A.apply(z, w);
                for (int pass = 0; pass < 2; ++pass) {       // classical Gram-Schmidt, twice
                    std::vector<double> h((size_t)k + 1);
                    for (int i = 0; i <= k; ++i) h[i] = dot(V[i], w);
                    for (int i = 0; i <= k; ++i) {
                        w.update(-h[i], V[i], 1.);
                        H[k][i] += h[i];
                    }
                }
                const double hn = w.norm2();
                H[k][k + 1] = hn;
                for (int i = 0; i < k; ++i) {
                    const double t = cs[i] * H[k][i] + sn[i] * H[k][i + 1];
                    H[k][i + 1] = -sn[i] * H[k][i] + cs[i] * H[k][i + 1];
                    H[k][i] = t;
                }
                const double d = std::hypot(H[k][k], H[k][k + 1]);
                cs[k] = H[k][k] / d;
                sn[k] = H[k][k + 1] / d;
                H[k][k] = d;
                H[k][k + 1] = 0.;
                g[k + 1] = -sn[k] * g[k];
                g[k] = cs[k] * g[k];
                ++its;
                rel = std::fabs(g[k + 1]) / bnorm;
                if (rel <= tol || hn == 0.) { ++k; break; }
                V.emplace_back(map);
                V[k + 1].update(1. / hn, w, 0.);
            }
            std::vector<double> y((size_t)k, 0.);
            for (int i = k - 1; i >= 0; --i) {
                double sum = g[i];
                for (int j = i + 1; j < k; ++j) sum -= H[j][i] * y[j];
                y[i] = sum / H[i][i];
            }
            w.putScalar(0.);
            for (int i = 0; i < k; ++i) w.update(y[i], V[i], 1.);
            M.apply(w, z);
            x.update(1., z, 1.);
            if (rel <= tol) break;
            A.apply(x, w);
            r.update(1., b, 0.);
            r.update(-1., w, 1.);
            beta = r.norm2();
            rel = beta / bnorm;
            if (rel <= tol) break;
        }
        return its;
    }
};

template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class Problem {
public:
    typedef Domain<SC, LO, GO, NO> Domain_Type;
    typedef Teuchos::RCP<const Domain_Type> DomainConstPtr_Type;
    typedef std::vector<DomainConstPtr_Type> DomainConstPtr_vec_Type;
    typedef Matrix<SC, LO, GO, NO> Matrix_Type;
    typedef Teuchos::RCP<Matrix_Type> MatrixPtr_Type;
    typedef BlockMatrix<SC, LO, GO, NO> BlockMatrix_Type;
    typedef Teuchos::RCP<BlockMatrix_Type> BlockMatrixPtr_Type;
    typedef MultiVector<SC, LO, GO, NO> MultiVector_Type;
    typedef Teuchos::RCP<MultiVector_Type> MultiVectorPtr_Type;
    typedef BlockMultiVector<SC, LO, GO, NO> BlockMultiVector_Type;
    typedef Teuchos::RCP<BlockMultiVector_Type> BlockMultiVectorPtr_Type;
    typedef BCBuilder<SC, LO, GO, NO> BC_Type;
    typedef Teuchos::RCP<const BC_Type> BCConstPtr_Type;
    typedef FE<SC, LO, GO, NO> FEFac_Type;
    typedef Teuchos::RCP<FEFac_Type> FEFacPtr_Type;
    typedef Teuchos::Comm<int> Comm_Type;
    typedef Teuchos::RCP<const Comm_Type> CommConstPtr_Type;

    Problem(ParameterListPtr_Type& parameterList, CommConstPtr_Type comm)
        : dim_(-1), comm_(comm), verbose_(comm->getRank() == 0), parameterList_(parameterList), feFactory_(new FEFac_Type()) {}
    virtual ~Problem() {}
    virtual void info() = 0;
    void infoProblem() {
        if (verbose_) {
            std::cout << "\t ### Problem Information ###" << std::endl;
            for (size_t i = 0; i < domainPtr_vec_.size(); ++i)
                std::cout << "\t ### Variable " << variableName_vec_[i] << ": dofs/node " << dofsPerNode_vec_[i] << ", FE " << domain_FEType_vec_[i] << std::endl;
        }
    }
    void addVariable(const DomainConstPtr_Type& domain, std::string FEType, std::string name, int dofsPerNode) {
        domainPtr_vec_.push_back(domain); domain_FEType_vec_.push_back(FEType); variableName_vec_.push_back(name);
        dofsPerNode_vec_.push_back(dofsPerNode); feFactory_->addFE(domain);
    }
    void addRhsFunction(RhsFunc_Type func) { rhsFuncVec_.push_back(func); }
    RhsFunc_Type& getRhsFunction(int i) { return rhsFuncVec_.at(i); }
    virtual void assemble(std::string type = "") const = 0;

    void assembleSourceTerm(double time = 0.) const {                 // Problem_def.hpp:170-181
        TEUCHOS_TEST_FOR_EXCEPTION(sourceTerm_.is_null(), std::runtime_error, "Initialize source term before you assemble it - sourceTerm pointer is null");
        sourceTerm_->putScalar(0.);
        std::string sourceType = parameterList_->sublist("Parameter").get("Source Type", "volume");
        TEUCHOS_TEST_FOR_EXCEPTION(sourceType != "volume", std::logic_error, "only volume source terms are built");
        assembleVolumeTerm(time);
    }
    void assembleVolumeTerm(double time) const {                      // Problem_def.hpp:184-216
        for (UN i = 0; i < sourceTerm_->size(); ++i) {
            if (i < rhsFuncVec_.size() && rhsFuncVec_[i]) {
                vec_dbl_Type funcParameter(1, 0.);
                funcParameter[0] = time;
                for (double p : parasSourceFunc_) funcParameter.push_back(p);
                const std::string type = getDofsPerNode((int)i) > 1 ? "Vector" : "Scalar";
                feFactory_->assemblyRHS(dim_, domain_FEType_vec_.at(i), sourceTerm_->getBlockNonConst(i), type, rhsFuncVec_[i], funcParameter);
            }
        }
    }
    bool hasSourceTerm() const { return !sourceTerm_.is_null(); }
    int solve(BlockMultiVectorPtr_Type rhs = Teuchos::null) {         // Problem_def.hpp:257-295
        if (verbose_) std::cout << "-- Solve System ..." << std::endl;
        LinearSolver<SC, LO, GO, NO> linSolver;
        std::string type = parameterList_->sublist("General").get("Preconditioner Method", "Monolithic");
        int its = linSolver.solve(this, rhs, type);
        lastRelativeResidual_ = linSolver.lastRelativeResidual;
        if (verbose_) std::cout << " done. -- " << its << " iterations, relative residual " << lastRelativeResidual_ << std::endl;
        return its;
    }
    // Problem::setPreconditionerThyraFromLinOp (Problem_def.hpp:397-399): a user-supplied preconditioner; the solve then
    // runs the host-side GMRES below (the stand-in for Belos in this build) on Matrix::apply and this operator
    typedef PreconditionerOperator<SC, LO, GO, NO> PrecOp_Type;
    void setPreconditionerThyraFromLinOp(const Teuchos::RCP<PrecOp_Type>& op) { userPrec_ = op; }
    Teuchos::RCP<PrecOp_Type> getUserPreconditioner() const { return userPrec_; }
    void addBoundaries(const BCConstPtr_Type& bcFactory) { bcFactory_ = bcFactory; }
    void setBoundaries(double time = .0) const {                      // Problem_def.hpp:298-304
        TEUCHOS_TEST_FOR_EXCEPTION(bcFactory_.is_null(), std::runtime_error, "No boundary conditions added.");
        bcFactory_->set(system_, rhs_, time);
    }
    void initializeProblem(int nmbVectors = 1) {                      // Problem_def.hpp:134-139
        system_.reset(new BlockMatrix_Type((UN)domainPtr_vec_.size()));
        initializeVectors(nmbVectors);
    }
    void initializeVectors(int nmbVectors = 1) {                      // Problem_def.hpp:332-360
        const UN size = (UN)domainPtr_vec_.size();
        solution_.reset(new BlockMultiVector_Type(size));
        rhs_.reset(new BlockMultiVector_Type(size));
        sourceTerm_.reset(new BlockMultiVector_Type(size));
        for (UN i = 0; i < size; ++i) {
            auto map = dofsPerNode_vec_[i] > 1 ? domainPtr_vec_[i]->getMapVecFieldUnique() : domainPtr_vec_[i]->getMapUnique();
            solution_->addBlock(Teuchos::rcp(new MultiVector_Type(map, nmbVectors)), i);
            rhs_->addBlock(Teuchos::rcp(new MultiVector_Type(map, nmbVectors)), i);
            sourceTerm_->addBlock(Teuchos::rcp(new MultiVector_Type(map, nmbVectors)), i);
        }
    }
    BlockMultiVectorPtr_Type getRhs() const { return rhs_; }
    BlockMultiVectorPtr_Type getSolution() { return solution_; }
    BlockMatrixPtr_Type getSystem() const { return system_; }
    bool getVerbose() const { return verbose_; }
    DomainConstPtr_Type getDomain(int i) const { return domainPtr_vec_.at(i); }
    std::string getFEType(int i) const { return domain_FEType_vec_.at(i); }
    std::string getVariableName(int i) const { return variableName_vec_.at(i); }
    int getDofsPerNode(int i) const { return dofsPerNode_vec_.at(i); }
    ParameterListPtr_Type getParameterList() const { return parameterList_; }
    void addToRhs(BlockMultiVectorPtr_Type x) const { rhs_->update(1., *x, 1.); }   // Problem_def.hpp:450-454
    BlockMultiVectorPtr_Type getSourceTerm() { return sourceTerm_; }
    CommConstPtr_Type getComm() const { return comm_; }
    void addParemeterRhs(double para) { parasSourceFunc_.push_back(para); }
    double getLastRelativeResidual() const { return lastRelativeResidual_; }

    int dim_;
    mutable CommConstPtr_Type comm_;
    mutable BlockMatrixPtr_Type system_;
    mutable BlockMultiVectorPtr_Type rhs_;
    mutable BlockMultiVectorPtr_Type solution_;
    bool verbose_;
protected:
    mutable ParameterListPtr_Type parameterList_;
    mutable DomainConstPtr_vec_Type domainPtr_vec_;
    std::vector<std::string> domain_FEType_vec_, variableName_vec_;
    mutable BCConstPtr_Type bcFactory_;
    FEFacPtr_Type feFactory_;
    std::vector<int> dofsPerNode_vec_;
    mutable BlockMultiVectorPtr_Type sourceTerm_;
    std::vector<RhsFunc_Type> rhsFuncVec_;
    vec_dbl_Type parasSourceFunc_;
    double lastRelativeResidual_ = 0.;
    Teuchos::RCP<PrecOp_Type> userPrec_;
};

template <class SC, class LO, class GO, class NO>
int LinearSolver<SC, LO, GO, NO>::solve(Problem_Type* problem, BlockMultiVectorPtr_Type rhs, std::string type) {
    TEUCHOS_TEST_FOR_EXCEPTION(type != "Monolithic" && type != "MonolithicConstPrec", std::logic_error,
                               "Unknown solver type; only the monolithic path (LinearSolver_def.hpp:72-135) is built.");
    auto system = problem->getSystem();
    const bool blocks = system->size() > 1;
    TEUCHOS_TEST_FOR_EXCEPTION(blocks && system->mergedDevice().is_null(), std::logic_error,
                               "block system: it has not been merged on the device (BlockMatrix::merge, done by the problem's assemble)");
    if (!blocks) TEUCHOS_TEST_FOR_EXCEPTION(!system->getBlock(0, 0)->isResident(), std::runtime_error, "solve: the system matrix is not resident on the device");
    fedd_ctx* ctx = blocks ? system->mergedDevice()->ctx : system->getBlock(0, 0)->device()->ctx;
    auto pl = problem->getParameterList();
    // same keys the reference's XML files use (laplace/parametersSolver.xml, parametersPrec.xml)
    auto& belos = pl->sublist("ThyraSolver").sublist("Linear Solver Types").sublist("Belos");
    const std::string solverType = belos.get("Solver Type", "Block GMRES");
    auto& gm = belos.sublist("Solver Types").sublist(solverType);
    const double tol = gm.get("Convergence Tolerance", 1e-8);
    const int maxIt = gm.get("Maximum Iterations", 100);
    const int numBlocks = gm.get("Num Blocks", 100);
    auto& frosch = pl->sublist("ThyraPreconditioner").sublist("Preconditioner Types").sublist("FROSch");
    const std::string precType = pl->sublist("ThyraPreconditioner").get("Preconditioner Type", "FROSch");
    const int overlap = frosch.get("Overlap", 1);
    auto& ovl = frosch.sublist("AlgebraicOverlappingOperator");
    std::string combine = ovl.get("Combine Values in Overlap", "");
    if (combine.empty()) combine = ovl.get("Overlapping Operator Combination", "Restricted");
    const int cmb = combine == "Averaging" ? FEDD_COMBINE_AVERAGING : (combine == "Full" ? FEDD_COMBINE_FULL : FEDD_COMBINE_RESTRICTED);
    const bool usePrec = precType != "None";
    if (usePrec && type != "MonolithicConstPrec") {
        // "TwoLevel" = true (parametersPrec.xml:17) switches the coarse level on.  The coarse space is
        // this library's lattice space, not FROSch's GDSW: say so.
        const bool twoLevel = frosch.get("TwoLevel", false);
        // "CoarseOperator Type" (parametersPrec.xml:23): GDSWCoarseOperator / RGDSWCoarseOperator -> the library's GDSW /
        // RGDSW level on the coarse lattice; "Q1" = lattice hat functions; anything else (IPOUHarmonicCoarseOperator, the
        // third value the reference's XML files list) is not built: an error, not a silent substitute
        const std::string coarseType = frosch.get("CoarseOperator Type", "GDSWCoarseOperator");
        const int coarseKind = coarseType == "Q1" ? FEDD_COARSE_Q1 : (coarseType == "RGDSWCoarseOperator" ? FEDD_COARSE_RGDSW : FEDD_COARSE_GDSW);
        TEUCHOS_TEST_FOR_EXCEPTION(twoLevel && coarseType != "GDSWCoarseOperator" && coarseType != "RGDSWCoarseOperator" && coarseType != "Q1",
                                   std::logic_error, "CoarseOperator Type \"" + coarseType + "\" is not built (GDSWCoarseOperator, RGDSWCoarseOperator, Q1 are)");
        const int target = frosch.get("Subdomain Nodes", 0);   // 0 = the library's default (27 / dofs per node)
        feddCheck(fedd_schwarz_set_target(ctx, target, 1.0), "fedd_schwarz_set_target");
        feddCheck(fedd_schwarz_set_coarse(ctx, frosch.get("Coarse Cells", 0.0)), "fedd_schwarz_set_coarse");
        // merged block systems: monolithic one-level Schwarz on the large-subdomain path; the coarse level takes
        // node-interleaved systems only (said once)
        if (blocks && twoLevel && problem->getVerbose())
            std::cout << "-- note: the coarse level is not built for merged block systems; running one level --" << std::endl;
        const bool two = twoLevel && !blocks;
        // rotations in the null space of the coarse space: "Rotations" of the coarse operator's block 1
        // (steadyLinElas/parametersPrec.xml:100), which FROSch can only build from the node coordinates the reference hands over
        // with "Use node lists" (Preconditioner_def.hpp:266, 353-381; default true).  The library ignores the switch for
        // scalar problems (a rotation needs dofs = dim).
        if (two && coarseKind != FEDD_COARSE_Q1) {
            const bool nodeLists = pl->get("Use node lists", true);
            const bool rotations = frosch.sublist(coarseType).sublist("Blocks").sublist("1").get("Rotations", false);
            feddCheck(fedd_set_option(ctx, "gdsw_rotations", nodeLists && rotations ? 1.0 : 0.0), "fedd_set_option(gdsw_rotations)");
        }
        feddCheck(fedd_schwarz_setup(ctx, overlap, cmb, two ? 1 : 0, two ? coarseKind : 0), "fedd_schwarz_setup");
    }
    auto b = rhs.is_null() ? problem->getRhs() : rhs;
    auto x = problem->getSolution();
    int its = 0;
    double rel = 0.;
    // "Zero Initial Guess" (LinearSolver_def.hpp:76-78): true clears the solution vector, false keeps it as x_0.
    // "Level Combination" = "Multiplicative" (:98-104): one coarse-only application of the preconditioner to the right-hand
    // side goes into the solution vector before the solve, which then starts from it
    const bool zeroGuess = pl->get("Zero Initial Guess", true);
    if (zeroGuess) x->putScalar(0.);
    const bool multiplicative = usePrec && std::string(frosch.get("Level Combination", "Additive")) == "Multiplicative";
    TEUCHOS_TEST_FOR_EXCEPTION(multiplicative && (blocks || !frosch.get("TwoLevel", false)), std::logic_error,
                               "Level Combination = Multiplicative needs the coarse level (TwoLevel = true, single-block system)");
    if (!problem->getUserPreconditioner().is_null()) {
        // INTEGRATION.md 3(b), "keep the iterative solver, plug in operator and preconditioner": right-preconditioned
        // GMRES(m) on the host -- what Belos' Block GMRES does with Thyra operators (LinearSolver_def.hpp:72-135), two-pass
        // classical Gram-Schmidt like its ICGS manager -- with A = Matrix::apply (fedd_spmv) and M^-1 = the user's operator
        TEUCHOS_TEST_FOR_EXCEPTION(blocks, std::logic_error, "user preconditioner operators: single-block systems in this build");
        its = hostGmres(*system->getBlock(0, 0), *problem->getUserPreconditioner(), *b->getBlock(0), *x->getBlockNonConst(0), tol, maxIt,
                        numBlocks, rel);
        lastRelativeResidual = rel;
        return its;
    }
    if (!blocks) {
        if (multiplicative && problem->getUserPreconditioner().is_null())
            feddCheck(fedd_schwarz_coarse_apply(ctx, b->getBlock(0)->raw().data(), x->getBlockNonConst(0)->raw().data()), "fedd_schwarz_coarse_apply");
        if (multiplicative || !zeroGuess)
            feddCheck(fedd_gmres_x0(ctx, b->getBlock(0)->raw().data(), x->getBlockNonConst(0)->raw().data(), tol, maxIt, numBlocks,
                                    usePrec ? 1 : 0, &its, &rel), "fedd_gmres_x0");
        else
            feddCheck(fedd_gmres(ctx, b->getBlock(0)->raw().data(), x->getBlockNonConst(0)->raw().data(), tol, maxIt, numBlocks,
                                 usePrec ? 1 : 0, &its, &rel), "fedd_gmres");
    } else {       // merged vectors = the blocks one after the other (BlockMap::merge, BlockMap_def.hpp:55-80)
        std::vector<double> bb, xx;
        for (UN k = 0; k < b->size(); ++k) bb.insert(bb.end(), b->getBlock(k)->raw().begin(), b->getBlock(k)->raw().end());
        xx.assign(bb.size(), 0.);
        feddCheck(fedd_gmres(ctx, bb.data(), xx.data(), tol, maxIt, numBlocks, usePrec ? 1 : 0, &its, &rel), "fedd_gmres");
        size_t off = 0;
        for (UN k = 0; k < x->size(); ++k) {
            auto& dst = x->getBlockNonConst(k)->raw();
            std::copy(xx.begin() + off, xx.begin() + off + dst.size(), dst.begin());
            off += dst.size();
        }
    }
    lastRelativeResidual = rel;
    return its;
}

// Laplace (feddlib/problems/specific/Laplace_def.hpp:16-60)
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class Laplace : public Problem<SC, LO, GO, NO> {
public:
    typedef Problem<SC, LO, GO, NO> Problem_Type;
    typedef typename Problem_Type::DomainConstPtr_Type DomainConstPtr_Type;
    typedef typename Problem_Type::Matrix_Type Matrix_Type;
    typedef typename Problem_Type::MatrixPtr_Type MatrixPtr_Type;
    Laplace(const DomainConstPtr_Type& domain, std::string FEType, ParameterListPtr_Type parameterList, bool vectorLaplace = false)
        : Problem_Type(parameterList, domain->getComm()), vectorLaplace_(vectorLaplace) {
        this->addVariable(domain, FEType, "u", vectorLaplace ? (int)domain->getDimension() : 1);
        this->dim_ = (int)this->getDomain(0)->getDimension();
    }
    void info() override { this->infoProblem(); }
    void assemble(std::string type = "") const override {
        (void)type;
        if (this->verbose_) std::cout << "-- Assembly Laplace ... " << std::flush;
        MatrixPtr_Type A;
        if (vectorLaplace_) {
            A = Teuchos::rcp(new Matrix_Type(this->domainPtr_vec_.at(0)->getMapVecFieldUnique(), this->getDomain(0)->getApproxEntriesPerRow()));
            this->feFactory_->assemblyLaplaceVecField(this->dim_, this->domain_FEType_vec_.at(0), 2, A);
        } else {
            A = Teuchos::rcp(new Matrix_Type(this->domainPtr_vec_.at(0)->getMapUnique(), this->getDomain(0)->getApproxEntriesPerRow()));
            this->feFactory_->assemblyLaplace(this->dim_, this->domain_FEType_vec_.at(0), 2, A);
        }
        this->system_->addBlock(A, 0, 0);
        this->assembleSourceTerm(0.);
        this->addToRhs(this->sourceTerm_);
        if (this->verbose_) std::cout << "done -- " << std::endl;
    }
    MatrixPtr_Type getMassMatrix() const {
        MatrixPtr_Type A = Teuchos::rcp(new Matrix_Type(this->domainPtr_vec_.at(0)->getMapUnique(), this->getDomain(0)->getApproxEntriesPerRow()));
        this->feFactory_->assemblyMass(this->dim_, this->domain_FEType_vec_.at(0), "Scalar", A);
        return A;
    }
private:
    bool vectorLaplace_;
};

// LinElas (feddlib/problems/specific/LinElas_def.hpp:64-99): lambda, E from mu, nu at :76-77
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class LinElas : public Problem<SC, LO, GO, NO> {
public:
    typedef Problem<SC, LO, GO, NO> Problem_Type;
    typedef typename Problem_Type::DomainConstPtr_Type DomainConstPtr_Type;
    typedef typename Problem_Type::Matrix_Type Matrix_Type;
    typedef typename Problem_Type::MatrixPtr_Type MatrixPtr_Type;
    LinElas(const DomainConstPtr_Type& domain, std::string FEType, ParameterListPtr_Type parameterList)
        : Problem_Type(parameterList, domain->getComm()) {
        this->addVariable(domain, FEType, "d_s", (int)domain->getDimension());
        this->dim_ = (int)this->getDomain(0)->getDimension();
    }
    void info() override { this->infoProblem(); }
    void assemble(std::string type = "") const override {
        (void)type;
        if (this->verbose_) std::cout << "-- Assembly linear elasticity ... " << std::flush;
        const double mu = this->parameterList_->sublist("Parameter").get("Mu", 2.0e6);
        const double nu = this->parameterList_->sublist("Parameter").get("Poisson Ratio", 0.4);
        const double E = mu * 2. * (1. + nu);
        const double lambda = nu * E / ((1. + nu) * (1. - 2. * nu));
        MatrixPtr_Type K = Teuchos::rcp(new Matrix_Type(this->getDomain(0)->getMapVecFieldUnique(), this->getDomain(0)->getApproxEntriesPerRow()));
        this->feFactory_->assemblyLinElasXDim(this->dim_, this->getDomain(0)->getFEType(), K, lambda, mu);
        this->system_->addBlock(K, 0, 0);
        this->assembleSourceTerm(0.);
        this->addToRhs(this->sourceTerm_);
        if (this->verbose_) std::cout << "done -- " << std::endl;
    }
};

// Stokes (feddlib/problems/specific/Stokes_def.hpp:26-138): A = nu * vector Laplacian, B and B^T = -div blocks; the
// velocity domain is the P2 mesh built from the pressure domain's P1 mesh (its first nodes are the pressure nodes).
// The blocks live in slots of the velocity domain's device context and are merged there at the end of assemble()
// (BlockMatrix::merge at solve time in the reference); getBlock(i, j) gives host views of the blocks as assembled.
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class Stokes : public Problem<SC, LO, GO, NO> {
public:
    typedef Problem<SC, LO, GO, NO> Problem_Type;
    typedef typename Problem_Type::DomainConstPtr_Type DomainConstPtr_Type;
    typedef typename Problem_Type::Matrix_Type Matrix_Type;
    typedef typename Problem_Type::MatrixPtr_Type MatrixPtr_Type;
    typedef typename Problem_Type::BlockMatrix_Type BlockMatrix_Type;
    Stokes(const DomainConstPtr_Type& domainVelocity, std::string FETypeVelocity, const DomainConstPtr_Type& domainPressure,
           std::string FETypePressure, ParameterListPtr_Type parameterList)
        : Problem_Type(parameterList, domainVelocity->getComm()) {
        this->addVariable(domainVelocity, FETypeVelocity, "u", (int)domainVelocity->getDimension());
        this->addVariable(domainPressure, FETypePressure, "p", 1);
        this->dim_ = (int)this->getDomain(0)->getDimension();
    }
    void info() override { this->infoProblem(); }
    void assemble(std::string type = "") const override {
        (void)type;
        if (this->verbose_) std::cout << "-- Assembly ... " << std::flush;
        const double viscosity = this->parameterList_->sublist("Parameter").get("Viscosity", 1.);
        TEUCHOS_TEST_FOR_EXCEPTION(this->parameterList_->sublist("Parameter").get("Symmetric gradient", false), std::logic_error,
                                   "assemblyStress (symmetric gradient) is not built");
        TEUCHOS_TEST_FOR_EXCEPTION(this->getFEType(1) != "P1", std::logic_error, "Stokes: P1 pressure in this build");
        auto domV = this->getDomain(0);
        auto dev = domV->device();
        MatrixPtr_Type A(new Matrix_Type(domV->getMapVecFieldUnique(), domV->getApproxEntriesPerRow()));
        MatrixPtr_Type BT(new Matrix_Type(domV->getMapVecFieldUnique(), this->getDomain(1)->getDimension() * this->getDomain(1)->getApproxEntriesPerRow()));
        auto pressureMap = this->getDomain(1)->getMapUnique();
        MatrixPtr_Type B(new Matrix_Type(pressureMap, domV->getDimension() * domV->getApproxEntriesPerRow()));
        if (this->verbose_) std::cout << " A ... " << std::flush;
        this->feFactory_->assemblyLaplaceVecField(this->dim_, this->domain_FEType_vec_.at(0), 2, A, true);
        A->resumeFill();
        feddCheck(fedd_matrix_scale(dev->ctx, -1, viscosity), "fedd_matrix_scale");        // A->scale(viscosity), Stokes_def.hpp:83
        feddCheck(fedd_matrix_store(dev->ctx, 0), "fedd_matrix_store");
        A->bindSlot(dev, 0);
        if (this->verbose_) std::cout << "B and B^T ... " << std::flush;
        this->feFactory_->assemblyDivAndDivT(this->dim_, this->getFEType(0), this->getFEType(1), 2, B, BT, domV->getMapVecFieldUnique(), pressureMap, true);
        B->resumeFill();
        BT->resumeFill();
        B->scale(-1.);
        BT->scale(-1.);
        A->fillComplete(domV->getMapVecFieldUnique(), domV->getMapVecFieldUnique());
        B->fillComplete(domV->getMapVecFieldUnique(), pressureMap);
        BT->fillComplete(pressureMap, domV->getMapVecFieldUnique());
        this->system_.reset(new BlockMatrix_Type(2));
        this->system_->addBlock(A, 0, 0);
        this->system_->addBlock(BT, 0, 1);
        this->system_->addBlock(B, 1, 0);
        int slotC = -1;
        if (this->getFEType(0) == "P1") {       // Stokes_def.hpp:98-105: P1/P1 needs the stabilisation block C = -1/nu * BD
            if (this->verbose_) std::cout << "C ... " << std::flush;
            MatrixPtr_Type C(new Matrix_Type(pressureMap, this->getDomain(1)->getApproxEntriesPerRow()));
            this->feFactory_->assemblyBDStabilization(this->dim_, "P1", C, true);
            C->resumeFill();
            feddCheck(fedd_matrix_scale(dev->ctx, -1, -1. / viscosity), "fedd_matrix_scale");   // C->scale(-1./viscosity)
            feddCheck(fedd_matrix_store(dev->ctx, 3), "fedd_matrix_store");
            C->bindSlot(dev, 3);
            C->fillComplete(pressureMap, pressureMap);
            this->system_->addBlock(C, 1, 1);
            slotC = 3;
        }
        // BlockMatrix::merge: system <- [A B^T; B C] on the device (C empty for P2/P1)
        feddCheck(fedd_block_merge(dev->ctx, 0, 2, 1, slotC), "fedd_block_merge");
        dev->generation++;
        this->system_->setMerged(dev);
        if (this->verbose_) std::cout << "done -- " << std::endl;
    }
};

// MeshPartitioner (feddlib/core/Mesh/MeshPartitioner_decl.hpp): reads the mesh named by "Mesh 1 Name" into the domain;
// on one rank no partitioning happens (MeshPartitioner_def.hpp:321-331).  The partitioner itself is in the library
// (fedd_mesh_partition*); the facade is one rank.
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class MeshPartitioner {
public:
    typedef Domain<SC, LO, GO, NO> Domain_Type;
    typedef Teuchos::RCP<Domain_Type> DomainPtr_Type;
    typedef std::vector<DomainPtr_Type> DomainPtrArray_Type;
    MeshPartitioner(DomainPtrArray_Type domains, ParameterListPtr_Type pL, std::string feType, int dimension)
        : domains_(domains), pList_(pL), feType_(feType), dim_(dimension) {}
    void readAndPartition(int volumeID = 10) {
        for (size_t i = 0; i < domains_.size(); ++i) {
            const std::string name = pList_->get("Mesh " + std::to_string(i + 1) + " Name", std::string("noName"));
            domains_[i]->readMeshFile(name, dim_, feType_, volumeID);
        }
    }
private:
    DomainPtrArray_Type domains_;
    ParameterListPtr_Type pList_;
    std::string feType_;
    int dim_;
};

// ExporterParaView (feddlib/core/General/ExporterParaView_decl.hpp:45-175, _def.hpp:484-601): nodal fields of the
// unique map over the mesh, as an XDMF file ParaView opens.  The reference writes the heavy data through EpetraExt::HDF5;
// HDF5 is not available here, so the DataItems use XDMF's raw binary format (little endian): <name>.xmf plus
// <name>.conn.bin / <name>.xyz.bin / <name>.<variable>.<step>.bin.  One rank (the facade's scope); P2 meshes are written
// with their vertex connectivity (mid-edge nodes stay in the point list).
template <class SC = default_sc, class LO = default_lo, class GO = default_go, class NO = default_no>
class ExporterParaView {
public:
    typedef Domain<SC, LO, GO, NO> Mesh_Type;
    typedef Teuchos::RCP<const Mesh_Type> MeshPtr_Type;
    typedef MultiVector<SC, LO, GO, NO> MultiVec_Type;
    typedef Teuchos::RCP<const MultiVec_Type> MultiVecConstPtr_Type;
    typedef Map<LO, GO, NO> Map_Type;
    typedef Teuchos::RCP<const Map_Type> MapConstPtr_Type;

    ExporterParaView() {}
    void setup(std::string filename, MeshPtr_Type mesh, std::string FEType, ParameterListPtr_Type parameterList = Teuchos::null) {
        setup(filename, mesh, FEType, 1, parameterList);
    }
    // The exported arrays are GLOBAL arrays in files of their own, as in the reference's HDF5 file (ExporterParaView_def.hpp:
    // 484-601: every rank writes its hyperslab): points and nodal values in global-id order, the connectivity in global ids.
    // Rank 0 creates each file at its final size, then every rank writes its entries in place -- its unique nodes at
    // position gid, the elements it exports (those whose vertex of smallest global id it owns: exactly one rank per element,
    // and that rank holds the element among its own or ghost elements) behind those of the lower ranks.  The files do not
    // depend on the number of ranks.
    void setup(std::string filename, MeshPtr_Type mesh, std::string FEType, int saveTimestep, ParameterListPtr_Type = Teuchos::null) {
        TEUCHOS_TEST_FOR_EXCEPTION(mesh.is_null(), std::runtime_error, "ExporterParaView::setup: null mesh");
        filename_ = filename; mesh_ = mesh; FEType_ = FEType; saveTimestep_ = std::max(1, saveTimestep);
        comm_ = mesh->getComm();
        const int size = comm_->getSize(), rank = comm_->getRank();
        const int dim = (int)mesh->getDimension(), nen = mesh->nodesPerElement(), nv = dim + 1;
        const auto& conn = mesh->connectivity();
        const size_t ne = conn.size() / nen;
        auto mapRep = mesh->getMapRepeated();
        auto mapUni = mesh->getMapUnique();
        nPoints_ = (size_t)mapUni->getGlobalNumElements();
        TEUCHOS_TEST_FOR_EXCEPTION(size > 1 && (GO)nPoints_ <= mapUni->getMaxAllGlobalIndex(), std::logic_error,
                                   "ExporterParaView: node ids are not 0 .. N-1");
        std::vector<char> owned(mapRep->getNodeNumElements(), 0);
        for (int32_t rl : mesh->uniqueLocalOfRepeated()) owned[rl] = 1;
        // connectivity in global ids of the elements this rank exports, vertices only
        std::vector<int32_t> c;
        c.reserve(ne * nv);
        for (size_t e = 0; e < ne; ++e) {
            int jmin = 0;
            for (int j = 1; j < nv; ++j)
                if (mapRep->getGlobalElement(conn[e * nen + j]) < mapRep->getGlobalElement(conn[e * nen + jmin])) jmin = j;
            if (!owned[conn[e * nen + jmin]]) continue;
            for (int j = 0; j < nv; ++j) c.push_back((int32_t)mapRep->getGlobalElement(conn[e * nen + j]));
        }
        std::vector<int64_t> counts = gatherCounts((int64_t)(c.size() / nv));
        int64_t before = 0, total = 0;
        for (int r = 0; r < size; ++r) {
            if (r < rank) before += counts[r];
            total += counts[r];
        }
        nElements_ = (size_t)total;
        createFile(filename_ + ".conn.bin", (size_t)total * nv * sizeof(int32_t));
        writeAt(filename_ + ".conn.bin", (size_t)before * nv * sizeof(int32_t), c.data(), c.size() * sizeof(int32_t));
        // points of the unique nodes, at their global positions
        uniGid_.assign(mapUni->gids().begin(), mapUni->gids().end());
        auto pts = mesh->getPointsUnique();
        std::vector<double> xyz(uniGid_.size() * 3, 0.0);
        for (size_t i = 0; i < uniGid_.size(); ++i)
            for (int d = 0; d < dim; ++d) xyz[i * 3 + d] = (*pts)[i][d];
        createFile(filename_ + ".xyz.bin", nPoints_ * 3 * sizeof(double));
        writeRecords(filename_ + ".xyz.bin", xyz.data(), 3 * sizeof(double));
        comm_->barrier();
        topology_ = dim == 3 ? "Tetrahedron" : "Triangle";
        nv_ = nv;
    }
    void addVariable(MultiVecConstPtr_Type& u, std::string varName, std::string varType, int dofPerNode,
                     MapConstPtr_Type mapUnique = Teuchos::null, MapConstPtr_Type mapUniqueLeading = Teuchos::null) {
        (void)mapUnique; (void)mapUniqueLeading;
        TEUCHOS_TEST_FOR_EXCEPTION(varType != "Scalar" && varType != "Vector", std::logic_error, "Unknown variable type for exporter.");
        TEUCHOS_TEST_FOR_EXCEPTION(u->getLocalLength() != uniGid_.size() * (size_t)dofPerNode, std::logic_error,
                                   "ExporterParaView::addVariable: vector length does not match the mesh");
        vars_.push_back({u, varName, varType, dofPerNode});
    }
    void save(double time) { save(time, 0.); }
    void save(double time, double dt) {
        (void)dt;
        if (timeIndex_ % saveTimestep_ == 0) {
            for (auto& v : vars_) {
                const auto& x = v.u->raw();
                const int comps = v.type == "Vector" ? 3 : 1;
                const size_t nl = uniGid_.size();
                std::vector<double> out(nl * comps, 0.0);
                for (size_t i = 0; i < nl; ++i)
                    for (int d = 0; d < v.dofs && d < comps; ++d) out[i * comps + d] = x[i * v.dofs + d];
                const std::string f = filename_ + "." + v.name + "." + std::to_string(nSaved_) + ".bin";
                createFile(f, nPoints_ * comps * sizeof(double));
                writeRecords(f, out.data(), comps * sizeof(double));
            }
            comm_->barrier();
            times_.push_back(time);
            ++nSaved_;
            if (comm_->getRank() == 0) writeXmf();
        }
        ++timeIndex_;
    }
    void closeExporter() { if (comm_.is_null() || comm_->getRank() == 0) writeXmf(); }
private:
    struct Var { MultiVecConstPtr_Type u; std::string name, type; int dofs; };
    std::vector<int64_t> gatherCounts(int64_t mine) const {
        std::vector<char> all;
        comm_->gatherAll(&mine, sizeof(mine), all);
        std::vector<int64_t> out((size_t)comm_->getSize());
        std::memcpy(out.data(), all.data(), out.size() * sizeof(int64_t));
        return out;
    }
    // rank 0 creates the file at its final size; nobody writes before it exists
    void createFile(const std::string& f, size_t bytes) const {
        if (comm_->getRank() == 0) {
            std::ofstream os(f, std::ios::binary | std::ios::trunc);
            TEUCHOS_TEST_FOR_EXCEPTION(!os, std::runtime_error, "ExporterParaView: cannot write " << f);
            if (bytes > 0) {
                os.seekp((std::streamoff)bytes - 1);
                os.put('\0');
            }
        }
        comm_->barrier();
    }
    static void writeAt(const std::string& f, size_t offset, const void* p, size_t bytes) {
        if (bytes == 0) return;
        std::fstream os(f, std::ios::binary | std::ios::in | std::ios::out);
        TEUCHOS_TEST_FOR_EXCEPTION(!os, std::runtime_error, "ExporterParaView: cannot write " << f);
        os.seekp((std::streamoff)offset);
        os.write((const char*)p, (std::streamsize)bytes);
        TEUCHOS_TEST_FOR_EXCEPTION(!os, std::runtime_error, "ExporterParaView: short write to " << f);
    }
    // record i of `data` (one per unique node) to position gid(i): runs of consecutive global ids go out in one write
    void writeRecords(const std::string& f, const double* data, size_t recBytes) const {
        if (uniGid_.empty()) return;
        std::fstream os(f, std::ios::binary | std::ios::in | std::ios::out);
        TEUCHOS_TEST_FOR_EXCEPTION(!os, std::runtime_error, "ExporterParaView: cannot write " << f);
        size_t i = 0;
        const size_t n = uniGid_.size();
        while (i < n) {
            size_t j = i + 1;
            while (j < n && uniGid_[j] == uniGid_[j - 1] + 1) ++j;
            os.seekp((std::streamoff)((size_t)uniGid_[i] * recBytes));
            os.write((const char*)data + i * recBytes, (std::streamsize)((j - i) * recBytes));
            i = j;
        }
        TEUCHOS_TEST_FOR_EXCEPTION(!os, std::runtime_error, "ExporterParaView: short write to " << f);
    }
    static std::string base(const std::string& f) { const size_t p = f.find_last_of('/'); return p == std::string::npos ? f : f.substr(p + 1); }
    void writeXmf() const {
        std::ofstream os(filename_ + ".xmf");
        TEUCHOS_TEST_FOR_EXCEPTION(!os, std::runtime_error, "ExporterParaView: cannot write " << filename_ << ".xmf");
        const std::string b = base(filename_);
        os << "<?xml version=\"1.0\" ?>\n<Xdmf Version=\"2.0\">\n<Domain>\n<Grid Name=\"" << b
           << "\" GridType=\"Collection\" CollectionType=\"Temporal\">\n";
        for (size_t k = 0; k < times_.size(); ++k) {
            os << " <Grid Name=\"step" << k << "\" GridType=\"Uniform\">\n  <Time Value=\"" << times_[k] << "\"/>\n"
               << "  <Topology TopologyType=\"" << topology_ << "\" NumberOfElements=\"" << nElements_ << "\">\n"
               << "   <DataItem Format=\"Binary\" DataType=\"Int\" Precision=\"4\" Endian=\"Little\" Dimensions=\"" << nElements_ << " " << nv_
               << "\">" << b << ".conn.bin</DataItem>\n  </Topology>\n"
               << "  <Geometry GeometryType=\"XYZ\">\n   <DataItem Format=\"Binary\" DataType=\"Float\" Precision=\"8\" Endian=\"Little\" Dimensions=\""
               << nPoints_ << " 3\">" << b << ".xyz.bin</DataItem>\n  </Geometry>\n";
            for (auto& v : vars_) {
                const bool vec = v.type == "Vector";
                os << "  <Attribute Name=\"" << v.name << "\" AttributeType=\"" << (vec ? "Vector" : "Scalar") << "\" Center=\"Node\">\n"
                   << "   <DataItem Format=\"Binary\" DataType=\"Float\" Precision=\"8\" Endian=\"Little\" Dimensions=\"" << nPoints_
                   << (vec ? " 3" : "") << "\">" << b << "." << v.name << "." << k << ".bin</DataItem>\n  </Attribute>\n";
            }
            os << " </Grid>\n";
        }
        os << "</Grid>\n</Domain>\n</Xdmf>\n";
    }
    std::string filename_, FEType_, topology_;
    MeshPtr_Type mesh_;
    Teuchos::RCP<const Teuchos::Comm<int> > comm_;
    std::vector<int64_t> uniGid_;
    std::vector<Var> vars_;
    std::vector<double> times_;
    size_t nPoints_ = 0, nElements_ = 0;
    int nv_ = 0, saveTimestep_ = 1, timeIndex_ = 0, nSaved_ = 0;
};

// device time of the kernel classes of the hot path (HIP events on the library's stream, fedd_timing_*) as children of
// the running stacked timer: the StackedTimer report of the drivers' tails then shows where the GPU time went
inline void addDeviceTimers(const DeviceContextPtr& dev, Teuchos::StackedTimer& st) {
    static const char* names[FEDD_T_COUNT] = {"symbolic", "assemble", "rhs", "dirichlet", "spmv", "schwarz setup", "schwarz apply",
                                             "orthogonalisation", "coarse setup", "coarse apply", "halo", "all-reduce", "spmv setup",
                                             "orthogonalisation: dot sweep", "orthogonalisation: update sweep"};
    for (int t = 0; t < FEDD_T_COUNT; ++t) {
        double ms = 0.;
        int64_t n = 0;
        if (fedd_timing_get(dev->ctx, t, &ms, &n) == 0 && n > 0) st.addExternal(std::string("FEDD - device - ") + names[t], 1e-3 * ms, (long)n);
    }
}

}  // namespace FEDD
