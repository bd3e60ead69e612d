// Minimal stand-ins for the Teuchos types that appear in FEDDLib's public interface, so that the
// host facade (namespace FEDD) compiles without Trilinos.  With a real Trilinos installed, define
// FEDD_HAVE_TRILINOS and include the real headers instead: the facade only uses the subset below.
// API subset mirrored: Teuchos::RCP / rcp / null / rcp_const_cast, Teuchos::ParameterList (get,
// sublist, set, setParameters, isParameter, isSublist), Teuchos::getParametersFromXmlFile,
// Teuchos::Comm<int> (getRank/getSize/barrier; several ranks: see CommBackend), Teuchos::GlobalMPISession,
// Teuchos::DefaultComm<int>::getComm, Teuchos::ArrayRCP / ArrayView (pointer + size),
// TEUCHOS_TEST_FOR_EXCEPTION, Teuchos::Time / TimeMonitor / StackedTimer (report with OutputOptions).
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <deque>
#include <exception>
#include <iostream>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#define TEUCHOS_TEST_FOR_EXCEPTION(cond, exc, msg)                      \
    do {                                                                \
        if (cond) {                                                     \
            std::ostringstream os__;                                    \
            os__ << __FILE__ << ":" << __LINE__ << ": " << msg;         \
            throw exc(os__.str());                                      \
        }                                                               \
    } while (0)

namespace Teuchos {

struct ENull { };
static const ENull null = ENull();

template <class T>
class RCP {
public:
    RCP() {}
    RCP(ENull) {}
    explicit RCP(T* p) : p_(p) {}
    RCP(const std::shared_ptr<T>& p) : p_(p) {}
    template <class U>
    RCP(const RCP<U>& o) : p_(o.shared()) {}
    T* operator->() const { return p_.get(); }
    T& operator*() const { return *p_; }
    T* get() const { return p_.get(); }
    bool is_null() const { return !p_; }
    void reset(T* p = nullptr) { p_.reset(p); }
    const std::shared_ptr<T>& shared() const { return p_; }
    bool operator==(ENull) const { return !p_; }
    bool operator!=(ENull) const { return (bool)p_; }
private:
    std::shared_ptr<T> p_;
};

template <class T>
RCP<T> rcp(T* p) { return RCP<T>(p); }
template <class T, class U>
RCP<T> rcp_const_cast(const RCP<U>& p) { return RCP<T>(std::const_pointer_cast<T>(p.shared())); }
template <class T, class U>
RCP<T> rcp_dynamic_cast(const RCP<U>& p) { return RCP<T>(std::dynamic_pointer_cast<T>(p.shared())); }

template <class T>
class ArrayView {
public:
    ArrayView() {}
    ArrayView(T* p, size_t n) : p_(p), n_(n) {}
    T& operator[](size_t i) const { return p_[i]; }
    size_t size() const { return n_; }
    T* getRawPtr() const { return p_; }
private:
    T* p_ = nullptr;
    size_t n_ = 0;
};
template <class T>
using ArrayRCP = ArrayView<T>;

// ---- several ranks -------------------------------------------------------------------------------------------------
// The facade needs little of a communicator itself (a barrier, an all-gather of a few bytes: the 128-byte RCCL id, the
// halo request lists, norms): the data path of the solver is inside the library (RCCL, or its host-callback transport).
// Two backends:
//  * ThreadGroup: the ranks are threads of ONE process (Teuchos::runAsRanks); also carries the library's host-callback
//    transport (point-to-point mailboxes).  For functional runs of the N > 1 path on a box with fewer GPUs than ranks.
//  * FileRendezvous: the ranks are processes started by any launcher that exports RANK / WORLD_SIZE (torchrun), FEDD_RANK /
//    FEDD_NRANKS or OMPI_COMM_WORLD_RANK / _SIZE; the few bytes travel through files in the directory FEDD_RENDEZVOUS
//    (shared by the ranks, empty at start).  The solver's traffic goes over RCCL.
// With a real Teuchos (MPI) the facade's Comm calls map onto Teuchos::gatherAll / barrier one to one.
class CommBackend {
public:
    virtual ~CommBackend() {}
    virtual bool inProcess() const = 0;
    // out = the `bytes` of every rank, in rank order (every rank passes the same `bytes`)
    virtual void allgather(int rank, const void* in, size_t bytes, std::vector<char>& out) = 0;
    virtual void send(int src, int dst, const double* p, size_t n) = 0;
    virtual void recv(int src, int dst, double* p, size_t n) = 0;
};

class ThreadGroup : public CommBackend {
public:
    explicit ThreadGroup(int world) : world_(world), slots_(world) {}
    bool inProcess() const override { return true; }
    void allgather(int rank, const void* in, size_t bytes, std::vector<char>& out) override {
        std::unique_lock<std::mutex> lk(m_);
        slots_[rank].assign((const char*)in, (const char*)in + bytes);
        arrive(lk);                              // everyone has written
        out.resize(bytes * world_);
        for (int r = 0; r < world_; ++r) std::copy(slots_[r].begin(), slots_[r].end(), out.begin() + r * bytes);
        arrive(lk);                              // everyone has read: the slots may be rewritten
    }
    void send(int src, int dst, const double* p, size_t n) override {
        std::lock_guard<std::mutex> lk(m_);
        box_[{src, dst}].emplace_back(p, p + n);
        cv_.notify_all();
    }
    void recv(int src, int dst, double* p, size_t n) override {
        std::unique_lock<std::mutex> lk(m_);
        auto& q = box_[{src, dst}];
        const bool ok = cv_.wait_for(lk, std::chrono::seconds(120), [&] { return !q.empty(); });
        TEUCHOS_TEST_FOR_EXCEPTION(!ok, std::runtime_error, "ThreadGroup::recv " << dst << " <- " << src << " timed out");
        TEUCHOS_TEST_FOR_EXCEPTION(q.front().size() != n, std::runtime_error, "ThreadGroup::recv: message size");
        std::copy(q.front().begin(), q.front().end(), p);
        q.pop_front();
    }
private:
    void arrive(std::unique_lock<std::mutex>& lk) {      // reusable barrier (generation counter)
        TEUCHOS_TEST_FOR_EXCEPTION(broken_, std::runtime_error, "ThreadGroup: a rank did not reach an earlier collective");
        const long gen = gen_;
        if (++count_ == world_) {
            count_ = 0;
            ++gen_;
            cv_.notify_all();
        } else {
            const bool ok = cv_.wait_for(lk, std::chrono::seconds(120), [&] { return gen_ != gen || broken_; });
            if (!ok || broken_) {
                // a rank never arrived: the barrier is poisoned for everyone (the count of this generation can no longer be
                // trusted), so the remaining ranks fail at their next collective instead of pairing up with a stale count
                broken_ = true;
                cv_.notify_all();
            }
            TEUCHOS_TEST_FOR_EXCEPTION(broken_, std::runtime_error, "ThreadGroup: a rank did not reach the collective");
        }
    }
    int world_, count_ = 0;
    long gen_ = 0;
    bool broken_ = false;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<std::vector<char>> slots_;
    std::map<std::pair<int, int>, std::deque<std::vector<double>>> box_;
};

class FileRendezvous : public CommBackend {
public:
    FileRendezvous(const std::string& dir, int world) : dir_(dir), world_(world) {}
    bool inProcess() const override { return false; }
    void allgather(int rank, const void* in, size_t bytes, std::vector<char>& out) override {
        const long seq = seq_++;
        const std::string mine = name(seq, rank);
        {
            std::ofstream f(mine + ".tmp", std::ios::binary);
            f.write((const char*)in, (std::streamsize)bytes);
            TEUCHOS_TEST_FOR_EXCEPTION(!f, std::runtime_error, "FileRendezvous: cannot write in " << dir_);
        }
        TEUCHOS_TEST_FOR_EXCEPTION(std::rename((mine + ".tmp").c_str(), mine.c_str()) != 0, std::runtime_error,
                                   "FileRendezvous: rename in " << dir_);
        out.resize(bytes * world_);
        for (int r = 0; r < world_; ++r) {
            const std::string fn = name(seq, r);
            bool ok = false;
            for (int tries = 0; tries < 12000 && !ok; ++tries) {      // 120 s
                std::ifstream f(fn, std::ios::binary);
                if (f && f.read(out.data() + r * bytes, (std::streamsize)bytes)) ok = true;
                else std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            TEUCHOS_TEST_FOR_EXCEPTION(!ok, std::runtime_error, "FileRendezvous: rank " << r << " did not arrive (" << fn << ")");
        }
        // everyone who wrote file seq has finished reading the files of seq - 1
        if (seq > 0) std::remove(name(seq - 1, rank).c_str());
    }
    void send(int, int, const double*, size_t) override { TEUCHOS_TEST_FOR_EXCEPTION(true, std::logic_error, "FileRendezvous: no point-to-point (RCCL carries the data)"); }
    void recv(int, int, double*, size_t) override { TEUCHOS_TEST_FOR_EXCEPTION(true, std::logic_error, "FileRendezvous: no point-to-point (RCCL carries the data)"); }
private:
    std::string name(long seq, int rank) const { return dir_ + "/fedd_" + std::to_string(seq) + "_" + std::to_string(rank) + ".bin"; }
    std::string dir_;
    int world_;
    long seq_ = 0;
};

template <class Ordinal>
class Comm {
public:
    Comm(int rank = 0, int size = 1, std::shared_ptr<CommBackend> backend = nullptr) : rank_(rank), size_(size), backend_(backend) {
        TEUCHOS_TEST_FOR_EXCEPTION(size > 1 && !backend, std::logic_error, "Teuchos::Comm: " << size << " ranks need a backend (Teuchos::DefaultComm / runAsRanks)");
    }
    int getRank() const { return rank_; }
    int getSize() const { return size_; }
    void barrier() const {
        if (size_ > 1) {
            char c = 0;
            std::vector<char> all;
            backend_->allgather(rank_, &c, 1, all);
        }
    }
    // the facade's collectives (Teuchos::gatherAll / reduceAll(REDUCE_SUM) / broadcast); sums in rank order: same bits everywhere
    void gatherAll(const void* in, size_t bytes, std::vector<char>& out) const {
        if (size_ > 1) backend_->allgather(rank_, in, bytes, out);
        else out.assign((const char*)in, (const char*)in + bytes);
    }
    void sumAll(double* v, int n) const {
        if (size_ == 1) return;
        std::vector<char> all;
        backend_->allgather(rank_, v, n * sizeof(double), all);
        const double* a = (const double*)all.data();
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int r = 0; r < size_; ++r) s += a[(size_t)r * n + i];
            v[i] = s;
        }
    }
    void broadcast(int root, size_t bytes, void* buf) const {
        if (size_ == 1) return;
        std::vector<char> all;
        backend_->allgather(rank_, buf, bytes, all);
        std::copy(all.begin() + root * bytes, all.begin() + (root + 1) * bytes, (char*)buf);
    }
    CommBackend* backend() const { return backend_.get(); }
private:
    int rank_, size_;
    std::shared_ptr<CommBackend> backend_;
};

// Teuchos::GlobalMPISession / DefaultComm: the communicator of the calling rank -- the one set by runAsRanks for this
// thread, else the one the launcher's environment describes (one rank when it describes none)
class GlobalMPISession {
public:
    GlobalMPISession(int*, char***) {}
};
namespace detail {
inline RCP<const Comm<int>>& threadComm() {
    static thread_local RCP<const Comm<int>> c;
    return c;
}
inline int envInt(const char* const* names, int dflt) {
    for (; *names; ++names)
        if (const char* v = std::getenv(*names)) return std::atoi(v);
    return dflt;
}
}  // namespace detail
template <class Ordinal>
class DefaultComm {
public:
    static RCP<const Comm<Ordinal>> getComm() {
        if (!detail::threadComm().is_null()) return detail::threadComm();
        static RCP<const Comm<Ordinal>> process;
        if (process.is_null()) {
            static const char* const rk[] = {"FEDD_RANK", "RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", nullptr};
            static const char* const sz[] = {"FEDD_NRANKS", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", nullptr};
            const int rank = detail::envInt(rk, 0), size = detail::envInt(sz, 1);
            std::shared_ptr<CommBackend> be;
            if (size > 1) {
                const char* dir = std::getenv("FEDD_RENDEZVOUS");
                TEUCHOS_TEST_FOR_EXCEPTION(!dir, std::runtime_error, "Teuchos::DefaultComm: " << size << " ranks: set FEDD_RENDEZVOUS to an empty directory shared by the ranks");
                be = std::make_shared<FileRendezvous>(dir, size);
            }
            process = rcp(new Comm<Ordinal>(rank, size, be));
        }
        return process;
    }
};

// run `body(rank)` as `world` ranks = threads of this process; inside, DefaultComm<int>::getComm() is the rank's
// communicator.  Returns the largest return value; an exception of any rank is rethrown after all have ended.
template <class F>
int runAsRanks(int world, F body) {
    auto group = std::make_shared<ThreadGroup>(world);
    std::vector<int> rc(world, 0);
    std::vector<std::exception_ptr> err(world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            detail::threadComm() = rcp(new Comm<int>(r, world, group));
            try {
                rc[r] = body(r);
            } catch (...) {
                err[r] = std::current_exception();
            }
            detail::threadComm() = RCP<const Comm<int>>();
        });
    for (auto& t : th) t.join();
    for (auto& e : err)
        if (e) std::rethrow_exception(e);
    int worst = 0;
    for (int v : rc) worst = std::max(worst, v);
    return worst;
}

class ParameterList {
public:
    ParameterList(const std::string& name = "") : name_(name) {}
    const std::string& name() const { return name_; }
    ParameterList& sublist(const std::string& n) {
        auto it = sub_.find(n);
        if (it == sub_.end()) it = sub_.emplace(n, ParameterList(n)).first;
        return it->second;
    }
    bool isSublist(const std::string& n) const { return sub_.count(n) > 0; }
    bool isParameter(const std::string& n) const { return val_.count(n) > 0; }
    template <class T>
    ParameterList& set(const std::string& n, const T& v) {
        std::ostringstream os;
        os.precision(17);
        os << v;
        val_[n] = os.str();
        return *this;
    }
    ParameterList& set(const std::string& n, const char* v) { val_[n] = v; return *this; }
    ParameterList& set(const std::string& n, bool v) { val_[n] = v ? "true" : "false"; return *this; }
    // get with default: like Teuchos, the default is stored when the entry is missing
    int get(const std::string& n, int d) { return has(n) ? std::atoi(val_[n].c_str()) : (set(n, d), d); }
    double get(const std::string& n, double d) { return has(n) ? std::atof(val_[n].c_str()) : (set(n, d), d); }
    bool get(const std::string& n, bool d) {
        if (!has(n)) { set(n, d); return d; }
        const std::string& s = val_[n];
        return s == "true" || s == "1" || s == "True";
    }
    std::string get(const std::string& n, const std::string& d) { return has(n) ? val_[n] : (val_[n] = d, d); }
    std::string get(const std::string& n, const char* d) { return get(n, std::string(d)); }
    // merge, entries of `o` win (Teuchos::ParameterList::setParameters)
    ParameterList& setParameters(const ParameterList& o) {
        for (auto& kv : o.val_) val_[kv.first] = kv.second;
        for (auto& kv : o.sub_) sublist(kv.first).setParameters(kv.second);
        return *this;
    }
    void print(std::ostream& os, int indent = 0) const {
        for (auto& kv : val_) os << std::string(indent, ' ') << kv.first << " = " << kv.second << "\n";
        for (auto& kv : sub_) {
            os << std::string(indent, ' ') << kv.first << " ->\n";
            kv.second.print(os, indent + 2);
        }
    }
private:
    bool has(const std::string& n) const { return val_.count(n) > 0; }
    std::string name_;
    std::map<std::string, std::string> val_;
    std::map<std::string, ParameterList> sub_;
};

inline RCP<ParameterList> sublist(const RCP<ParameterList>& pl, const std::string& name) {
    // aliasing pointer into the parent (keeps the parent alive)
    return RCP<ParameterList>(std::shared_ptr<ParameterList>(pl.shared(), &pl->sublist(name)));
}

namespace detail {
inline std::string attr(const std::string& tag, const std::string& key) {
    const std::string k = key + "=\"";
    size_t p = tag.find(k);
    if (p == std::string::npos) return "";
    p += k.size();
    size_t e = tag.find('"', p);
    return tag.substr(p, e - p);
}
}  // namespace detail

// Reader for the Teuchos XML parameter-list dialect FEDDLib's drivers use
// (<ParameterList name=..> <Parameter name=.. type=.. value=../> </ParameterList>, <!-- comments -->).
inline RCP<ParameterList> getParametersFromXmlFile(const std::string& file) {
    std::ifstream in(file);
    TEUCHOS_TEST_FOR_EXCEPTION(!in, std::runtime_error, "cannot open parameter file " << file);
    std::stringstream ss;
    ss << in.rdbuf();
    std::string s = ss.str();
    for (size_t p; (p = s.find("<!--")) != std::string::npos;) {
        size_t e = s.find("-->", p);
        s.erase(p, e == std::string::npos ? std::string::npos : e + 3 - p);
    }
    RCP<ParameterList> root;
    std::vector<ParameterList*> stack;
    size_t pos = 0;
    while ((pos = s.find('<', pos)) != std::string::npos) {
        size_t e = s.find('>', pos);
        TEUCHOS_TEST_FOR_EXCEPTION(e == std::string::npos, std::runtime_error, "malformed XML in " << file);
        const std::string tag = s.substr(pos + 1, e - pos - 1);
        pos = e + 1;
        if (tag.compare(0, 14, "/ParameterList") == 0) {
            if (!stack.empty()) stack.pop_back();
        } else if (tag.compare(0, 13, "ParameterList") == 0) {
            const std::string name = detail::attr(tag, "name");
            if (stack.empty()) {
                root = rcp(new ParameterList(name));
                stack.push_back(root.get());
            } else {
                stack.push_back(&stack.back()->sublist(name));
            }
            if (!tag.empty() && tag.back() == '/') stack.pop_back();
        } else if (tag.compare(0, 9, "Parameter") == 0) {
            TEUCHOS_TEST_FOR_EXCEPTION(stack.empty(), std::runtime_error, "Parameter outside a ParameterList in " << file);
            stack.back()->set(detail::attr(tag, "name"), detail::attr(tag, "value").c_str());
        }
    }
    TEUCHOS_TEST_FOR_EXCEPTION(root.is_null(), std::runtime_error, "no ParameterList in " << file);
    return root;
}

// ---- timers: the subset of Teuchos::Time / TimeMonitor / StackedTimer the drivers' tails use
// (steadyLinElas_Perf/main.cpp:114-115, 245-249; FEDD_TIMER_START / FEDD_TIMER_STOP of the reference wrap TimeMonitor) ----
typedef std::ostream FancyOStream;

class StackedTimer {
public:
    struct OutputOptions {
        bool output_fraction = false, output_total_updates = false, output_histogram = false, output_minmax = false,
             print_warnings = true, align_columns = true, print_names_before_values = true;
        int max_levels = 100;
    };
    explicit StackedTimer(const std::string& name) {
        root_.name = name;
        stack_.push_back(&root_);
        begin(root_);
    }
    void start(const std::string& name) {
        Node* parent = stack_.back();
        Node* n = nullptr;
        for (auto& ch : parent->children)
            if (ch->name == name) n = ch.get();
        if (!n) {
            parent->children.emplace_back(new Node());
            n = parent->children.back().get();
            n->name = name;
        }
        stack_.push_back(n);
        begin(*n);
    }
    void stop(const std::string& name) {
        TEUCHOS_TEST_FOR_EXCEPTION(stack_.empty() || stack_.back()->name != name, std::runtime_error,
                                   "StackedTimer::stop(\"" << name << "\"): the running timer is \""
                                                           << (stack_.empty() ? std::string("<none>") : stack_.back()->name) << "\"");
        end(*stack_.back());
        stack_.pop_back();
    }
    // device-side or otherwise externally measured time as a child of the running timer
    void addExternal(const std::string& name, double seconds, long count) {
        Node* parent = stack_.empty() ? &root_ : stack_.back();
        parent->children.emplace_back(new Node());
        Node* n = parent->children.back().get();
        n->name = name;
        n->total = seconds;
        n->count = count;
    }
    double accumulatedTime(const std::string& name) const {
        const Node* n = find(&root_, name);
        return n ? n->total : 0.0;
    }
    template <class CommPtr>
    void report(std::ostream& os, const CommPtr&, const OutputOptions& options) const { report(os, options); }
    template <class CommPtr>
    void report(std::ostream& os, const CommPtr&) const { report(os); }
    void report(std::ostream& os, const OutputOptions& options) const {
        print(os, root_, 0, root_.total > 0 ? root_.total : 1.0, options);
    }
    void report(std::ostream& os) const {
        OutputOptions options;
        report(os, options);
    }
private:
    struct Node {
        std::string name;
        double total = 0.0, t0 = 0.0;
        long count = 0;
        bool running = false;
        std::vector<std::unique_ptr<Node>> children;
    };
    static double now() {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    static void begin(Node& n) { n.t0 = now(); n.running = true; }
    static void end(Node& n) { n.total += now() - n.t0; n.count += 1; n.running = false; }
    static const Node* find(const Node* n, const std::string& name) {
        if (n->name == name) return n;
        for (auto& ch : n->children)
            if (const Node* f = find(ch.get(), name)) return f;
        return nullptr;
    }
    static void print(std::ostream& os, const Node& n, int level, double parent_total, const OutputOptions& o) {
        if (level > o.max_levels) return;
        const double t = n.running ? n.total + (now() - n.t0) : n.total;
        os << std::string(2 * level, ' ') << n.name << ": " << t << " [" << n.count << "]";
        if (o.output_fraction && level > 0) os << " (" << (parent_total > 0 ? t / parent_total : 0.0) << ")";
        os << "\n";
        double sum = 0.0;
        for (auto& ch : n.children) {
            print(os, *ch, level + 1, t, o);
            sum += ch->total;
        }
        if (!n.children.empty() && level < o.max_levels)
            os << std::string(2 * (level + 1), ' ') << "Remainder: " << (t - sum) << "\n";
    }
    Node root_;
    std::vector<Node*> stack_;
};

class Time {
public:
    explicit Time(const std::string& name, bool start_now = false) : name_(name) { if (start_now) start(); }
    void start() { t0_ = wall(); running_ = true; }
    double stop() { if (running_) { total_ += wall() - t0_; ++calls_; running_ = false; } return total_; }
    double totalElapsedTime() const { return total_; }
    int numCalls() const { return calls_; }
    const std::string& name() const { return name_; }
private:
    static double wall() {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    std::string name_;
    double total_ = 0.0, t0_ = 0.0;
    int calls_ = 0;
    bool running_ = false;
};

// RAII monitor; timers are registered by name, mirrored into the stacked timer when one is set
class TimeMonitor {
public:
    explicit TimeMonitor(Time& t) : t_(&t) {
        t_->start();
        if (!stacked().is_null()) stacked()->start(t_->name());
    }
    ~TimeMonitor() noexcept {
        // a destructor may run while another exception unwinds the FEDD_TIMER scopes of a rank thread: a throw from here
        // (StackedTimer::stop checks that the names nest) would be std::terminate, so a mismatch is reported, not thrown
        try {
            t_->stop();
            if (!stacked().is_null()) stacked()->stop(t_->name());
        } catch (const std::exception& e) {
            std::cerr << "TimeMonitor: " << e.what() << std::endl;
        } catch (...) {
        }
    }
    static RCP<Time> getNewCounter(const std::string& name) {
        auto& reg = registry();
        auto it = reg.find(name);
        if (it == reg.end()) it = reg.emplace(name, rcp(new Time(name))).first;
        return it->second;
    }
    static void setStackedTimer(const RCP<StackedTimer>& st) { stacked() = st; }
    static RCP<StackedTimer> getStackedTimer() { return stacked(); }
    static void summarize(std::ostream& os = std::cout) {
        os << "TimeMonitor results\n";
        for (auto& kv : registry())
            os << "  " << kv.first << ": " << kv.second->totalElapsedTime() << " s (" << kv.second->numCalls() << ")\n";
    }
    static void report(std::ostream& os) { summarize(os); }
private:
    // per thread: with the ranks as threads of one process (runAsRanks) every rank has its own timers, like every MPI process
    static std::map<std::string, RCP<Time>>& registry() { static thread_local std::map<std::string, RCP<Time>> r; return r; }
    static RCP<StackedTimer>& stacked() { static thread_local RCP<StackedTimer> s; return s; }
    Time* t_;
};

}  // namespace Teuchos
