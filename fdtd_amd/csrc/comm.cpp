// RCCL binding for the halo exchange (see comm.h).  Only the six entry points
// the exchange needs are resolved.
#include "comm.h"

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "internal.h"

namespace fdtd {

namespace {

// ABI of rccl.h (NCCL 2.x): ncclUniqueId is 128 opaque bytes passed by value,
// ncclFloat32 = 7, results are ints with 0 = success.
struct UniqueId {
    char internal[128];
};
typedef void *CommHandle;
constexpr int kFloat32 = 7;

struct Api {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(CommHandle *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(CommHandle) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, CommHandle, void *) = nullptr;
    int (*Recv)(void *, size_t, int, int, CommHandle, void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

Api g_api;
std::once_flag g_once;
std::string g_load_error;

void load_api()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_api.lib) break;
    }
    if (!g_api.lib) {
        g_load_error = std::string("cannot load librccl: ") + dlerror();
        return;
    }
    auto sym = [&](const char *name) {
        void *p = dlsym(g_api.lib, name);
        if (!p && g_load_error.empty()) g_load_error = std::string("librccl lacks ") + name;
        return p;
    };
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))sym("ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))sym("ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))sym("ncclCommDestroy");
    g_api.GroupStart = (decltype(g_api.GroupStart))sym("ncclGroupStart");
    g_api.GroupEnd = (decltype(g_api.GroupEnd))sym("ncclGroupEnd");
    g_api.Send = (decltype(g_api.Send))sym("ncclSend");
    g_api.Recv = (decltype(g_api.Recv))sym("ncclRecv");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))sym("ncclGetErrorString");
}

int ensure_api()
{
    std::call_once(g_once, load_api);
    if (!g_load_error.empty()) {
        set_error(g_load_error);
        return -20;
    }
    return 0;
}

int check(int rc, const char *what)
{
    if (rc == 0) return 0;
    set_error(std::string(what) + ": " + (g_api.GetErrorString ? g_api.GetErrorString(rc) : "RCCL error"));
    return -21;
}

} // namespace

// ---- in-process transport ------------------------------------------------

// Ranks meet on the host: a sender waits until its data is final (it synchronises the stream the
// exchange is queued on, not the context's other streams, so a sweep running beside the exchange
// keeps running), posts the buffer, the receiver copies it and acknowledges.  No event of one
// rank's stream is ever waited for by another rank's stream.
struct LocalMsg {
    const float *buf;
    size_t count;
};

struct LocalGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::deque<LocalMsg>> box; // [src * world + dst]: sends posted, not yet received
    std::vector<int> ack;                  // [src * world + dst]: messages of src that dst has copied out
    // Set by the first exchange that fails on any rank (time limit, size mismatch, HIP error): posted
    // messages and acknowledgements are then out of step, so every later exchange of the group fails
    // too instead of pairing a stale message with a new receive.
    bool failed = false;
};

struct LocalOp {
    bool send;
    float *buf;
    size_t count;
    int peer;
    hipStream_t stream;
};

struct Comm {
    CommHandle handle = nullptr; // RCCL
    LocalGroup *local = nullptr; // or the in-process group
    int rank = 0;
    bool grouped = false;
    std::vector<LocalOp> ops;
};

namespace {

// A peer that never arrives is an error, not a hang (seconds; FDTD_LOCAL_TIMEOUT overrides).
int local_timeout()
{
    static const int v = std::getenv("FDTD_LOCAL_TIMEOUT") ? std::max(1, std::atoi(std::getenv("FDTD_LOCAL_TIMEOUT"))) : 120;
    return v;
}

int hipcheck(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return -10;
}

int local_run_ops(Comm *c, const std::vector<LocalOp> &ops);

int local_run(Comm *c)
{
    LocalGroup *g = c->local;
    std::vector<LocalOp> ops;
    ops.swap(c->ops);
    {
        std::lock_guard<std::mutex> lk(g->m);
        if (g->failed) {
            set_error("in-process halo exchange: an earlier exchange of this group failed; the group is unusable");
            return -23;
        }
    }
    const int rc = local_run_ops(c, ops);
    if (rc) {
        std::lock_guard<std::mutex> lk(g->m);
        g->failed = true;
        g->cv.notify_all();
    }
    return rc;
}

int local_run_ops(Comm *c, const std::vector<LocalOp> &ops)
{
    LocalGroup *g = c->local;
    const int w = g->world, r = c->rank;
    // 1. the data to send is final once the stream has drained; post every send (never blocks on a peer)
    hipStream_t drained = nullptr;
    bool any = false;
    for (const LocalOp &op : ops) {
        if (!op.send) continue;
        if (!any || op.stream != drained) {
            if (int rc = hipcheck(hipStreamSynchronize(op.stream), "hipStreamSynchronize")) return rc;
            drained = op.stream;
            any = true;
        }
        std::lock_guard<std::mutex> lk(g->m);
        g->box[(size_t)r * w + op.peer].push_back({op.buf, op.count});
        g->cv.notify_all();
    }
    // 2. receive: wait until the peer has posted, copy, acknowledge
    for (const LocalOp &op : ops) {
        if (op.send) continue;
        LocalMsg msg;
        {
            std::unique_lock<std::mutex> lk(g->m);
            auto &q = g->box[(size_t)op.peer * w + r];
            if (!g->cv.wait_for(lk, std::chrono::seconds(local_timeout()), [&] { return !q.empty() || g->failed; }) ||
                g->failed) {
                set_error(g->failed ? "in-process halo exchange: the exchange failed on another rank"
                                    : "in-process halo exchange: the neighbour rank did not send within the time limit");
                return -23;
            }
            msg = q.front();
            q.pop_front();
        }
        if (msg.count != op.count) {
            set_error("in-process halo exchange: neighbour ranks disagree on the message size "
                      "(different kernel or steps per pass?)");
            return -23;
        }
        int rc = hipcheck(hipMemcpyAsync(op.buf, msg.buf, op.count * sizeof(float), hipMemcpyDefault, op.stream),
                          "hipMemcpyAsync");
        if (!rc) rc = hipcheck(hipStreamSynchronize(op.stream), "hipStreamSynchronize");
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(g->m);
        g->ack[(size_t)op.peer * w + r]++;
        g->cv.notify_all();
    }
    // 3. a send buffer may be reused once the peer has copied it out
    for (const LocalOp &op : ops) {
        if (!op.send) continue;
        std::unique_lock<std::mutex> lk(g->m);
        int &n = g->ack[(size_t)r * w + op.peer];
        if (!g->cv.wait_for(lk, std::chrono::seconds(local_timeout()), [&] { return n > 0 || g->failed; }) || n <= 0) {
            set_error(g->failed ? "in-process halo exchange: the exchange failed on another rank"
                                : "in-process halo exchange: the neighbour rank did not receive within the time limit");
            return -23;
        }
        n--;
    }
    return 0;
}

} // namespace

LocalGroup *local_group_create(int world)
{
    if (world < 1) return nullptr;
    LocalGroup *g = new LocalGroup();
    g->world = world;
    g->box.resize((size_t)world * world);
    g->ack.assign((size_t)world * world, 0);
    return g;
}

void local_group_destroy(LocalGroup *g) { delete g; }

int comm_create_local(LocalGroup *g, int rank, Comm **out)
{
    if (!g || rank < 0 || rank >= g->world) {
        set_error("in-process communicator: bad group or rank");
        return -3;
    }
    Comm *c = new Comm();
    c->local = g;
    c->rank = rank;
    *out = c;
    return 0;
}

int comm_unique_id(char id[128])
{
    if (int rc = ensure_api()) return rc;
    UniqueId u;
    if (int rc = check(g_api.GetUniqueId(&u), "ncclGetUniqueId")) return rc;
    std::memcpy(id, u.internal, 128);
    return 0;
}

int comm_create(const char id[128], int rank, int world, Comm **out)
{
    if (int rc = ensure_api()) return rc;
    UniqueId u;
    std::memcpy(u.internal, id, 128);
    Comm *c = new Comm();
    c->rank = rank;
    if (int rc = check(g_api.CommInitRank(&c->handle, world, u, rank), "ncclCommInitRank")) {
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}

void comm_destroy(Comm *c)
{
    if (!c) return;
    if (c->handle) g_api.CommDestroy(c->handle);
    delete c;
}

int comm_group_start(Comm *c)
{
    if (c->local) {
        c->grouped = true;
        return 0;
    }
    return check(g_api.GroupStart(), "ncclGroupStart");
}

int comm_group_end(Comm *c)
{
    if (c->local) {
        c->grouped = false;
        return local_run(c);
    }
    return check(g_api.GroupEnd(), "ncclGroupEnd");
}

int comm_send(Comm *c, const float *buf, size_t count, int peer, void *stream)
{
    if (c->local) {
        c->ops.push_back({true, const_cast<float *>(buf), count, peer, (hipStream_t)stream});
        return c->grouped ? 0 : local_run(c);
    }
    return check(g_api.Send(buf, count, kFloat32, peer, c->handle, stream), "ncclSend");
}

int comm_recv(Comm *c, float *buf, size_t count, int peer, void *stream)
{
    if (c->local) {
        c->ops.push_back({false, buf, count, peer, (hipStream_t)stream});
        return c->grouped ? 0 : local_run(c);
    }
    return check(g_api.Recv(buf, count, kFloat32, peer, c->handle, stream), "ncclRecv");
}

} // namespace fdtd
