// RCCL binding for the halo exchange (see comm.h).  Only the six entry points
// the exchange needs are resolved.
#include "comm.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

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

struct Comm {
    CommHandle handle = nullptr;
};

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

int comm_group_start(Comm *) { return check(g_api.GroupStart(), "ncclGroupStart"); }
int comm_group_end(Comm *) { return check(g_api.GroupEnd(), "ncclGroupEnd"); }

int comm_send(Comm *c, const float *buf, size_t count, int peer, void *stream)
{
    return check(g_api.Send(buf, count, kFloat32, peer, c->handle, stream), "ncclSend");
}

int comm_recv(Comm *c, float *buf, size_t count, int peer, void *stream)
{
    return check(g_api.Recv(buf, count, kFloat32, peer, c->handle, stream), "ncclRecv");
}

} // namespace fdtd
