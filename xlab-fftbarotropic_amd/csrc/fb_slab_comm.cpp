// fb_slab_comm.cpp -- transports of the multi-GPU all-to-all transposes (fb_transport.h): RCCL over xGMI, the
// in-process rehearsal hub, and the caller-supplied callback.  No reference counterpart (the reference is single-process).
//
// RCCL is dlopen'ed on first use instead of being linked: single-GPU users never load it, and inside a torch process the
// loader hands back the librccl.so.1 torch has already mapped (one RCCL per process, like the HIP runtime).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fftbaro.h"
#include "fb_transport.h"

extern "C" void fb_internal_set_error(const char *msg);
static int fail(int code, const std::string &msg) { fb_internal_set_error(msg.c_str()); return code; }
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(FB_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// RCCL
// ------------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void *handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);          // the three below: optional (reporting only)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *);
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return FB_OK;
    const char *names[] = {getenv("FB_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if (n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(FB_EUNSUPPORTED, std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "librccl.so.1 not found"));
    RcclApi a; memset(&a, 0, sizeof(a));
    a.handle = h;
#define SYM(field, name) do { *(void **)(&a.field) = dlsym(h, name); if (!a.field) return fail(FB_EUNSUPPORTED, std::string("RCCL lacks ") + name); } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void **)(&a.CommCount) = dlsym(h, "ncclCommCount");
    *(void **)(&a.CommUserRank) = dlsym(h, "ncclCommUserRank");
    *(void **)(&a.CommCuDevice) = dlsym(h, "ncclCommCuDevice");
    g_rccl = a;
    return FB_OK;
}
#define NCCLCHK(expr)                                                                                                 \
    do {                                                                                                              \
        ncclResult_t r_ = (expr);                                                                                     \
        if (r_ != ncclSuccess) return fail(FB_EHIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));           \
    } while (0)

struct RcclTransport { ncclComm_t comm; int rank, world; bool self_through_rccl; };

int rccl_alltoall(void *self, const float *send, float *recv, size_t stride, size_t offset, size_t count, hipStream_t stream)
{
    RcclTransport *t = (RcclTransport *)self;
    // one group = one fused launch: every peer's block leaves over its own xGMI link at the same time.  The group is always
    // closed, also when a send or receive was refused: an open group would swallow every later RCCL call of this thread.
    NCCLCHK(g_rccl.GroupStart());
    ncclResult_t bad = ncclSuccess;
    const char *what = "";
    for (int p = 0; p < t->world && bad == ncclSuccess; ++p) {
        if (p == t->rank && !t->self_through_rccl) continue;
        bad = g_rccl.Send(send + (size_t)p * stride + offset, count, ncclFloat, p, t->comm, stream);
        if (bad != ncclSuccess) { what = "ncclSend"; break; }
        bad = g_rccl.Recv(recv + (size_t)p * stride + offset, count, ncclFloat, p, t->comm, stream);
        if (bad != ncclSuccess) what = "ncclRecv";
    }
    const ncclResult_t end = g_rccl.GroupEnd();
    if (bad != ncclSuccess) return fail(FB_EHIP, std::string(what) + ": " + g_rccl.GetErrorString(bad));
    if (end != ncclSuccess) return fail(FB_EHIP, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(end));
    if (!t->self_through_rccl)                             // this rank's own block never touches a link
        HIPCHK(hipMemcpyAsync(recv + (size_t)t->rank * stride + offset, send + (size_t)t->rank * stride + offset, count * sizeof(float),
                              hipMemcpyDeviceToDevice, stream));
    return FB_OK;
}
// what the communicator itself says it is: the proof, on a bench line, that RCCL connected `world` ranks (not a name string)
int rccl_info(void *self, int *comm_ranks, int *comm_rank, int *device)
{
    RcclTransport *t = (RcclTransport *)self;
    int v = -1;
    if (comm_ranks) { *comm_ranks = -1; if (g_rccl.CommCount && g_rccl.CommCount(t->comm, &v) == ncclSuccess) *comm_ranks = v; }
    if (comm_rank) { *comm_rank = -1; if (g_rccl.CommUserRank && g_rccl.CommUserRank(t->comm, &v) == ncclSuccess) *comm_rank = v; }
    if (device) { *device = -1; if (g_rccl.CommCuDevice && g_rccl.CommCuDevice(t->comm, &v) == ncclSuccess) *device = v; }
    return FB_OK;
}
void rccl_destroy(void *self)
{
    RcclTransport *t = (RcclTransport *)self;
    if (t->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(t->comm);
    delete t;
}
}  // namespace

extern "C" int fb_slab_unique_id(char *id128)
{
    if (!id128) return fail(FB_EINVAL, "fb_slab_unique_id: NULL");
    int rc = rccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == FB_UNIQUE_ID_BYTES, "FB_UNIQUE_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return FB_OK;
}

int fb_transport_rccl(fb_transport *tp, const char *unique_id, int rank, int world)
{
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    RcclTransport *t = new RcclTransport{nullptr, rank, world, getenv("FB_RCCL_SELF") != nullptr};   // FB_RCCL_SELF: route the own block through RCCL too (1-GPU test of the call path)
    ncclResult_t r = g_rccl.CommInitRank(&t->comm, world, id, rank);
    if (r != ncclSuccess) { delete t; return fail(FB_EHIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)); }
    tp->self = t; tp->rank = rank; tp->world = world; tp->alltoall = rccl_alltoall; tp->destroy = rccl_destroy; tp->name = "rccl";
    tp->info = rccl_info;
    return FB_OK;
}

// ------------------------------------------------------------------------------------------------------------
// local hub: `world` ranks driven by `world` host threads of one process, all on one device
// ------------------------------------------------------------------------------------------------------------
namespace {
struct LocalHub {
    int world;
    std::mutex mu;
    std::condition_variable cv;
    struct Slot { const float *send; size_t stride, offset, count; hipEvent_t ready, done; };
    std::vector<Slot> slot;
    std::vector<long> posted, copied;          // per rank: sequence number of the last posted / copied operation
    bool failed;
    int timeout_s;                             // a rank that never arrives must not hang the others for ever (FB_LOCAL_TIMEOUT_S, default 300)
};
struct LocalTransport { LocalHub *hub; int rank; long seq; };

int local_alltoall(void *self, const float *send, float *recv, size_t stride, size_t offset, size_t count, hipStream_t stream)
{
    LocalTransport *t = (LocalTransport *)self;
    LocalHub *h = t->hub;
    const int r = t->rank, W = h->world;
    const long seq = ++t->seq;
    // 1. publish: my blocks are ready once `ready` has fired on my stream
    HIPCHK(hipEventRecord(h->slot[r].ready, stream));
    {
        std::unique_lock<std::mutex> lk(h->mu);
        h->slot[r].send = send; h->slot[r].stride = stride; h->slot[r].offset = offset; h->slot[r].count = count;
        h->posted[r] = seq;
        h->cv.notify_all();
        // 2. wait until every rank has published the same operation
        const bool ok = h->cv.wait_for(lk, std::chrono::seconds(h->timeout_s), [&] { if (h->failed) return true; for (int p = 0; p < W; ++p) if (h->posted[p] < seq) return false; return true; });
        if (!ok) { h->failed = true; h->cv.notify_all(); return fail(FB_EHIP, "local transport: timed out waiting for the other ranks (a rank left the schedule)"); }
        if (h->failed) return fail(FB_EHIP, "local transport: a peer failed");
        for (int p = 0; p < W; ++p)
            if (h->slot[p].stride != stride || h->slot[p].offset != offset || h->slot[p].count != count) {
                h->failed = true; h->cv.notify_all();
                return fail(FB_EINVAL, "local transport: ranks disagree on the exchange (schedule mismatch)");
            }
    }
    // 3. pull my blocks out of every rank's send buffer, on my stream, behind that rank's `ready`
    for (int p = 0; p < W; ++p) {
        HIPCHK(hipStreamWaitEvent(stream, h->slot[p].ready, 0));
        HIPCHK(hipMemcpyAsync(recv + (size_t)p * stride + offset, h->slot[p].send + (size_t)r * stride + offset, count * sizeof(float),
                              hipMemcpyDeviceToDevice, stream));
    }
    // 4. my copies out of the peers' buffers are done once `done` has fired
    HIPCHK(hipEventRecord(h->slot[r].done, stream));
    {
        std::unique_lock<std::mutex> lk(h->mu);
        h->copied[r] = seq;
        h->cv.notify_all();
        // 5. my send buffer is reusable when every rank has copied its block out of it
        const bool ok = h->cv.wait_for(lk, std::chrono::seconds(h->timeout_s), [&] { if (h->failed) return true; for (int p = 0; p < W; ++p) if (h->copied[p] < seq) return false; return true; });
        if (!ok) { h->failed = true; h->cv.notify_all(); return fail(FB_EHIP, "local transport: timed out waiting for the other ranks (a rank left the schedule)"); }
        if (h->failed) return fail(FB_EHIP, "local transport: a peer failed");
    }
    for (int p = 0; p < W; ++p)
        if (p != r) HIPCHK(hipStreamWaitEvent(stream, h->slot[p].done, 0));
    return FB_OK;
}
void local_destroy(void *self) { delete (LocalTransport *)self; }

int callback_alltoall(void *self, const float *send, float *recv, size_t stride, size_t offset, size_t count, hipStream_t stream)
{
    struct CB { fb_alltoall_fn fn; void *user; } *c = (CB *)self;
    const int rc = c->fn(c->user, send, recv, stride, offset, count, (void *)stream);
    return rc ? fail(FB_EHIP, "transport callback failed") : FB_OK;
}
void callback_destroy(void *self) { free(self); }
}  // namespace

extern "C" int fb_local_hub_create(void **hub, int world)
{
    if (!hub || world < 1) return fail(FB_EINVAL, "fb_local_hub_create: bad argument");
    LocalHub *h = new LocalHub();
    h->world = world; h->failed = false;
    h->timeout_s = getenv("FB_LOCAL_TIMEOUT_S") ? atoi(getenv("FB_LOCAL_TIMEOUT_S")) : 300;
    if (h->timeout_s < 1) h->timeout_s = 1;
    h->slot.resize(world); h->posted.assign(world, 0); h->copied.assign(world, 0);
    for (int p = 0; p < world; ++p) {
        memset(&h->slot[p], 0, sizeof(h->slot[p]));
        if (hipEventCreateWithFlags(&h->slot[p].ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->slot[p].done, hipEventDisableTiming) != hipSuccess) {
            delete h; return fail(FB_EHIP, "fb_local_hub_create: cannot create events (no HIP device?)");
        }
    }
    *hub = h;
    return FB_OK;
}
extern "C" int fb_local_hub_destroy(void *hub)
{
    LocalHub *h = (LocalHub *)hub;
    if (!h) return FB_OK;
    for (auto &s : h->slot) { if (s.ready) hipEventDestroy(s.ready); if (s.done) hipEventDestroy(s.done); }
    delete h;
    return FB_OK;
}

int fb_transport_local(fb_transport *tp, void *hub, int rank, int world)
{
    LocalHub *h = (LocalHub *)hub;
    if (!h || h->world != world || rank < 0 || rank >= world) return fail(FB_EINVAL, "fb_slab_connect_local: hub/world mismatch");
    tp->self = new LocalTransport{h, rank, 0}; tp->rank = rank; tp->world = world;
    tp->alltoall = local_alltoall; tp->destroy = local_destroy; tp->name = "local";
    return FB_OK;
}

int fb_transport_callback(fb_transport *tp, fb_alltoall_fn fn, void *user, int rank, int world)
{
    if (!fn) return fail(FB_EINVAL, "fb_slab_connect_callback: NULL callback");
    struct CB { fb_alltoall_fn fn; void *user; } *c = (CB *)malloc(sizeof(CB));
    if (!c) return fail(FB_ENOMEM, "out of memory");
    c->fn = fn; c->user = user;
    tp->self = c; tp->rank = rank; tp->world = world; tp->alltoall = callback_alltoall; tp->destroy = callback_destroy; tp->name = "callback";
    return FB_OK;
}
