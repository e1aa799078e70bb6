// Gradient all-reduce over xGMI: RCCL bound directly (reference: the
// DistributedDataParallel wrapper of utils/model_utils.py:43-52 whose backward hooks all-reduce gradient buckets over NCCL).
//
// One communicator per process (= per GPU).  ncclAllReduce is enqueued by the caller on a HIP stream it owns (the reducer's side
// stream, event-ordered against the backward stream), in place on windows of the flat fp32 gradient buffer -- no bucket copies.
// librccl is resolved at run time with dlopen("librccl.so.1"): inside a PyTorch process that is the RCCL instance torch already
// loaded (one RCCL per process), in a plain C host it is ROCm's.  Nothing here needs torch: a C host passes raw pointers.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#define TUBER_OK 0
#define TUBER_EINVAL (-1)
#define TUBER_ENOLIB (-2)
#define TUBER_ETIMEDOUT (-3)

namespace {

// the stable subset of the NCCL/RCCL C API (rccl.h): opaque communicator, 128-byte unique id, enum values fixed by the ABI
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
enum { kFloat32 = 7, kBfloat16 = 9, kSum = 0 };

struct Api {
    void* handle = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommCount)(Comm, int*) = nullptr;
    int (*CommUserRank)(Comm, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Api g_api;
std::mutex g_mu;
std::string g_err;

void set_err(const std::string& s) {
    std::lock_guard<std::mutex> l(g_mu);
    g_err = s;
}

int fail(const char* what, int rc) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s (rccl result %d)", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?", rc);
    set_err(buf);
    return rc == 0 ? TUBER_EINVAL : rc;
}

int load_api() {
    static std::once_flag once;
    static int status = TUBER_ENOLIB;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void* h = nullptr;
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) {
            set_err(std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found"));
            return;
        }
        g_api.handle = h;
#define SYM(field, name)                                                   \
    g_api.field = (decltype(g_api.field))dlsym(h, name);                   \
    if (!g_api.field) { set_err(std::string("librccl lacks ") + name); return; }
        SYM(GetVersion, "ncclGetVersion")
        SYM(GetUniqueId, "ncclGetUniqueId")
        SYM(CommInitRank, "ncclCommInitRank")
        SYM(CommDestroy, "ncclCommDestroy")
        SYM(CommCount, "ncclCommCount")
        SYM(CommUserRank, "ncclCommUserRank")
        SYM(AllReduce, "ncclAllReduce")
        SYM(GroupStart, "ncclGroupStart")
        SYM(GroupEnd, "ncclGroupEnd")
        SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        status = TUBER_OK;
    });
    return status;
}

}  // namespace

extern "C" {

// RCCL version code (e.g. 22606) or a negative error when librccl cannot be loaded.
// the return code tuber_comm_init_timeout uses for "a rank never reached the bootstrap" (callers compare against this, not a literal)
int tuber_comm_etimedout(void) { return TUBER_ETIMEDOUT; }

int tuber_comm_version(void) {
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    int v = 0;
    const int rc = g_api.GetVersion(&v);
    return rc == 0 ? v : fail("ncclGetVersion", rc);
}

// message of the last failing tuber_comm_* call (valid until the next failure).
const char* tuber_comm_last_error(void) {
    std::lock_guard<std::mutex> l(g_mu);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

// rank 0: fill the 128-byte rendez-vous id that every rank passes to tuber_comm_init (ship it over any side channel).
int tuber_comm_unique_id(void* id128) {
    if (!id128) return TUBER_EINVAL;
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    UniqueId id;
    const int rc = g_api.GetUniqueId(&id);
    if (rc != 0) return fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof id);
    return TUBER_OK;
}

// collective: every rank calls it with the same id; binds the communicator to HIP device `device`.
int tuber_comm_init(const void* id128, int nranks, int rank, int device, void** comm_out) {
    if (!id128 || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return TUBER_EINVAL;
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { set_err(std::string("hipSetDevice: ") + hipGetErrorString(e)); return (int)e; }
    UniqueId id;
    memcpy(&id, id128, sizeof id);
    Comm c = nullptr;
    const int rc = g_api.CommInitRank(&c, nranks, id, rank);
    if (rc != 0) return fail("ncclCommInitRank", rc);
    *comm_out = c;
    return TUBER_OK;
}

// tuber_comm_init with a deadline.  ncclCommInitRank is itself a collective: a rank that never arrives (died before it, no
// librccl, wrong device) would block every other rank forever.  The bootstrap runs on a helper thread; when it has not returned
// after timeout_ms the call fails with TUBER_ETIMEDOUT and a message naming the rank -- the caller aborts the job with a readable
// error instead of a hang (ddp.py treats it as fatal: the helper thread stays parked inside RCCL holding the device, so running
// another transport's collectives beside the half-made communicator is not safe; if the peers do arrive later the helper destroys
// the communicator it gets).  timeout_ms <= 0: plain tuber_comm_init.
int tuber_comm_init_timeout(const void* id128, int nranks, int rank, int device, int timeout_ms, void** comm_out) {
    if (timeout_ms <= 0) return tuber_comm_init(id128, nranks, rank, device, comm_out);
    if (!id128 || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return TUBER_EINVAL;
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    struct State {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
        int rc = 0;
        void* comm = nullptr;
        bool abandoned = false;      // the caller gave up at the deadline: a communicator that completes later is destroyed, not leaked
        char id[128];
    };
    auto st = std::make_shared<State>();
    memcpy(st->id, id128, sizeof st->id);
    std::thread([st, nranks, rank, device] {
        void* c = nullptr;
        const int rc = tuber_comm_init(st->id, nranks, rank, device, &c);
        bool orphan;
        {
            std::lock_guard<std::mutex> l(st->mu);
            st->rc = rc;
            st->comm = c;
            st->done = true;
            orphan = st->abandoned;
            st->cv.notify_all();
        }
        if (orphan && rc == TUBER_OK && c) g_api.CommDestroy((Comm)c);      // the peers arrived after the deadline: nobody owns this one
    }).detach();
    std::unique_lock<std::mutex> l(st->mu);
    if (!st->cv.wait_for(l, std::chrono::milliseconds(timeout_ms), [&] { return st->done; })) {
        char buf[256];
        snprintf(buf, sizeof buf, "ncclCommInitRank: rank %d of %d (device %d) still waiting for its peers after %d ms -- a rank is missing or "
                 "cannot reach the rendez-vous", rank, nranks, device, timeout_ms);
        set_err(buf);
        st->abandoned = true;
        return TUBER_ETIMEDOUT;
    }
    if (st->rc != 0) return st->rc;
    hipError_t e = hipSetDevice(device);      // the bootstrap ran on another thread: bind the caller's thread too
    if (e != hipSuccess) { set_err(std::string("hipSetDevice: ") + hipGetErrorString(e)); return (int)e; }
    *comm_out = st->comm;
    return TUBER_OK;
}

// ranks RCCL itself reports for this communicator (ncclCommCount) and this process's rank in it (ncclCommUserRank): what the first
// multi-GPU run prints, so "8 processes, each alone in a 1-rank communicator" cannot pass for data parallelism.
int tuber_comm_count(void* comm, int* count_out, int* rank_out) {
    if (!comm || !count_out) return TUBER_EINVAL;
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    int rc = g_api.CommCount((Comm)comm, count_out);
    if (rc != 0) return fail("ncclCommCount", rc);
    if (rank_out) {
        rc = g_api.CommUserRank((Comm)comm, rank_out);
        if (rc != 0) return fail("ncclCommUserRank", rc);
    }
    return TUBER_OK;
}

// in-place sum over all ranks of buf[0..count) (dtype 0 = fp32, 1 = bf16), enqueued on `stream`; capturable into a hipGraph.
int tuber_comm_allreduce_sum(void* comm, void* buf, long count, int dtype, hipStream_t stream) {
    if (!comm || !buf || count <= 0 || (dtype != 0 && dtype != 1)) return TUBER_EINVAL;
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    const int rc = g_api.AllReduce(buf, buf, (size_t)count, dtype == 0 ? kFloat32 : kBfloat16, kSum, (Comm)comm, stream);
    return rc == 0 ? TUBER_OK : fail("ncclAllReduce", rc);
}

// several windows as ONE RCCL group (one launch on the stream): ptrs[i] / counts[i] are host arrays of n entries.
int tuber_comm_allreduce_sum_multi(void* comm, void* const* ptrs, const long* counts, int n, int dtype, hipStream_t stream) {
    if (!comm || !ptrs || !counts || n <= 0 || (dtype != 0 && dtype != 1)) return TUBER_EINVAL;
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    int rc = g_api.GroupStart();
    if (rc != 0) return fail("ncclGroupStart", rc);
    for (int i = 0; i < n; ++i) {
        if (counts[i] <= 0) continue;
        rc = g_api.AllReduce(ptrs[i], ptrs[i], (size_t)counts[i], dtype == 0 ? kFloat32 : kBfloat16, kSum, (Comm)comm, stream);
        if (rc != 0) { g_api.GroupEnd(); return fail("ncclAllReduce", rc); }
    }
    rc = g_api.GroupEnd();
    return rc == 0 ? TUBER_OK : fail("ncclGroupEnd", rc);
}

int tuber_comm_destroy(void* comm) {
    if (!comm) return TUBER_EINVAL;
    if (load_api() != TUBER_OK) return TUBER_ENOLIB;
    const int rc = g_api.CommDestroy((Comm)comm);
    return rc == 0 ? TUBER_OK : fail("ncclCommDestroy", rc);
}

}  // extern "C"
