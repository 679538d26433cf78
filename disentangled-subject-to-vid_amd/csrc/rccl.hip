// Replica set-up over RCCL behind the C ABI (SURVEY.md section 8e / 8b `s2v_bcast_weights(rccl_comm)`): the ONE collective of the path,
// the broadcast of a finalized model's weight arena rank `root` -> all ranks of a node over xGMI.  No reference code exists for this (the
// reference is single-process); the protocol is RCCL's own: rank 0 draws a 128-byte unique id, every rank joins with
// ncclCommInitRank, ncclBroadcast moves bytes in place on the caller's stream.
//
// RCCL is bound at FIRST USE with dlopen("librccl.so.1"): a process that already carries RCCL (PyTorch-ROCm ships one under the same
// soname) shares that copy, a plain C host gets /opt/rocm/lib's; a single-GPU user of libs2v_hip.so never loads it.  Missing library or
// symbol = an error return with the reason in s2v_last_error(), never a silent skip.
#define S2V_HOST
#include <dlfcn.h>
#include <string.h>

#include <mutex>
#include <string>

#include "../../include/s2v_hip.h"
#include "common.h"

namespace {

// the slice of rccl.h this file needs (RCCL keeps NCCL's ABI: ncclUniqueId is 128 opaque bytes, ncclUint8 = 1)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*GroupFn)(void);

struct Rccl {
    void* so = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    BroadcastFn broadcast = nullptr;
    AllGatherFn all_gather = nullptr;
    GetErrorStringFn error_string = nullptr;
    GroupFn group_start = nullptr, group_end = nullptr;
    std::string why;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void bind_rccl() {
    const char* names[] = {getenv("S2V_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.so) break;
        g_rccl.why = dlerror();
    }
    if (!g_rccl.so) return;
    struct { const char* name; void** fn; } syms[] = {
        {"ncclGetUniqueId", (void**)&g_rccl.get_unique_id}, {"ncclCommInitRank", (void**)&g_rccl.comm_init_rank},
        {"ncclCommDestroy", (void**)&g_rccl.comm_destroy}, {"ncclBroadcast", (void**)&g_rccl.broadcast}, {"ncclAllGather", (void**)&g_rccl.all_gather},
        {"ncclGetErrorString", (void**)&g_rccl.error_string}, {"ncclGroupStart", (void**)&g_rccl.group_start},
        {"ncclGroupEnd", (void**)&g_rccl.group_end}};
    for (auto& s : syms) {
        *s.fn = dlsym(g_rccl.so, s.name);
        if (!*s.fn) { g_rccl.why = std::string("symbol missing in librccl: ") + s.name; g_rccl.so = nullptr; return; }
    }
}

int need_rccl() {
    std::call_once(g_rccl_once, bind_rccl);
    if (g_rccl.so) return 0;
    const std::string m = "RCCL is not available (dlopen librccl.so.1): " + g_rccl.why;
    return s2v_fail(__FILE__, __LINE__, m.c_str(), -4);
}

int nccl_fail(const char* what, int rc) {
    const std::string m = std::string(what) + ": " + (g_rccl.error_string ? g_rccl.error_string(rc) : "rccl error");
    return s2v_fail(__FILE__, __LINE__, m.c_str(), -4);
}

}  // namespace

struct s2v_rccl_comm {
    Comm comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int s2v_rccl_available(void) { return need_rccl(); }

extern "C" int s2v_rccl_unique_id(void* id128) {
    S2V_REQUIRE(id128, "s2v_rccl_unique_id: null argument");
    S2V_TRY(need_rccl());
    UniqueId id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

extern "C" int s2v_rccl_comm_create(const void* id128, int32_t rank, int32_t world, s2v_rccl_comm** out) {
    S2V_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "s2v_rccl_comm_create: bad argument");
    S2V_TRY(need_rccl());
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    s2v_rccl_comm* c = new s2v_rccl_comm();
    c->rank = rank; c->world = world;
    {  // the communicator is bound to the CURRENT device (one process per GPU)
        const hipError_t e = hipGetDevice(&c->device);
        if (e != hipSuccess) { delete c; return s2v_fail(__FILE__, __LINE__, hipGetErrorString(e), -2); }
    }
    const int rc = g_rccl.comm_init_rank(&c->comm, world, id, rank);
    if (rc) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    *out = c;
    return 0;
}

extern "C" void s2v_rccl_comm_destroy(s2v_rccl_comm* c) {
    if (!c) return;
    if (c->comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(c->comm);
    delete c;
}

// In place, on `stream`, in chunks: each collective is large enough to run an xGMI ring at its link rate (>= 64 MB) and small enough
// that a failed rank is noticed between chunks; one ncclGroup per call keeps the launches back to back.
extern "C" int s2v_rccl_bcast(s2v_rccl_comm* c, void* dev_ptr, int64_t bytes, int32_t root, s2v_stream stream) {
    S2V_REQUIRE(c && c->comm && dev_ptr && bytes >= 0 && root >= 0 && root < c->world, "s2v_rccl_bcast: bad argument");
    if (bytes == 0) return 0;  // world == 1 still goes through ncclBroadcast (a one-rank communicator is valid): the call path is the same
    const int64_t chunk = (int64_t)256 << 20;
    int rc = g_rccl.group_start();
    if (rc) return nccl_fail("ncclGroupStart", rc);
    for (int64_t off = 0; off < bytes && !rc; off += chunk) {
        char* p = (char*)dev_ptr + off;
        const int64_t n = bytes - off < chunk ? bytes - off : chunk;
        rc = g_rccl.broadcast(p, p, (size_t)n, /*ncclUint8*/ 1, root, c->comm, (hipStream_t)stream);
    }
    const int rc_end = g_rccl.group_end();
    if (rc) return nccl_fail("ncclBroadcast", rc);
    if (rc_end) return nccl_fail("ncclGroupEnd", rc_end);
    return 0;
}

// The per-step exchange of CFG-parallel (api.hip s2v_denoise_step_cfg_parallel): every rank contributes `bytes_per_rank`, `recv` holds
// world x bytes_per_rank in rank order on every rank; send == recv + rank * bytes_per_rank is the in-place form.  2.2 MB per rank at
// 49 x 480 x 720: latency-bound on one xGMI hop, never inside a captured graph.
extern "C" int s2v_rccl_allgather(s2v_rccl_comm* c, const void* send, void* recv, int64_t bytes_per_rank, s2v_stream stream) {
    S2V_REQUIRE(c && c->comm && send && recv && bytes_per_rank >= 0, "s2v_rccl_allgather: bad argument");
    if (bytes_per_rank == 0) return 0;
    const int rc = g_rccl.all_gather(send, recv, (size_t)bytes_per_rank, /*ncclUint8*/ 1, c->comm, (hipStream_t)stream);
    if (rc) return nccl_fail("ncclAllGather", rc);
    return 0;
}

// s2v_denoise_step_cfg_parallel's communicator: the two ranks of ONE pair, rank = the half of the CFG pair the caller computes
int s2v_rccl_pair_check(s2v_rccl_comm* c, int slot) {
    S2V_REQUIRE(c && c->comm, "s2v_denoise_step_cfg_parallel: null communicator");
    S2V_REQUIRE(c->world == 2 && c->rank == slot, "s2v_denoise_step_cfg_parallel: the communicator must hold exactly the two ranks of the pair, and its rank must equal `slot`");
    return 0;
}

// The transformer's merged weights (LoRA merged, QKV fused, fp8 copies and scales included) root -> all; a receiver's tensors are
// marked loaded once the broadcast has been ENQUEUED -- like every entry point this is asynchronous: work submitted to `stream`
// afterwards sees the weights, the host must not read them before the stream has passed.
extern "C" int s2v_bcast_weights(s2v_ctx* ctx, s2v_rccl_comm* c, int32_t root, s2v_stream stream) {
    S2V_REQUIRE(ctx && c, "s2v_bcast_weights: null argument");
    void* p = nullptr;
    int64_t n = 0;
    S2V_TRY(s2v_weight_arena(ctx, &p, &n));
    S2V_TRY(s2v_rccl_bcast(c, p, n, root, stream));
    if (c->rank != root) S2V_TRY(s2v_mark_weights_loaded(ctx));
    return 0;
}
