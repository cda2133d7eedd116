// comm.hip -- the multi-GPU merge step behind the C-ABI: RCCL collectives over xGMI issued by the library itself.
//
// The reference is one process that tracks its cameras serially and reads every camera's features and pose directly
// (src/app/SL_CoSLAM.cpp:299-305, src/app/SL_InterCamPoseEstimator.cpp:24-37); there is no collective to mirror.  With
// the cameras sharded over the GPUs of a node (one process per GPU) the same information travels in
//   collective 1  (every frame)   ONE all-gather of a fixed-size record per camera: N x KLT_TrackedFeature || R || t,
//                                 packed by ONE kernel for all of the rank's cameras (cs_exchange_allgather_dev);
//   collective 2  (per LM step)   the joint bundle adjustment sliced by points: ONE all-reduce of S || rhs per LM step,
//                                 four scalars for the LM / outlier decisions, points and flags once at the end
//                                 (cs_ba_dist_solve: the whole schedule of coslam_amd/csrc/ba.hip's phase API, enqueued
//                                 from C++ with no host synchronisation).
// RCCL is loaded at run time (dlopen): the library keeps loading on a box without it, and a process that already
// carries PyTorch's copy shares it.  Payloads are small (40 KB per camera; 166 KB of S at order 144): latency-bound on
// xGMI, so each is ONE collective, never chunked.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <new>
#include <thread>

#include "cs_common.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclAllReduce) allReduce = nullptr;
    decltype(&ncclBroadcast) broadcast = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
};

RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) {
        cs_set_error("RCCL not found (dlopen librccl.so.1): %s", dlerror());
        return nullptr;
    }
    api.getUniqueId = (decltype(api.getUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.commInitRank = (decltype(api.commInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.commDestroy = (decltype(api.commDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.allGather = (decltype(api.allGather))dlsym(api.handle, "ncclAllGather");
    api.allReduce = (decltype(api.allReduce))dlsym(api.handle, "ncclAllReduce");
    api.getErrorString = (decltype(api.getErrorString))dlsym(api.handle, "ncclGetErrorString");
    api.broadcast = (decltype(api.broadcast))dlsym(api.handle, "ncclBroadcast");
    if (!api.broadcast || !api.getUniqueId || !api.commInitRank || !api.commDestroy || !api.allGather || !api.allReduce || !api.getErrorString) {
        cs_set_error("RCCL library lacks a required symbol");
        dlclose(api.handle);
        api.handle = nullptr;
        return nullptr;
    }
    return &api;
}

#define CS_NCCL(call)                                                                                     \
    do {                                                                                                  \
        ncclResult_t _r = (call);                                                                         \
        if (_r != ncclSuccess) {                                                                          \
            cs_set_error("%s failed: %s (%s:%d)", #call, rccl_api()->getErrorString(_r), __FILE__, __LINE__); \
            return CS_ERR_HIP;                                                                            \
        }                                                                                                 \
    } while (0)

}  // namespace

// ---- the test transport: ranks that SHARE one GPU (RCCL refuses two ranks on a device) -----------------------------------------------
// cs_comm_create_host: the same collectives staged through a POSIX shared-memory segment -- stream synchronised, device -> segment,
// barrier, segment -> device, barrier.  It blocks the host and moves every byte twice: it exists so that the N > 1 code paths of a
// frame loop can be RUN where only one GPU is visible (tests/test_cxx_dropin_gpu.py: two ranks of tools/cxx/frame_loop.bin on one
// MI355X end in the one-rank run's map), the role gloo plays for the Python loop's tests.  Never the transport of a measured number.
struct HostShm {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    int world;
    size_t capacity;   // bytes of payload behind the header
    // who made the segment and when (a rank could attach to a same-named segment a crashed run left behind -- rank 0 unlinks and
    // re-creates it, but an attaching rank may open the OLD name first; the stale sizes and counters passed every check and the ranks waited out
    // 60 s on different segments): rank 0 stamps its segment, an attaching rank takes only one that is stamped, younger than two minutes and
    // not yet full, and opens the name again otherwise
    std::atomic<unsigned long long> magic;
    long long createdNs;
};
constexpr unsigned long long CS_HOST_MAGIC = 0x43534C414D484F53ull;   // "CSLAMHOS"
static long long host_now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}
constexpr size_t CS_HOST_SHM_BYTES = (size_t)96 << 20;

struct cs_comm {
    ncclComm_t comm;
    int world, rank, device;
    HostShm* host = nullptr;   // non-null: the test transport
    unsigned char* hostData = nullptr;
    char hostName[96] = {0};
};

namespace {
int host_barrier(cs_comm* c) {
    HostShm* h = c->host;
    const int gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.store(gen + 1, std::memory_order_release);
        return CS_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load(std::memory_order_acquire) == gen) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
            cs_set_error("host transport: a rank did not reach the barrier within 60 s");
            return CS_ERR_HIP;
        }
    }
    return CS_OK;
}
// every rank's `bytes` at d_send to every rank's d_recv; root >= 0: only root's part travels, into d_recv of every rank
int host_collect(cs_comm* c, hipStream_t s, const void* d_send, void* d_recv, size_t bytes, int root) {
    const size_t total = root >= 0 ? bytes : bytes * (size_t)c->world;
    if (total > c->host->capacity) {
        cs_set_error("host transport: %zu bytes exceed the segment's %zu", total, c->host->capacity);
        return CS_ERR_INVALID;
    }
    CS_HIP(hipStreamSynchronize(s));
    if (root < 0 || root == c->rank)
        CS_HIP(hipMemcpy(c->hostData + (root >= 0 ? 0 : bytes * (size_t)c->rank), d_send, bytes, hipMemcpyDeviceToHost));
    int rc = host_barrier(c);
    if (rc) return rc;
    if (root < 0 || root != c->rank) CS_HIP(hipMemcpy(d_recv, c->hostData, total, hipMemcpyHostToDevice));
    return host_barrier(c);   // (nobody overwrites the segment before everybody has read it)
}
}  // namespace

constexpr int CS_EX_MAX_CAMS = 16;
struct cs_exchange {
    cs_comm* c;
    int nCams, nFeat;
    size_t recWords;  // int32 words of one camera's record
    int* send;
    int* recv;
};

namespace {

struct PackArgs {
    int nFeatWords;  // N * 5
    int recWords;
    int* send;
    const double* R;
    const double* t;
    const int* dest[CS_EX_MAX_CAMS];
};

// one launch packs every local camera's record: grid.y = camera
__global__ __launch_bounds__(256) void k_exchange_pack(PackArgs A) {
    const int cam = blockIdx.y;
    int* out = A.send + (size_t)cam * A.recWords;
    const int* src = A.dest[cam];
    for (int q = blockIdx.x * 256 + threadIdx.x; q < A.recWords; q += gridDim.x * 256) {
        int v;
        const int fp = (A.nFeatWords + 1) & ~1;
        if (q < A.nFeatWords) {
            v = src[q];
        } else if (q < fp) {
            v = 0;  // (padding word of an odd feature count)
        } else {
            const int w = q - fp;  // 18 words of R, 6 of t
            const int* p = (w < 18) ? (const int*)(A.R + 9 * (size_t)cam) + w : (const int*)(A.t + 3 * (size_t)cam) + (w - 18);
            v = *p;
        }
        out[q] = v;
    }
}

// every gathered camera's pose out of the records into contiguous [nCamsAll][9] / [nCamsAll][3] arrays (the layout the pose
// update, the registration and the window take); cameras [skip0, skip0 + nSkip) are left alone (the rank's own: already there)
__global__ __launch_bounds__(256) void k_exchange_unpack_poses(const int* __restrict__ recv, int recWords, int nFeatWords, int nCamsAll, int skip0,
                                                               int nSkip, double* __restrict__ R, double* __restrict__ t) {
    const int q = blockIdx.x * 256 + threadIdx.x, g = q / 12, e = q - 12 * g;
    if (g >= nCamsAll || (g >= skip0 && g < skip0 + nSkip)) return;
    const double* p = (const double*)(recv + (size_t)g * recWords + ((nFeatWords + 1) & ~1));
    if (e < 9)
        R[9 * (size_t)g + e] = p[e];
    else
        t[3 * (size_t)g + (e - 9)] = p[e];
}

}  // namespace

extern "C" {

// 1: RCCL could be loaded and has every entry point used here (what a group of ranks should agree on BEFORE any of them enters
// ncclCommInitRank: a rank that cannot load the library would leave the others waiting inside it)
int cs_comm_available(void) { return rccl_api() ? 1 : 0; }

int cs_comm_unique_id(unsigned char id[128]) {
    RcclApi* api = rccl_api();
    if (!api || !id) return CS_ERR_INVALID;
    ncclUniqueId u;
    CS_NCCL(api->getUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return CS_OK;
}

cs_comm* cs_comm_create(const unsigned char id[128], int world, int rank, int device) {
    RcclApi* api = rccl_api();
    if (!api) return nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) {
        cs_set_error("cs_comm_create: bad arguments");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        cs_set_error("cs_comm_create: cannot select device %d", device);
        return nullptr;
    }
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t r = api->commInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) {
        cs_set_error("ncclCommInitRank failed: %s", api->getErrorString(r));
        return nullptr;
    }
    cs_comm* c = new (std::nothrow) cs_comm();
    if (!c) return nullptr;
    c->comm = comm;
    c->world = world;
    c->rank = rank;
    c->device = device;
    return c;
}

// the test transport (above): every rank names the same segment; rank 0 creates it, the others attach
cs_comm* cs_comm_create_host(const char* name, int world, int rank, int device) {
    if (!name || !name[0] || strlen(name) > 90 || world < 1 || rank < 0 || rank >= world) {
        cs_set_error("cs_comm_create_host: bad arguments");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        cs_set_error("cs_comm_create_host: cannot select device %d", device);
        return nullptr;
    }
    const size_t bytes = sizeof(HostShm) + CS_HOST_SHM_BYTES;
    void* m = MAP_FAILED;
    if (rank == 0) {
        (void)shm_unlink(name);
        int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)bytes) != 0) {
            close(fd);
            fd = -1;
        }
        if (fd < 0) {
            cs_set_error("cs_comm_create_host: cannot create the shared-memory segment %s", name);
            return nullptr;
        }
        m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m != MAP_FAILED) {   // (a fresh segment is zero-filled: counters at 0)
            HostShm* h = (HostShm*)m;
            h->capacity = CS_HOST_SHM_BYTES, h->world = world, h->createdNs = host_now_ns();
            h->magic.store(CS_HOST_MAGIC, std::memory_order_release);
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (m == MAP_FAILED && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(60)) {
            int fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes)) {   // (created, not yet sized)
                close(fd);
                fd = -1;
            }
            if (fd >= 0) {
                void* q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                close(fd);
                if (q != MAP_FAILED) {
                    HostShm* h = (HostShm*)q;
                    // this run's segment: stamped by rank 0 (it may be a moment away from that), made within the last two minutes, not full
                    bool mine = false;
                    for (int tries = 0; tries < 40 && !mine; ++tries) {
                        mine = h->magic.load(std::memory_order_acquire) == CS_HOST_MAGIC && h->world == world && h->attached.load() < world &&
                               llabs(host_now_ns() - h->createdNs) < 120LL * 1000000000LL;
                        if (!mine) std::this_thread::sleep_for(std::chrono::milliseconds(5));
                    }
                    if (mine)
                        m = q;
                    else
                        munmap(q, bytes);   // (a segment a crashed run left behind, or one rank 0 is about to replace: open the name again)
                }
            }
            if (m == MAP_FAILED) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
    }
    if (m == MAP_FAILED) {
        cs_set_error("cs_comm_create_host: cannot open / map this run's shared-memory segment %s", name);
        return nullptr;
    }
    cs_comm* c = new (std::nothrow) cs_comm();
    if (!c) return nullptr;
    c->comm = nullptr, c->world = world, c->rank = rank, c->device = device;
    c->host = (HostShm*)m, c->hostData = (unsigned char*)m + sizeof(HostShm);
    snprintf(c->hostName, sizeof(c->hostName), "%s", name);
    c->host->attached.fetch_add(1);   // (the others wait for `world` before their first barrier)
    const auto t0 = std::chrono::steady_clock::now();
    while (c->host->attached.load() < world || c->host->world != world) {
        std::this_thread::yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
            cs_set_error("cs_comm_create_host: %d of %d ranks attached within 60 s", c->host->attached.load(), world);
            munmap(m, bytes);
            delete c;
            return nullptr;
        }
    }
    return c;
}

void cs_comm_destroy(cs_comm* c) {
    if (!c) return;
    if (c->host) {
        munmap((void*)c->host, sizeof(HostShm) + CS_HOST_SHM_BYTES);
        if (c->rank == 0) (void)shm_unlink(c->hostName);
        delete c;
        return;
    }
    RcclApi* api = rccl_api();
    if (api && c->comm) (void)api->commDestroy(c->comm);
    delete c;
}

int cs_comm_world(const cs_comm* c) { return c ? c->world : 0; }
int cs_comm_rank(const cs_comm* c) { return c ? c->rank : -1; }

// ---- collective 1: features || pose of every camera to every rank ---------------------------------------------------
cs_exchange* cs_exchange_create(cs_comm* c, int nCamsLocal, int nFeatures) {
    if (!c || nCamsLocal < 1 || nCamsLocal > CS_EX_MAX_CAMS || nFeatures < 1) {
        cs_set_error("cs_exchange_create: bad arguments (1..%d local cameras)", CS_EX_MAX_CAMS);
        return nullptr;
    }
    cs_exchange* x = new (std::nothrow) cs_exchange();
    if (!x) return nullptr;
    x->c = c;
    x->nCams = nCamsLocal;
    x->nFeat = nFeatures;
    x->recWords = (((size_t)nFeatures * 5 + 1) & ~(size_t)1) + 24;  // features padded to 8 bytes: R | t behind them stay aligned
    x->send = x->recv = nullptr;
    const size_t sendBytes = sizeof(int) * x->recWords * nCamsLocal;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void**)&x->send, sendBytes) != hipSuccess ||
        hipMalloc((void**)&x->recv, sendBytes * c->world) != hipSuccess) {
        cs_set_error("cs_exchange_create: allocation failed");
        if (x->send) (void)hipFree(x->send);
        delete x;
        return nullptr;
    }
    return x;
}

void cs_exchange_destroy(cs_exchange* x) {
    if (!x) return;
    (void)hipFree(x->send);
    (void)hipFree(x->recv);
    delete x;
}

int cs_exchange_allgather_dev(cs_exchange* x, void* hip_stream, const void* const* d_dests, const double* d_R,
                              const double* d_t) {
    RcclApi* api = (x && x->c->host) ? nullptr : rccl_api();
    if ((!api && !(x && x->c->host)) || !x || !d_dests || !d_R || !d_t) {
        cs_set_error("cs_exchange_allgather_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(x->c->device));
    hipStream_t s = (hipStream_t)hip_stream;
    PackArgs A;
    memset(&A, 0, sizeof(A));
    A.nFeatWords = x->nFeat * 5;
    A.recWords = (int)x->recWords;
    A.send = x->send;
    A.R = d_R;
    A.t = d_t;
    for (int i = 0; i < x->nCams; ++i) {
        if (!d_dests[i]) {
            cs_set_error("cs_exchange_allgather_dev: null dest[] of camera %d", i);
            return CS_ERR_INVALID;
        }
        A.dest[i] = (const int*)d_dests[i];
    }
    int gx = (int)((x->recWords + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_exchange_pack, dim3(gx, x->nCams), dim3(256), 0, s, A);
    CS_CHECK_LAUNCH();
    const size_t sendBytes = sizeof(int) * x->recWords * x->nCams;
    if (x->c->host) return host_collect(x->c, s, x->send, x->recv, sendBytes, -1);
    CS_NCCL(api->allGather(x->send, x->recv, sendBytes, ncclInt8, x->c->comm, s));
    return CS_OK;
}

// gathered records: global camera g (rank g / nCamsLocal, local index g % nCamsLocal) at d_recv + g * record_bytes:
// N x cs_klt_feature, then R (9 doubles), then t (3 doubles)
int cs_exchange_buffers(cs_exchange* x, void** d_recv, size_t* record_bytes) {
    if (!x) return CS_ERR_INVALID;
    if (d_recv) *d_recv = x->recv;
    if (record_bytes) *record_bytes = sizeof(int) * x->recWords;
    return CS_OK;
}

// the poses of every gathered camera into d_R [world * nCamsLocal][9], d_t [..][3]; skipOwn != 0: this rank's own cameras are not
// written (the arrays already hold them: the pose solve wrote them in place)
int cs_exchange_unpack_poses_dev(cs_exchange* x, void* hip_stream, double* d_R, double* d_t, int skipOwn) {
    if (!x || !d_R || !d_t) {
        cs_set_error("cs_exchange_unpack_poses_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(x->c->device));
    const int nAll = x->nCams * x->c->world;
    hipLaunchKernelGGL(k_exchange_unpack_poses, dim3((nAll * 12 + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, x->recv, (int)x->recWords,
                       x->nFeat * 5, nAll, x->c->rank * x->nCams, skipOwn ? x->nCams : 0, d_R, d_t);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// one buffer from rank `root` to every rank, in place, on hip_stream (ncclBroadcast): the packed result of a bundle adjustment
// (cs_ba_output_*) from the rank that solved the window to every replica of the map
int cs_comm_broadcast_dev(cs_comm* c, void* hip_stream, void* d_buf, size_t bytes, int root) {
    RcclApi* api = (c && c->host) ? nullptr : rccl_api();
    if ((!api && !(c && c->host)) || !c || !d_buf || root < 0 || root >= c->world) {
        cs_set_error("cs_comm_broadcast_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    if (c->world == 1 || bytes == 0) return CS_OK;
    CS_HIP(hipSetDevice(c->device));
    if (c->host) return host_collect(c, (hipStream_t)hip_stream, d_buf, d_buf, bytes, root);
    CS_NCCL(api->broadcast(d_buf, d_buf, bytes, ncclInt8, root, c->comm, (hipStream_t)hip_stream));
    return CS_OK;
}

// every rank's `bytes` at d_send to every rank's d_recv (rank r's part at r * bytes), on hip_stream (ncclAllGather): the candidate
// tables of the registration search (cs_register_candidates_pack_dev) before the decision every rank then takes on its replica
int cs_comm_allgather_dev(cs_comm* c, void* hip_stream, const void* d_send, void* d_recv, size_t bytes) {
    RcclApi* api = (c && c->host) ? nullptr : rccl_api();
    if ((!api && !(c && c->host)) || !c || !d_send || !d_recv) {
        cs_set_error("cs_comm_allgather_dev: bad arguments");
        return CS_ERR_INVALID;
    }
    CS_HIP(hipSetDevice(c->device));
    if (c->host) return host_collect(c, (hipStream_t)hip_stream, d_send, d_recv, bytes, -1);
    CS_NCCL(api->allGather(d_send, d_recv, bytes, ncclInt8, c->comm, (hipStream_t)hip_stream));
    return CS_OK;
}

// ---- collective 2: bundleAdjustRobust over all ranks, points sliced by rank ------------------------------------------
// Every rank passes the same replicated problem (cs_ba_upload) and ends with the same result (cs_ba_download).
// Everything -- phases and collectives -- is enqueued on hip_stream; the host never synchronises.
int cs_ba_dist_solve(cs_ba* b, cs_comm* c, void* hip_stream, int C, int P, int nObs, const double* d_Rs0, const double* d_Ts0,
                     const double* d_pts0, int nCamsCon, int nPtsCon, double maxErr, int maxIter, int innerMaxIter) {
    RcclApi* api = rccl_api();
    if (!api || !b || !c || c->host) {
        cs_set_error("cs_ba_dist_solve: bad arguments (RCCL communicators only: the host test transport has no all-reduce)");
        return CS_ERR_INVALID;
    }
    // ONE stream for the phases and the collectives: a NULL argument means the workspace's own stream for both (the phases
    // would pick it on their own while the all-reduces went to the legacy null stream, unordered against them)
    if (!hip_stream) hip_stream = cs_ba_stream(b);
    hipStream_t s = (hipStream_t)hip_stream;
    const int per = (P + c->world - 1) / c->world;
    int lo = c->rank * per;
    if (lo > P) lo = P;
    int hi = lo + per;
    if (hi > P) hi = P;
    int rc = cs_ba_dist_begin(b, hip_stream, C, P, nObs, d_Rs0, d_Ts0, d_pts0, nCamsCon, nPtsCon, maxErr, innerMaxIter, lo, hi,
                              c->rank == 0 ? 1 : 0);
    if (rc) return rc;
    void *dS = nullptr, *dScal = nullptr, *dPts = nullptr, *dOut = nullptr;
    int nRed = 0;
    rc = cs_ba_dist_buffers(b, &dS, &nRed, &dScal, &dPts, &dOut);
    if (rc) return rc;
    auto phase = [&](int ph) { return cs_ba_dist_phase(b, hip_stream, ph); };
    auto sum_d = [&](void* buf, size_t n) -> int {
        if (c->world == 1 || n == 0) return CS_OK;
        CS_NCCL(api->allReduce(buf, buf, n, ncclFloat64, ncclSum, c->comm, s));
        return CS_OK;
    };
    for (int outer = 0; outer < maxIter; ++outer) {
        if ((rc = phase(CS_BA_PH_COST0))) return rc;
        if ((rc = sum_d(dScal, 4))) return rc;
        if ((rc = phase(CS_BA_PH_CONTROL0))) return rc;
        for (int it = 0; it < innerMaxIter; ++it) {
            if ((rc = phase(CS_BA_PH_LIN_SCHUR))) return rc;
            if ((rc = sum_d(dS, (size_t)nRed))) return rc;
            if ((rc = phase(CS_BA_PH_SOLVE_UPDATE))) return rc;
            if ((rc = sum_d(dScal, 4))) return rc;
            if ((rc = phase(CS_BA_PH_CONTROL1))) return rc;
        }
        if ((rc = phase(CS_BA_PH_FLAG))) return rc;
        if ((rc = sum_d(dScal, 4))) return rc;
        if ((rc = phase(CS_BA_PH_OUTER_END))) return rc;
    }
    if ((rc = phase(CS_BA_PH_FINAL_PREP))) return rc;
    if ((rc = sum_d(dPts, (size_t)3 * P))) return rc;
    if (c->world > 1 && nObs > 0) CS_NCCL(api->allReduce(dOut, dOut, (size_t)nObs, ncclInt32, ncclSum, c->comm, s));
    return phase(CS_BA_PH_FINISH);
}

}  // extern "C"
