// ZygpuGroup: the GPUs of one node behind the C ABI (include/zygpu.h "Device groups"; the seam is the
// reference's Driver.render / renderIterations handing tiles to its workers, driver.zig:115-152, 293-348: here the workers are
// devices).  One ZygpuContext and one host thread per device, the compiled scene replicated, a frame's iterations (or rows) split
// into one contiguous share per rank of the world, every rank accumulating into its own film {sum w rgb, sum w}
// (buffer_opaque.zig:39-45: additive, so the shares' films add up to the frame's), and ONE exchange step: ncclReduce(sum) of the films
// onto rank 0 over xGMI (RCCL), in stream order behind each rank's kernels.  No collective on the data path.
//
// A world may span processes (one process per GPU under torch.distributed.run, or an MPI-style Zig host): rank 0 makes the
// communicator's id (zygpu_group_unique_id), the launcher carries its 128 bytes to the others, every process creates its group with the
// id, its first rank and the world's size.  Without an id the process is the whole world.
//
// Contexts that share a device (a group may list a device twice: how a box with one GPU tests the group path) are driven one after
// the other, because the kernel units' __constant__ blocks are per device, and their films are summed with a kernel instead of RCCL
// (which refuses a device that appears twice in a communicator).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/zygpu.h"

namespace {

thread_local std::string g_group_error;

// what an entry point returns: a failure's text becomes zygpu_last_error()'s (zygpu.hip) in the calling thread
extern "C" void zygpu_internal_set_error(const char* text);
int done(int rc) {
    if (0 != rc && !g_group_error.empty()) zygpu_internal_set_error(g_group_error.c_str());
    return rc;
}

__global__ void __launch_bounds__(256) film_add(float4* __restrict__ into, const float4* __restrict__ from, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = into[i], b = from[i];
    into[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

__global__ void __launch_bounds__(256) film_min(float4* __restrict__ into, const float4* __restrict__ from, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = into[i], b = from[i];
    into[i] = make_float4(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w));
}

__global__ void __launch_bounds__(256) plane_add(float* __restrict__ into, const float* __restrict__ from, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) into[i] += from[i];
}

// A member's host thread: runs the jobs the group hands it, one at a time
class Worker {
   public:
    Worker() : thread_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            quit_ = true;
        }
        wake_.notify_all();
        thread_.join();
    }
    void start(std::function<int()> job) {
        std::lock_guard<std::mutex> lock(mutex_);
        job_ = std::move(job);
        busy_ = true;
        wake_.notify_all();
    }
    int wait(std::string& error) {
        std::unique_lock<std::mutex> lock(mutex_);
        done_.wait(lock, [this] { return !busy_; });
        if (0 != result_) error = error_;
        return result_;
    }

   private:
    void loop() {
        std::unique_lock<std::mutex> lock(mutex_);
        for (;;) {
            wake_.wait(lock, [this] { return quit_ || (busy_ && job_); });
            if (quit_) return;
            std::function<int()> job = std::move(job_);
            job_ = nullptr;
            lock.unlock();
            const int         r = job();
            const std::string e = 0 != r ? (g_group_error.empty() ? std::string(zygpu_last_error()) : g_group_error) : std::string();
            g_group_error.clear();
            lock.lock();
            result_ = r;
            error_ = e;
            busy_ = false;
            done_.notify_all();
        }
    }
    std::mutex              mutex_;
    std::condition_variable wake_, done_;
    std::function<int()>    job_;
    bool                    busy_ = false, quit_ = false;
    int                     result_ = 0;
    std::string             error_;
    std::thread             thread_;
};

struct Member {
    ZygpuContext* ctx = nullptr;
    int           device = 0, rank = 0;
    ncclComm_t    comm = nullptr;
    Worker*       worker = nullptr;
};

bool ncclOk(ncclResult_t r, const char* what) {
    if (ncclSuccess == r) return true;
    g_group_error = std::string(what) + ": " + ncclGetErrorString(r);
    return false;
}
bool hipOk(hipError_t e, const char* what) {
    if (hipSuccess == e) return true;
    g_group_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

// rank r's share of [begin, begin + count) split into `world` contiguous shares, the first count % world one longer
void shareOf(uint32_t count, int world, int rank, uint32_t& first, uint32_t& length) {
    const uint32_t base = count / uint32_t(world), rest = count % uint32_t(world);
    first = uint32_t(rank) * base + std::min(uint32_t(rank), rest);
    length = base + (uint32_t(rank) < rest ? 1u : 0u);
}

}  // namespace

struct ZygpuGroup {
    std::vector<Member> members;
    int                 first_rank = 0, world = 1;
    bool                shared_device = false;  // a device is listed twice: members run one after the other, films are added by a kernel
    bool                rccl = false;
    bool                dirty = false;  // films other than rank 0's may hold samples
    bool                has_view = false;
    bool                rows_only = false;  // the view records AOVs: a pixel's samples stay on one device
    std::string         error;

    // fn(member) on every member's thread at once (or one after the other where members share a device); the first failure's text
    int each(const std::function<int(Member&)>& fn) {
        int rc = 0;
        if (shared_device) {
            // one after the other, each to the end of its device work: the next member's launches rewrite the device's __constant__ blocks
            for (Member& m : members) {
                m.worker->start([&fn, &m] {
                    const int r = fn(m);
                    return (0 == r && m.ctx) ? zygpu_synchronize(m.ctx) : r;
                });
                std::string e;
                const int   r = m.worker->wait(e);
                if (0 != r && 0 == rc) rc = r, error = e;
            }
        } else {
            for (Member& m : members) m.worker->start([&fn, &m] { return fn(m); });
            for (Member& m : members) {
                std::string e;
                const int   r = m.worker->wait(e);
                if (0 != r && 0 == rc) rc = r, error = e;
            }
        }
        if (0 != rc) g_group_error = error;
        return rc;
    }
};

extern "C" {

int zygpu_group_unique_id(uint8_t* out128) {
    if (!out128) return -1;
    ncclUniqueId id;
    if (!ncclOk(ncclGetUniqueId(&id), "ncclGetUniqueId")) return done(-1);
    static_assert(sizeof(id) == ZYGPU_UNIQUE_ID_BYTES, "ncclUniqueId");
    std::memcpy(out128, &id, sizeof(id));
    return 0;
}

int zygpu_group_destroy(ZygpuGroup* g) {
    if (!g) return -1;
    for (Member& m : g->members) {
        if (m.worker && (m.ctx || m.comm)) {
            m.worker->start([&m] {
                (void)hipSetDevice(m.device);
                if (m.ctx) (void)zygpu_synchronize(m.ctx);
                if (m.comm) (void)ncclCommDestroy(m.comm);
                if (m.ctx) (void)zygpu_destroy(m.ctx);
                return 0;
            });
            std::string e;
            (void)m.worker->wait(e);
        }
        delete m.worker;
    }
    delete g;
    return 0;
}

int zygpu_group_create(const int* device_ids, int n, const uint8_t* unique_id, int first_rank, int world, uint32_t flags, ZygpuGroup** out) {
    if (!device_ids || n < 1 || !out) return -1;
    *out = nullptr;
    if (!unique_id) first_rank = 0, world = n;
    if (first_rank < 0 || first_rank + n > world) {
        g_group_error = "zygpu_group_create: ranks [first_rank, first_rank + n) do not lie in the world";
        return done(-1);
    }
    auto* g = new ZygpuGroup();
    g->first_rank = first_rank, g->world = world;
    g->members.resize(size_t(n));
    for (int i = 0; i < n; ++i) {
        Member& m = g->members[size_t(i)];
        m.device = device_ids[i], m.rank = first_rank + i;
        m.worker = new Worker();
        for (int k = 0; k < i; ++k) g->shared_device |= device_ids[k] == device_ids[i];
    }
    if (0 != g->each([](Member& m) { return zygpu_create(m.device, &m.ctx); })) {
        const std::string e = g_group_error;
        zygpu_group_destroy(g);
        g_group_error = e;
        return done(-1);
    }
    // the communicator: one RCCL rank per member.  A world of one needs none (ZYGPU_GROUP_ALWAYS_RCCL asks for it all the same: the
    // path a box with one GPU can exercise); members that share a device cannot be RCCL ranks of one communicator.
    const bool want = (world > 1 || nullptr != unique_id || 0 != (flags & ZYGPU_GROUP_ALWAYS_RCCL));
    if (want && g->shared_device && unique_id) {
        zygpu_group_destroy(g);
        g_group_error = "zygpu_group_create: a world across processes cannot list a device twice";
        return done(-1);
    }
    if (want && !g->shared_device) {
        ncclUniqueId id;
        if (unique_id) std::memcpy(&id, unique_id, sizeof(id));
        else if (!ncclOk(ncclGetUniqueId(&id), "ncclGetUniqueId")) {
            const std::string e = g_group_error;
            zygpu_group_destroy(g);
            g_group_error = e;
            return done(-1);
        }
        // every member's ncclCommInitRank from its own thread at once: the call returns when the world's ranks have all arrived
        const int rc = g->each([&id, world](Member& m) {
            if (!hipOk(hipSetDevice(m.device), "hipSetDevice")) return -1;
            return ncclOk(ncclCommInitRank(&m.comm, world, id, m.rank), "ncclCommInitRank") ? 0 : -1;
        });
        if (0 != rc) {
            const std::string e = g_group_error;
            zygpu_group_destroy(g);
            g_group_error = e;
            return done(-1);
        }
        g->rccl = true;
    }
    *out = g;
    return 0;
}

int zygpu_group_share(uint32_t count, int world, int rank, uint32_t* out_first, uint32_t* out_length) {
    if (world < 1 || rank < 0 || rank >= world || !out_first || !out_length) return -1;
    shareOf(count, world, rank, *out_first, *out_length);
    return 0;
}

int zygpu_group_size(const ZygpuGroup* g) { return g ? int(g->members.size()) : -1; }
int zygpu_group_world(const ZygpuGroup* g) { return g ? g->world : -1; }
int zygpu_group_first_rank(const ZygpuGroup* g) { return g ? g->first_rank : -1; }
int zygpu_group_uses_rccl(const ZygpuGroup* g) { return g ? (g->rccl ? 1 : 0) : -1; }

ZygpuContext* zygpu_group_context(ZygpuGroup* g, int member) {
    return (g && member >= 0 && size_t(member) < g->members.size()) ? g->members[size_t(member)].ctx : nullptr;
}

int zygpu_group_upload_scene(ZygpuGroup* g, const ZygpuSceneDesc* scene) {
    if (!g || !scene) return -1;
    return done(g->each([scene](Member& m) { return zygpu_upload_scene(m.ctx, scene); }));
}

int zygpu_group_set_view(ZygpuGroup* g, const ZygpuViewDesc* view) {
    if (!g || !view) return -1;
    g->rows_only = g->world > 1 && 0 != view->aov_slots;
    const int rc = g->each([view](Member& m) { return zygpu_set_view(m.ctx, view); });
    g->has_view = 0 == rc;
    return done(rc);
}

int zygpu_group_set_option(ZygpuGroup* g, const char* name, int value) {
    if (!g || !name) return -1;
    int rc = 0;
    for (Member& m : g->members) rc |= zygpu_set_option(m.ctx, name, value);
    return 0 == rc ? 0 : -1;
}

// buffer.clear(weight) of the frame's film: the weight goes to rank 0's film, the others start from zero (the films add up)
int zygpu_group_clear_film(ZygpuGroup* g, float weight) {
    if (!g) return -1;
    g->dirty = false;
    return done(g->each([weight](Member& m) { return zygpu_clear_film(m.ctx, 0 == m.rank ? weight : 0.f); }));
}

int zygpu_group_render(ZygpuGroup* g, uint32_t frame, uint32_t iteration, uint32_t num_iterations, int32_t y_begin, int32_t y_end, uint32_t shard) {
    if (!g || shard > ZYGPU_SHARD_ROWS) return -1;
    if (g->rows_only) shard = ZYGPU_SHARD_ROWS;
    const int world = g->world;
    g->dirty = g->dirty || world > 1;
    return done(g->each([=](Member& m) {
        if (ZYGPU_SHARD_ITERATIONS == shard) {
            uint32_t first, count;
            shareOf(num_iterations, world, m.rank, first, count);
            return 0 == count ? 0 : zygpu_render(m.ctx, frame, iteration + first, count, y_begin, y_end);
        }
        uint32_t first, count;
        shareOf(uint32_t(std::max(y_end - y_begin, 0)), world, m.rank, first, count);
        return 0 == count ? 0 : zygpu_render(m.ctx, frame, iteration, num_iterations, y_begin + int32_t(first), y_begin + int32_t(first + count));
    }));
}

// One-shot rendering: a pixel's PCG stream runs on across its samples (worker.zig:141-165), so the samples of a pixel stay on one
// device and the rows are what is split
int zygpu_group_render_frame(ZygpuGroup* g, uint32_t frame, uint32_t iteration, uint32_t num_samples, int32_t y_begin, int32_t y_end) {
    if (!g) return -1;
    const int world = g->world;
    g->dirty = g->dirty || world > 1;
    return done(g->each([=](Member& m) {
        uint32_t first, count;
        shareOf(uint32_t(std::max(y_end - y_begin, 0)), world, m.rank, first, count);
        return 0 == count ? 0 : zygpu_render_frame(m.ctx, frame, iteration, num_samples, y_begin + int32_t(first), y_begin + int32_t(first + count));
    }));
}

int zygpu_group_synchronize(ZygpuGroup* g) {
    if (!g) return -1;
    return done(g->each([](Member& m) { return zygpu_synchronize(m.ctx); }));
}

// The exchange step: the world's films summed into rank 0's, the others cleared (so that a later reduce adds only what was rendered
// since).  RCCL: one ncclReduce per rank on the rank's stream, behind its kernels; the host does not wait (zygpu_group_synchronize
// does).  Every process of the world must call it.
int zygpu_group_reduce(ZygpuGroup* g) {
    if (!g || !g->has_view) return -1;
    if (1 == g->world && !g->rccl) return 0;
    int rc = 0;
    if (g->rccl) {
        rc = g->each([](Member& m) {
            void *film = nullptr, *alpha = nullptr, *stream = nullptr;
            uint64_t nf = 0, na = 0;
            if (0 != zygpu_film_device_ptr(m.ctx, &film, &nf) || 0 != zygpu_film_alpha_device_ptr(m.ctx, &alpha, &na) || 0 != zygpu_stream(m.ctx, &stream)) return -1;
            if (!hipOk(hipSetDevice(m.device), "hipSetDevice")) return -1;
            hipStream_t s = static_cast<hipStream_t>(stream);
            if (!ncclOk(ncclReduce(film, film, size_t(nf), ncclFloat, ncclSum, 0, m.comm, s), "ncclReduce(film)")) return -1;
            if (alpha && !ncclOk(ncclReduce(alpha, alpha, size_t(na), ncclFloat, ncclSum, 0, m.comm, s), "ncclReduce(alpha)")) return -1;
            if (0 != m.rank) {
                if (!hipOk(hipMemsetAsync(film, 0, size_t(nf) * sizeof(float), s), "hipMemsetAsync")) return -1;
                if (alpha && !hipOk(hipMemsetAsync(alpha, 0, size_t(na) * sizeof(float), s), "hipMemsetAsync")) return -1;
            }
            for (uint32_t k = 0; k < ZYGPU_AOV_NUM_CLASSES; ++k) {  // the AOV planes: sums, Depth a minimum (the rows were split)
                void*    plane = nullptr;
                uint64_t np = 0;
                if (0 != zygpu_aov_device_ptr(m.ctx, k, &plane, &np)) return -1;
                if (!plane) continue;
                if (!ncclOk(ncclReduce(plane, plane, size_t(np), ncclFloat, ZYGPU_AOV_DEPTH == k ? ncclMin : ncclSum, 0, m.comm, s), "ncclReduce(aov)")) return -1;
                if (0 != m.rank && 0 != zygpu_clear_aov(m.ctx, k)) return -1;
            }
            return 0;
        });
    } else {
        // members on one device (the single-GPU test set-up): rank 0's stream adds the others' films once they are complete
        Member& root = g->members[0];
        rc = g->each([](Member& m) { return zygpu_synchronize(m.ctx); });
        if (0 == rc) {
            void *film0 = nullptr, *alpha0 = nullptr, *stream0 = nullptr;
            uint64_t nf = 0, na = 0;
            if (0 != zygpu_film_device_ptr(root.ctx, &film0, &nf) || 0 != zygpu_film_alpha_device_ptr(root.ctx, &alpha0, &na) || 0 != zygpu_stream(root.ctx, &stream0)) return -1;
            if (!hipOk(hipSetDevice(root.device), "hipSetDevice")) return -1;
            hipStream_t  s = static_cast<hipStream_t>(stream0);
            const size_t pixels = size_t(nf / 4);
            for (size_t i = 1; i < g->members.size() && 0 == rc; ++i) {
                void *film = nullptr, *alpha = nullptr;
                if (0 != zygpu_film_device_ptr(g->members[i].ctx, &film, nullptr) || 0 != zygpu_film_alpha_device_ptr(g->members[i].ctx, &alpha, nullptr)) return -1;
                hipLaunchKernelGGL(film_add, dim3(unsigned((pixels + 255) / 256)), dim3(256), 0, s, static_cast<float4*>(film0), static_cast<const float4*>(film), pixels);
                if (alpha0 && alpha) hipLaunchKernelGGL(plane_add, dim3(unsigned((na + 255) / 256)), dim3(256), 0, s, static_cast<float*>(alpha0), static_cast<const float*>(alpha), size_t(na));
                if (!hipOk(hipMemsetAsync(film, 0, size_t(nf) * sizeof(float), s), "hipMemsetAsync")) rc = -1;
                if (alpha && !hipOk(hipMemsetAsync(alpha, 0, size_t(na) * sizeof(float), s), "hipMemsetAsync")) rc = -1;
                for (uint32_t k = 0; k < ZYGPU_AOV_NUM_CLASSES && 0 == rc; ++k) {
                    void *plane0 = nullptr, *plane = nullptr;
                    if (0 != zygpu_aov_device_ptr(root.ctx, k, &plane0, nullptr) || 0 != zygpu_aov_device_ptr(g->members[i].ctx, k, &plane, nullptr)) return -1;
                    if (!plane0 || !plane) continue;
                    if (ZYGPU_AOV_DEPTH == k) hipLaunchKernelGGL(film_min, dim3(unsigned((pixels + 255) / 256)), dim3(256), 0, s, static_cast<float4*>(plane0), static_cast<const float4*>(plane), pixels);
                    else hipLaunchKernelGGL(film_add, dim3(unsigned((pixels + 255) / 256)), dim3(256), 0, s, static_cast<float4*>(plane0), static_cast<const float4*>(plane), pixels);
                }
            }
            // (the members' planes are cleared on their own streams once the root's stream has read them)
            if (0 == rc && !hipOk(hipStreamSynchronize(s), "film_add")) rc = -1;
            for (size_t i = 1; i < g->members.size() && 0 == rc; ++i)
                for (uint32_t k = 0; k < ZYGPU_AOV_NUM_CLASSES; ++k) {
                    const int r = zygpu_clear_aov(g->members[i].ctx, k);
                    if (0 != r && -2 != r) rc = -1;
                }
            for (size_t i = 1; i < g->members.size() && 0 == rc; ++i) rc = zygpu_synchronize(g->members[i].ctx);
        }
    }
    if (0 == rc) g->dirty = false;
    return done(rc);
}

int zygpu_group_dirty(const ZygpuGroup* g) { return g ? (g->dirty ? 1 : 0) : -1; }

}  // extern "C"
