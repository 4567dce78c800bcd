// Several GPUs of one node driven by one process through the C ABI (include/rt_api.h, rt_multi_*): one rt_ctx per device,
// each rendering its interleaved 32x32 tiles (rt_set_shard), the tile blocks gathered to the first device and re-ordered
// there (rt_assemble_tiles).  The gather is the path's only exchange step: a 7-to-1 incast over xGMI, issued as one
// RCCL group of ncclRecv on the first device and one ncclSend per peer, each on its context's render stream so that it
// starts when that device's tiles are finished.  No reduction, no other collective.
//
// RCCL is loaded at run time (dlopen of librccl.so.1 and the handful of entry points used): the library has no link-time
// dependency on it, single-GPU callers never load it, and inside a process that already holds an RCCL (PyTorch) the
// same copy is used.  When a device appears more than once in the list (rehearsal of the sharded path on a one-GPU box)
// RCCL cannot form a communicator; the blocks are then moved with hipMemcpyAsync on the same streams.
#include "rt_api.h"
#include "rt_types.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

// the slice of rccl.h this file needs (rccl/rccl.h:236-933 of ROCm 7.2; ncclUint8 = 1, ncclSuccess = 0)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (so) return true;
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { so = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
        if (!so) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define RT_SYM(field, sym) do { *(void**)(&field) = dlsym(so, sym); if (!field) { err = std::string("librccl lacks ") + sym; return false; } } while (0)
        RT_SYM(CommInitAll, "ncclCommInitAll"); RT_SYM(CommDestroy, "ncclCommDestroy"); RT_SYM(GroupStart, "ncclGroupStart");
        RT_SYM(GroupEnd, "ncclGroupEnd"); RT_SYM(Send, "ncclSend"); RT_SYM(Recv, "ncclRecv"); RT_SYM(Broadcast, "ncclBroadcast"); RT_SYM(GetErrorString, "ncclGetErrorString");
#undef RT_SYM
        return true;
    }
};
const int kNcclUint8 = 1;

std::string g_multi_create_error;

} // namespace

struct rt_multi {
    int n = 0;
    std::vector<int> devices;
    std::vector<rt_ctx*> ctx;
    std::vector<unsigned char*> block;       // per device: this rank's tile block (device memory)
    unsigned char* gathered = nullptr;       // first device: n blocks, rank-major
    unsigned char* image = nullptr;          // first device: row-major frame
    size_t block_bytes = 0, image_bytes = 0;
    bool use_rccl = false;
    bool device_broadcast = true;            // rt_multi_upload: one upload + build, then the device arrays go to the peers (option "device_broadcast")
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    std::vector<hipEvent_t> done;            // per device: block rendered (memcpy transport only)
    std::vector<hipEvent_t> copied;          // per device: block copied to the first device (memcpy transport only)
    std::vector<char> copy_pending;          // ... and whether such a copy has been enqueued since the device's stream last waited for it
    std::string err;
    int width = 0, height = 0;
    bool uploaded = false;

    int fail(int code, const std::string& m) { err = m; return code; }
    int hip(hipError_t e, const char* what) { return e == hipSuccess ? RT_OK : fail(RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
    int nccl(ncclResult_t r, const char* what) { return r == 0 ? RT_OK : fail(RT_ERR_HIP, std::string(what) + ": " + rccl.GetErrorString(r)); }
    void free_buffers() {
        for (int i = 0; i < n; ++i) if (block[(size_t)i]) { (void)hipSetDevice(devices[(size_t)i]); (void)hipFree(block[(size_t)i]); block[(size_t)i] = nullptr; }
        if (gathered || image) (void)hipSetDevice(devices[0]);
        if (gathered) { (void)hipFree(gathered); gathered = nullptr; }
        if (image) { (void)hipFree(image); image = nullptr; }
    }
};

#define RTM_TRY(expr) do { int rc__ = (expr); if (rc__ != RT_OK) return rc__; } while (0)

extern "C" {

rt_multi* rt_multi_create(int32_t n_devices, const int32_t* device_ids, uint64_t workspace_bytes) {
    if (n_devices < 1 || n_devices > 64 || !device_ids) { g_multi_create_error = "rt_multi_create: need 1..64 device ids"; return nullptr; }
    rt_multi* m = new rt_multi();
    m->n = n_devices;
    m->devices.assign(device_ids, device_ids + n_devices);
    m->ctx.assign((size_t)n_devices, nullptr); m->block.assign((size_t)n_devices, nullptr); m->done.assign((size_t)n_devices, nullptr);
    m->copied.assign((size_t)n_devices, nullptr); m->copy_pending.assign((size_t)n_devices, 0);
    std::vector<int> sorted = m->devices; std::sort(sorted.begin(), sorted.end());
    const bool distinct = std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end();
    for (int i = 0; i < n_devices; ++i) {
        m->ctx[(size_t)i] = rt_create(m->devices[(size_t)i], workspace_bytes);
        if (!m->ctx[(size_t)i]) { g_multi_create_error = std::string("rt_multi_create: device ") + std::to_string(m->devices[(size_t)i]) + ": " + rt_last_error(nullptr); rt_multi_destroy(m); return nullptr; }
        if (rt_set_shard(m->ctx[(size_t)i], i, n_devices) != RT_OK) { g_multi_create_error = rt_last_error(m->ctx[(size_t)i]); rt_multi_destroy(m); return nullptr; }
        (void)hipSetDevice(m->devices[(size_t)i]);
        if (hipEventCreateWithFlags(&m->done[(size_t)i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&m->copied[(size_t)i], hipEventDisableTiming) != hipSuccess) { g_multi_create_error = "rt_multi_create: hipEventCreate failed"; rt_multi_destroy(m); return nullptr; }
    }
    if (distinct) {
        // one communicator per device, all made by this process (single-process multi-GPU)
        if (!m->rccl.load(g_multi_create_error)) { rt_multi_destroy(m); return nullptr; }
        m->comms.assign((size_t)n_devices, nullptr);
        ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), n_devices, m->devices.data());
        if (r != 0) { g_multi_create_error = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r); m->comms.clear(); rt_multi_destroy(m); return nullptr; }
        m->use_rccl = true;
    }
    return m;
}

void rt_multi_destroy(rt_multi* m) {
    if (!m) return;
    for (int i = 0; i < m->n; ++i) if (m->ctx[(size_t)i]) (void)rt_synchronize(m->ctx[(size_t)i]);
    for (ncclComm_t c : m->comms) if (c) (void)m->rccl.CommDestroy(c);
    m->free_buffers();
    for (int i = 0; i < m->n; ++i) {
        if (m->done[(size_t)i]) { (void)hipSetDevice(m->devices[(size_t)i]); (void)hipEventDestroy(m->done[(size_t)i]); }
        if (m->copied[(size_t)i]) { (void)hipSetDevice(m->devices[(size_t)i]); (void)hipEventDestroy(m->copied[(size_t)i]); }
        if (m->ctx[(size_t)i]) rt_destroy(m->ctx[(size_t)i]);
    }
    delete m;
}

const char* rt_multi_last_error(const rt_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }
int32_t rt_multi_uses_rccl(const rt_multi* m) { return m && m->use_rccl ? 1 : 0; }
rt_ctx* rt_multi_context(rt_multi* m, int32_t rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[(size_t)rank] : nullptr; }

int rt_multi_set_option(rt_multi* m, const char* name, int64_t value) {
    if (!m || !name) return RT_ERR_ARG;
    if (std::strcmp(name, "device_broadcast") == 0) { m->device_broadcast = value != 0; return RT_OK; }
    for (int i = 0; i < m->n; ++i)
        if (rt_set_option(m->ctx[(size_t)i], name, value) != RT_OK) return m->fail(RT_ERR_ARG, rt_last_error(m->ctx[(size_t)i]));
    return RT_OK;
}

// The scene and the BVH reach every device once per frame set.  Default: the first device takes the host arrays, builds
// its second-level hierarchies and the records the kernels read; the other contexts adopt that layout (they allocate, no
// host work, no build) and every device array is then broadcast from the first device - one ncclBroadcast per array,
// grouped over the devices, each on its context's stream (RCCL over xGMI; device copies when the list repeats a device
// id).  With the option "device_broadcast" off every device uploads and builds for itself, one host thread per device.
static int multi_broadcast_scene(rt_multi* m) {
    std::vector<unsigned char> layout((size_t)rt_scene_layout_bytes());
    if (rt_export_scene_layout(m->ctx[0], layout.data(), layout.size()) != RT_OK) return m->fail(RT_ERR_STATE, rt_last_error(m->ctx[0]));
    for (int i = 1; i < m->n; ++i)
        if (rt_adopt_scene_layout(m->ctx[(size_t)i], layout.data(), layout.size()) != RT_OK)
            return m->fail(RT_ERR_HIP, std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + rt_last_error(m->ctx[(size_t)i]));
    const int n_arrays = rt_scene_array_count(m->ctx[0]);
    for (int a = 0; a < n_arrays; ++a) {
        std::vector<void*> ptr((size_t)m->n, nullptr); uint64_t bytes = 0;
        for (int i = 0; i < m->n; ++i) {
            uint64_t b = 0;
            if (rt_scene_array(m->ctx[(size_t)i], a, &ptr[(size_t)i], &b) != RT_OK) return m->fail(RT_ERR_STATE, rt_last_error(m->ctx[(size_t)i]));
            if (i == 0) bytes = b; else if (b < bytes) return m->fail(RT_ERR_STATE, "adopted scene array smaller than its source");
        }
        if (!bytes) continue;
        if (m->use_rccl) {
            RTM_TRY(m->nccl(m->rccl.GroupStart(), "ncclGroupStart"));
            ncclResult_t first_bad = 0;
            for (int i = 0; i < m->n && first_bad == 0; ++i)
                first_bad = m->rccl.Broadcast(ptr[(size_t)i], ptr[(size_t)i], (size_t)bytes, kNcclUint8, 0, m->comms[(size_t)i], (hipStream_t)rt_get_stream(m->ctx[(size_t)i]));
            const ncclResult_t end = m->rccl.GroupEnd();
            if (first_bad != 0) return m->nccl(first_bad, "ncclBroadcast");
            RTM_TRY(m->nccl(end, "ncclGroupEnd"));
        } else {
            for (int i = 1; i < m->n; ++i) {
                RTM_TRY(m->hip(hipSetDevice(m->devices[(size_t)i]), "hipSetDevice"));
                RTM_TRY(m->hip(hipMemcpyAsync(ptr[(size_t)i], ptr[0], (size_t)bytes, hipMemcpyDefault, (hipStream_t)rt_get_stream(m->ctx[(size_t)i])), "hipMemcpyAsync"));
            }
        }
    }
    for (int i = 0; i < m->n; ++i) if (rt_synchronize(m->ctx[(size_t)i]) != RT_OK) return m->fail(RT_ERR_HIP, rt_last_error(m->ctx[(size_t)i]));
    return RT_OK;
}

int rt_multi_upload(rt_multi* m, const rt_scene* scene, const rt_bvh_node* nodes, int32_t n_nodes, const int32_t* prim_ids, int32_t n_prim_ids) {
    if (!m || !scene) return RT_ERR_ARG;
    m->uploaded = false;
    std::vector<int> rc((size_t)m->n, RT_OK);
    if (m->device_broadcast && m->n > 1) {
        int r = rt_upload_scene(m->ctx[0], scene);
        if (r == RT_OK) r = rt_upload_bvh(m->ctx[0], nodes, n_nodes, prim_ids, n_prim_ids);
        if (r != RT_OK) return m->fail(r, std::string("device ") + std::to_string(m->devices[0]) + ": " + rt_last_error(m->ctx[0]));
        RTM_TRY(multi_broadcast_scene(m));
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < m->n; ++i)
            th.emplace_back([&, i]() {
                int r = rt_upload_scene(m->ctx[(size_t)i], scene);
                if (r == RT_OK) r = rt_upload_bvh(m->ctx[(size_t)i], nodes, n_nodes, prim_ids, n_prim_ids);
                rc[(size_t)i] = r;
            });
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < m->n; ++i) if (rc[(size_t)i] != RT_OK) return m->fail(rc[(size_t)i], std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + rt_last_error(m->ctx[(size_t)i]));
    m->width = (int)scene->camera.res_x; m->height = (int)scene->camera.res_y;
    const size_t bb = (size_t)rt_max_tiles_per_rank(m->ctx[0]) * RT_TILE_PIX * 3, ib = (size_t)m->width * (size_t)m->height * 3;
    if (bb != m->block_bytes || ib != m->image_bytes || !m->image) {
        for (int i = 0; i < m->n; ++i) (void)rt_synchronize(m->ctx[(size_t)i]);
        m->free_buffers();
        m->block_bytes = bb; m->image_bytes = ib;
        for (int i = 0; i < m->n; ++i) {
            RTM_TRY(m->hip(hipSetDevice(m->devices[(size_t)i]), "hipSetDevice"));
            RTM_TRY(m->hip(hipMalloc((void**)&m->block[(size_t)i], std::max<size_t>(bb, 16)), "hipMalloc(tile block)"));
        }
        RTM_TRY(m->hip(hipSetDevice(m->devices[0]), "hipSetDevice"));
        RTM_TRY(m->hip(hipMalloc((void**)&m->gathered, std::max<size_t>(bb * (size_t)m->n, 16)), "hipMalloc(gathered blocks)"));
        RTM_TRY(m->hip(hipMalloc((void**)&m->image, std::max<size_t>(ib, 16)), "hipMalloc(image)"));
    }
    m->uploaded = true;
    return RT_OK;
}

// Enqueues one frame on every device, the gather and the re-ordering; the row-major frame is complete in
// rt_multi_image() once the first context's stream has drained (rt_multi_synchronize).
int rt_multi_render_device(rt_multi* m, const rt_config* cfg) {
    if (!m || !cfg) return RT_ERR_ARG;
    if (!m->uploaded) return m->fail(RT_ERR_STATE, "rt_multi_render before rt_multi_upload");
    // memcpy transport: the previous frame's block of device i is read by a copy on the FIRST device's stream; device i's
    // stream must not start overwriting the block before that copy is done (frames may be enqueued back to back)
    if (!m->use_rccl)
        for (int i = 1; i < m->n; ++i)
            if (m->copy_pending[(size_t)i]) {
                RTM_TRY(m->hip(hipSetDevice(m->devices[(size_t)i]), "hipSetDevice"));
                RTM_TRY(m->hip(hipStreamWaitEvent((hipStream_t)rt_get_stream(m->ctx[(size_t)i]), m->copied[(size_t)i], 0), "hipStreamWaitEvent"));
                m->copy_pending[(size_t)i] = 0;
            }
    for (int i = 0; i < m->n; ++i)
        if (rt_render_device(m->ctx[(size_t)i], cfg, m->block[(size_t)i], nullptr) != RT_OK)
            return m->fail(RT_ERR_HIP, std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + rt_last_error(m->ctx[(size_t)i]));
    hipStream_t s0 = (hipStream_t)rt_get_stream(m->ctx[0]);
    const size_t bb = m->block_bytes;
    if (m->n == 1) {
        RTM_TRY(m->hip(hipSetDevice(m->devices[0]), "hipSetDevice"));
        RTM_TRY(m->hip(hipMemcpyAsync(m->image, m->block[0], m->image_bytes, hipMemcpyDeviceToDevice, s0), "hipMemcpyAsync"));   // world 1: the block is the frame
        return RT_OK;
    }
    if (m->use_rccl) {
        // a group once opened is always closed: the first failure inside it is kept and reported after ncclGroupEnd
        RTM_TRY(m->nccl(m->rccl.GroupStart(), "ncclGroupStart"));
        ncclResult_t first_bad = 0; const char* where = "";
        for (int i = 1; i < m->n && first_bad == 0; ++i) {
            first_bad = m->rccl.Recv(m->gathered + (size_t)i * bb, bb, kNcclUint8, i, m->comms[0], s0); where = "ncclRecv";
            if (first_bad == 0) { first_bad = m->rccl.Send(m->block[(size_t)i], bb, kNcclUint8, 0, m->comms[(size_t)i], (hipStream_t)rt_get_stream(m->ctx[(size_t)i])); where = "ncclSend"; }
        }
        const ncclResult_t end = m->rccl.GroupEnd();
        if (first_bad != 0) return m->nccl(first_bad, where);
        RTM_TRY(m->nccl(end, "ncclGroupEnd"));
        RTM_TRY(m->hip(hipSetDevice(m->devices[0]), "hipSetDevice"));
        RTM_TRY(m->hip(hipMemcpyAsync(m->gathered, m->block[0], bb, hipMemcpyDeviceToDevice, s0), "hipMemcpyAsync"));
    } else {
        for (int i = 1; i < m->n; ++i) {
            RTM_TRY(m->hip(hipSetDevice(m->devices[(size_t)i]), "hipSetDevice"));
            RTM_TRY(m->hip(hipEventRecord(m->done[(size_t)i], (hipStream_t)rt_get_stream(m->ctx[(size_t)i])), "hipEventRecord"));
        }
        RTM_TRY(m->hip(hipSetDevice(m->devices[0]), "hipSetDevice"));
        for (int i = 0; i < m->n; ++i) {
            if (i) RTM_TRY(m->hip(hipStreamWaitEvent(s0, m->done[(size_t)i], 0), "hipStreamWaitEvent"));
            RTM_TRY(m->hip(hipMemcpyAsync(m->gathered + (size_t)i * bb, m->block[(size_t)i], bb, hipMemcpyDeviceToDevice, s0), "hipMemcpyAsync"));
            if (i) { RTM_TRY(m->hip(hipEventRecord(m->copied[(size_t)i], s0), "hipEventRecord")); m->copy_pending[(size_t)i] = 1; }
        }
    }
    if (rt_assemble_tiles(m->ctx[0], m->gathered, m->n, m->image) != RT_OK) return m->fail(RT_ERR_HIP, rt_last_error(m->ctx[0]));
    return RT_OK;
}

int rt_multi_synchronize(rt_multi* m) {
    if (!m) return RT_ERR_ARG;
    for (int i = 0; i < m->n; ++i) if (rt_synchronize(m->ctx[(size_t)i]) != RT_OK) return m->fail(RT_ERR_HIP, rt_last_error(m->ctx[(size_t)i]));
    return RT_OK;
}

void* rt_multi_image(rt_multi* m) { return m ? m->image : nullptr; }

int rt_multi_render(rt_multi* m, const rt_config* cfg, uint8_t* rgb_out) {
    if (!m || !cfg || !rgb_out) return RT_ERR_ARG;
    RTM_TRY(rt_multi_render_device(m, cfg));
    RTM_TRY(rt_multi_synchronize(m));
    RTM_TRY(m->hip(hipSetDevice(m->devices[0]), "hipSetDevice"));
    return m->hip(hipMemcpy(rgb_out, m->image, m->image_bytes, hipMemcpyDeviceToHost), "hipMemcpy");
}

} // extern "C"
