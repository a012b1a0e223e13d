// The one data-path exchange of the multi-GPU frame (camera.rs:113-123 is its CPU form: the row bands travel
// through a channel and are stitched by the caller): every rank's tile-major rows go to the root device, over xGMI, through
// RCCL.  One process drives all devices (rt_render_multi, abi.cpp): the communicators come from ncclCommInitAll and live in
// a per-process cache keyed by the device list (creating them costs far more than the exchange); a row travels as one grouped
// ncclSend / ncclRecv pair, posted the moment its rank has finished rendering (round 5; round 4 posted all rows in one group after
// the ranks had joined) -- each peer's row crosses its own direct link to the root, 7 links in parallel on an 8-GPU node --
// on one non-blocking stream per device.  Rows that already sit on the root device never get here unless a test asks for it
// (rt_tuning.multi_force_rccl): then they are sent from a rank to itself through the same calls.
// Host code only (no kernels); compiled by hipcc for the HIP and RCCL headers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "device.h"

namespace rtamd {

#define EX_HIP(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            throw RtError((_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? RT_ERR_NO_DEVICE : RT_ERR_HIP, \
                          std::string(#expr) + ": " + hipGetErrorString(_e));                                     \
    } while (0)
#define EX_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess) throw RtError(RT_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
    } while (0)

struct Exchange {
    std::vector<int> devices;  // comm rank r lives on devices[r]
    std::vector<ncclComm_t> comms;
    std::vector<hipStream_t> streams;
    std::vector<char> used;  // streams with rows in flight (exchange_post .. exchange_wait)
    bool busy = false;
};
static std::mutex g_ex_mu;
static std::vector<Exchange*> g_ex;  // never destroyed at exit (the HIP runtime may be gone by then); rt_release_workspaces frees idle ones

struct DeviceGuard {  // the calls below move the thread's current device; put it back
    int cur = -1;
    DeviceGuard() {
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
    }
    ~DeviceGuard() {
        if (cur >= 0) (void)hipSetDevice(cur);
    }
};

static void destroy(Exchange* e) {
    for (size_t r = 0; r < e->devices.size(); r++) {
        if (r < e->streams.size() && e->streams[r]) {
            (void)hipSetDevice(e->devices[r]);
            (void)hipStreamDestroy(e->streams[r]);
        }
        if (r < e->comms.size() && e->comms[r]) (void)ncclCommDestroy(e->comms[r]);
    }
    delete e;
}

Exchange* exchange_open(const std::vector<int>& devices, bool* created) {
    if (devices.empty()) throw RtError(RT_ERR_ARG, "exchange over no device");
    if (created) *created = false;
    {
        std::lock_guard<std::mutex> g(g_ex_mu);
        for (Exchange* e : g_ex)
            if (!e->busy && e->devices == devices) {
                e->busy = true;
                return e;
            }
    }
    if (created) *created = true;
    DeviceGuard guard;
    Exchange* e = new Exchange();
    e->devices = devices;
    e->comms.assign(devices.size(), nullptr);
    e->streams.assign(devices.size(), nullptr);
    e->used.assign(devices.size(), 0);
    try {
        EX_NCCL(ncclCommInitAll(e->comms.data(), (int)devices.size(), devices.data()));
        for (size_t r = 0; r < devices.size(); r++) {
            EX_HIP(hipSetDevice(devices[r]));
            EX_HIP(hipStreamCreateWithFlags(&e->streams[r], hipStreamNonBlocking));
        }
    } catch (...) {
        destroy(e);
        throw;
    }
    e->busy = true;
    std::lock_guard<std::mutex> g(g_ex_mu);
    g_ex.push_back(e);
    return e;
}

void exchange_close(Exchange* e, bool failed) {
    if (!e) return;
    if (failed) {  // a send / receive / group failed on these communicators: RCCL leaves them in an undefined state -- never hand them out again
        DeviceGuard guard;
        {
            std::lock_guard<std::mutex> g(g_ex_mu);
            for (auto it = g_ex.begin(); it != g_ex.end(); ++it)
                if (*it == e) {
                    g_ex.erase(it);
                    break;
                }
        }
        for (size_t r = 0; r < e->devices.size(); r++)  // (what is still queued on the exchange streams must be over before they go)
            if (e->streams[r] && hipSetDevice(e->devices[r]) == hipSuccess) (void)hipStreamSynchronize(e->streams[r]);
        destroy(e);
        return;
    }
    std::lock_guard<std::mutex> g(g_ex_mu);
    e->busy = false;
}

size_t exchange_release_idle() {
    DeviceGuard guard;
    std::lock_guard<std::mutex> g(g_ex_mu);
    size_t n = 0;
    for (auto it = g_ex.begin(); it != g_ex.end();) {
        if ((*it)->busy) {
            ++it;
            continue;
        }
        destroy(*it);
        it = g_ex.erase(it);
        n++;
    }
    return n;
}

// One row as one group: an ncclSend on the source rank's communicator and the matching ncclRecv on the destination's (rows between the
// same pair are matched in issue order), enqueued on the two ranks' exchange streams.  The rank threads of rt_render_multi call this one
// at a time (a communicator may be driven by several threads, not concurrently), each the moment its render is over: a peer's row then
// crosses its xGMI link while the other ranks still render; the root's receive kernels run as soon as its own render leaves them a CU.
void exchange_post(Exchange* e, const RowMove& m) {
    if (!e) throw RtError(RT_ERR_ARG, "exchange_post: no exchange");
    const int n = (int)e->devices.size();
    if (m.src_rank < 0 || m.src_rank >= n || m.dst_rank < 0 || m.dst_rank >= n || !m.src || !m.dst) throw RtError(RT_ERR_ARG, "exchange_post: bad move");
    if (m.count == 0) return;
    DeviceGuard guard;
    e->used[(size_t)m.src_rank] = e->used[(size_t)m.dst_rank] = 1;
    EX_NCCL(ncclGroupStart());
    ncclResult_t first = ncclSend(m.src, m.count, ncclDouble, m.dst_rank, e->comms[(size_t)m.src_rank], e->streams[(size_t)m.src_rank]);
    if (first == ncclSuccess) first = ncclRecv(m.dst, m.count, ncclDouble, m.src_rank, e->comms[(size_t)m.dst_rank], e->streams[(size_t)m.dst_rank]);
    const ncclResult_t end = ncclGroupEnd();  // always closed, also after a failed call inside the group
    if (first != ncclSuccess) throw RtError(RT_ERR_HIP, std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(first));
    if (end != ncclSuccess) throw RtError(RT_ERR_HIP, std::string("ncclGroupEnd: ") + ncclGetErrorString(end));
}
void exchange_wait(Exchange* e) {
    if (!e) return;
    DeviceGuard guard;
    for (size_t r = 0; r < e->devices.size(); r++)
        if (e->used[r]) {
            e->used[r] = 0;
            EX_HIP(hipSetDevice(e->devices[r]));
            EX_HIP(hipStreamSynchronize(e->streams[r]));
        }
}

int exchange_library_version() {
    int v = 0;
    if (ncclGetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int dev_get_device() {
    int d = 0;
    EX_HIP(hipGetDevice(&d));
    return d;
}
void dev_synchronize() { EX_HIP(hipDeviceSynchronize()); }

}  // namespace rtamd
