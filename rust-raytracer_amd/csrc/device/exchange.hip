// The one data-path exchange of the multi-GPU frame (camera.rs:113-123 is its CPU form: the row bands travel
// through a channel and are stitched by the caller): every rank's tile-major rows go to the root device, over xGMI, through
// RCCL.  One process drives all devices (rt_render_multi, abi.cpp): the communicators come from ncclCommInitAll and live in
// a per-process cache keyed by the device list (creating them costs far more than the exchange); rows travel as grouped
// ncclSend / ncclRecv pairs -- each peer's row crosses its own direct link to the root, 7 links in parallel on an 8-GPU node --
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

Exchange* exchange_open(const std::vector<int>& devices) {
    if (devices.empty()) throw RtError(RT_ERR_ARG, "exchange over no device");
    {
        std::lock_guard<std::mutex> g(g_ex_mu);
        for (Exchange* e : g_ex)
            if (!e->busy && e->devices == devices) {
                e->busy = true;
                return e;
            }
    }
    DeviceGuard guard;
    Exchange* e = new Exchange();
    e->devices = devices;
    e->comms.assign(devices.size(), nullptr);
    e->streams.assign(devices.size(), nullptr);
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

void exchange_close(Exchange* e) {
    if (!e) return;
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

// All moves as ONE group: for every row an ncclSend on the source rank's communicator and the matching ncclRecv on the
// destination's (several rows between the same pair are matched in issue order).  Returns when every involved stream is idle.
void exchange_rows(Exchange* e, const RowMove* moves, size_t n_moves) {
    if (!e || n_moves == 0) return;
    const int n = (int)e->devices.size();
    for (size_t i = 0; i < n_moves; i++)
        if (moves[i].src_rank < 0 || moves[i].src_rank >= n || moves[i].dst_rank < 0 || moves[i].dst_rank >= n || !moves[i].src || !moves[i].dst)
            throw RtError(RT_ERR_ARG, "exchange_rows: bad move");
    DeviceGuard guard;
    std::vector<char> used((size_t)n, 0);
    ncclResult_t first = ncclSuccess;
    EX_NCCL(ncclGroupStart());
    for (size_t i = 0; i < n_moves && first == ncclSuccess; i++) {
        const RowMove& m = moves[i];
        used[(size_t)m.src_rank] = used[(size_t)m.dst_rank] = 1;
        first = ncclSend(m.src, m.count, ncclDouble, m.dst_rank, e->comms[(size_t)m.src_rank], e->streams[(size_t)m.src_rank]);
        if (first == ncclSuccess) first = ncclRecv(m.dst, m.count, ncclDouble, m.src_rank, e->comms[(size_t)m.dst_rank], e->streams[(size_t)m.dst_rank]);
    }
    const ncclResult_t end = ncclGroupEnd();  // always closed, also after a failed call inside the group
    if (first != ncclSuccess) throw RtError(RT_ERR_HIP, std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(first));
    if (end != ncclSuccess) throw RtError(RT_ERR_HIP, std::string("ncclGroupEnd: ") + ncclGetErrorString(end));
    for (int r = 0; r < n; r++)
        if (used[(size_t)r]) {
            EX_HIP(hipSetDevice(e->devices[(size_t)r]));
            EX_HIP(hipStreamSynchronize(e->streams[(size_t)r]));
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
