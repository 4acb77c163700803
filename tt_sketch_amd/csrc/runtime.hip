// Runtime: device selection, streams, memory, timers, hipGraph capture.
#include <cstring>
#include <mutex>
#include "common.h"

namespace ttsk {

static thread_local char g_err[512] = "";
static hipStream_t g_streams[TTSK_NUM_STREAMS];
static hipEvent_t g_ev_start[TTSK_NUM_STREAMS], g_ev_stop[TTSK_NUM_STREAMS];
static hipEvent_t g_ev_sync[TTSK_NUM_STREAMS];
static bool g_init = false;
static int g_device = -1;
static std::mutex g_mu;

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ensure_init()
{
    if (g_init) return TTSK_OK;
    return ttsk_init(0);
}

// Grow-only scratch arena per library stream.  (hipMallocAsync's stream-ordered pool returned
// buffers that were corrupted under us on ROCm 7.2 / gfx950 -- measured with the QR kernels --
// so temporaries come from plain hipMalloc and are reused in stream order.)
static void *g_scratch[TTSK_NUM_STREAMS][SCRATCH_SLOTS];
static size_t g_scratch_bytes[TTSK_NUM_STREAMS][SCRATCH_SLOTS];

void *scratch(int s, int slot, size_t bytes)
{
    if (ensure_init() != TTSK_OK || s < 0 || s >= TTSK_NUM_STREAMS || slot < 0 || slot >= SCRATCH_SLOTS)
        return nullptr;
    if (bytes <= g_scratch_bytes[s][slot]) return g_scratch[s][slot];
    if (g_scratch[s][slot]) {
        (void)hipStreamSynchronize(g_streams[s]);
        (void)hipFree(g_scratch[s][slot]);
        g_scratch[s][slot] = nullptr;
        g_scratch_bytes[s][slot] = 0;
    }
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&g_scratch[s][slot], want);
    if (e != hipSuccess) {
        set_error("scratch allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        g_scratch[s][slot] = nullptr;
        return nullptr;
    }
    g_scratch_bytes[s][slot] = want;
    return g_scratch[s][slot];
}

static void *g_pa[PA_SLOTS];
static bool g_pa_host[PA_SLOTS];
static int g_generation = 0;

int init_generation() { return g_generation; }

void *persistent_alloc(int key, size_t bytes, bool host, bool zero)
{
    if (ensure_init() != TTSK_OK || key < 0 || key >= PA_SLOTS) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_pa[key]) return g_pa[key];
    void *q = nullptr;
    if ((host ? hipHostMalloc(&q, bytes, hipHostMallocDefault) : hipMalloc(&q, bytes)) != hipSuccess) return nullptr;
    if (zero) {
        if (host) memset(q, 0, bytes);
        else if (hipMemset(q, 0, bytes) != hipSuccess) { (void)hipFree(q); return nullptr; }
    }
    g_pa[key] = q;
    g_pa_host[key] = host;
    return q;
}

hipStream_t stream_of(int s)
{
    if (ensure_init() != TTSK_OK) return nullptr;
    if (s < 0 || s >= TTSK_NUM_STREAMS) {
        set_error("invalid stream index %d", s);
        return nullptr;
    }
    return g_streams[s];
}

}  // namespace ttsk

using namespace ttsk;

extern "C" {

const char *ttsk_last_error(void) { return g_err; }

int ttsk_init(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_init) {
        TTSK_ARG(device == g_device, "ttsk_init: already initialised on device %d", g_device);
        return TTSK_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        set_error("no HIP device available (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
        return TTSK_ERR_HIP;
    }
    TTSK_ARG(device >= 0 && device < n, "device %d out of range (%d devices)", device, n);
    TTSK_HIP(hipSetDevice(device));
    for (int i = 0; i < TTSK_NUM_STREAMS; ++i) {
        TTSK_HIP(hipStreamCreateWithFlags(&g_streams[i], hipStreamNonBlocking));
        TTSK_HIP(hipEventCreate(&g_ev_start[i]));
        TTSK_HIP(hipEventCreate(&g_ev_stop[i]));
        TTSK_HIP(hipEventCreateWithFlags(&g_ev_sync[i], hipEventDisableTiming));
    }
    g_device = device;
    g_init = true;
    return TTSK_OK;
}

int ttsk_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init) return TTSK_OK;
    for (int i = 0; i < TTSK_NUM_STREAMS; ++i) {
        (void)hipStreamSynchronize(g_streams[i]);
        (void)hipStreamDestroy(g_streams[i]);
        (void)hipEventDestroy(g_ev_start[i]);
        (void)hipEventDestroy(g_ev_stop[i]);
        (void)hipEventDestroy(g_ev_sync[i]);
        for (int k = 0; k < SCRATCH_SLOTS; ++k) {
            if (g_scratch[i][k]) (void)hipFree(g_scratch[i][k]);
            g_scratch[i][k] = nullptr;
            g_scratch_bytes[i][k] = 0;
        }
    }
    for (int k = 0; k < PA_SLOTS; ++k) {
        if (g_pa[k]) (void)(g_pa_host[k] ? hipHostFree(g_pa[k]) : hipFree(g_pa[k]));
        g_pa[k] = nullptr;
    }
    ++g_generation;
    g_init = false;
    g_device = -1;
    return TTSK_OK;
}

int ttsk_device_info(char *name, size_t name_len, int *num_cu, size_t *hbm_bytes)
{
    if (ensure_init() != TTSK_OK) return TTSK_ERR_HIP;
    hipDeviceProp_t p;
    TTSK_HIP(hipGetDeviceProperties(&p, g_device));
    if (name && name_len) {
        snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (num_cu) *num_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return TTSK_OK;
}

int ttsk_malloc(void **dev, size_t bytes)
{
    TTSK_ARG(dev != nullptr, "ttsk_malloc: NULL out pointer");
    if (ensure_init() != TTSK_OK) return TTSK_ERR_HIP;
    *dev = nullptr;
    if (bytes == 0) bytes = 8;
    TTSK_HIP(hipMalloc(dev, bytes));
    return TTSK_OK;
}

int ttsk_free(void *dev)
{
    if (!dev) return TTSK_OK;
    TTSK_HIP(hipFree(dev));
    return TTSK_OK;
}

int ttsk_memset(void *dev, int value, size_t bytes, int stream)
{
    TTSK_STREAM(st, stream);
    if (bytes == 0) return TTSK_OK;
    TTSK_HIP(hipMemsetAsync(dev, value, bytes, st));
    return TTSK_OK;
}

int ttsk_h2d(void *dev, const void *host, size_t bytes, int stream)
{
    TTSK_STREAM(st, stream);
    if (bytes == 0) return TTSK_OK;
    TTSK_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
    TTSK_HIP(hipStreamSynchronize(st));
    return TTSK_OK;
}

int ttsk_d2h(void *host, const void *dev, size_t bytes, int stream)
{
    TTSK_STREAM(st, stream);
    if (bytes == 0) return TTSK_OK;
    TTSK_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
    TTSK_HIP(hipStreamSynchronize(st));
    return TTSK_OK;
}

int ttsk_d2d(void *dst, const void *src, size_t bytes, int stream)
{
    TTSK_STREAM(st, stream);
    if (bytes == 0) return TTSK_OK;
    TTSK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    return TTSK_OK;
}

int ttsk_sync(int stream)
{
    if (ensure_init() != TTSK_OK) return TTSK_ERR_HIP;
    if (stream < 0) {
        for (int i = 0; i < TTSK_NUM_STREAMS; ++i) TTSK_HIP(hipStreamSynchronize(g_streams[i]));
        return TTSK_OK;
    }
    TTSK_STREAM(st, stream);
    TTSK_HIP(hipStreamSynchronize(st));
    return TTSK_OK;
}

int ttsk_stream_wait(int waiter, int signaller)
{
    TTSK_STREAM(sw, waiter);
    TTSK_STREAM(ss, signaller);
    if (waiter == signaller) return TTSK_OK;
    TTSK_HIP(hipEventRecord(g_ev_sync[signaller], ss));
    TTSK_HIP(hipStreamWaitEvent(sw, g_ev_sync[signaller], 0));
    return TTSK_OK;
}

int ttsk_timer_start(int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_HIP(hipEventRecord(g_ev_start[stream], st));
    return TTSK_OK;
}

int ttsk_timer_stop(int stream, float *ms)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(ms != nullptr, "ttsk_timer_stop: NULL out");
    TTSK_HIP(hipEventRecord(g_ev_stop[stream], st));
    TTSK_HIP(hipEventSynchronize(g_ev_stop[stream]));
    TTSK_HIP(hipEventElapsedTime(ms, g_ev_start[stream], g_ev_stop[stream]));
    return TTSK_OK;
}

int ttsk_graph_begin(int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    return TTSK_OK;
}

int ttsk_graph_end(int stream, void **graph_exec)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(graph_exec != nullptr, "ttsk_graph_end: NULL out");
    hipGraph_t g = nullptr;
    TTSK_HIP(hipStreamEndCapture(st, &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return TTSK_ERR_HIP;
    }
    *graph_exec = (void *)ge;
    return TTSK_OK;
}

int ttsk_graph_launch(void *graph_exec, int stream)
{
    TTSK_STREAM(st, stream);
    TTSK_ARG(graph_exec != nullptr, "ttsk_graph_launch: NULL graph");
    TTSK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, st));
    return TTSK_OK;
}

int ttsk_graph_free(void *graph_exec)
{
    if (!graph_exec) return TTSK_OK;
    TTSK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return TTSK_OK;
}

}  // extern "C"
