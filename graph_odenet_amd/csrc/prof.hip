// prof.hip — measurement aid: HIP-event brackets around the dominant kernel (the SpMM main
// kernel), recorded on the stream the kernel is launched on.  bench.py enables it over its timed
// region to obtain the live average launch duration that the roofline figure is computed from.
// Disabled (one pointer test per launch) unless gode_prof_enable() was called.
#include "common.h"
#include "prof.h"
#include <vector>

struct GodeProf {
    int capacity;
    int count;
    std::vector<hipEvent_t> ev;      // 2*capacity
    std::vector<int64_t> d, rows, extra;
    std::vector<int32_t> kind;
};

// process-global on purpose: torch's autograd engine runs backward on its own thread
static GodeProf* volatile g_prof = nullptr;

extern "C" void* gode_prof_create(int capacity) {
    if (capacity <= 0) return nullptr;
    GodeProf* p = new GodeProf();
    p->capacity = capacity; p->count = 0;
    p->ev.resize(2 * (size_t)capacity); p->d.resize(capacity); p->rows.resize(capacity); p->extra.resize(capacity); p->kind.resize(capacity);
    for (auto& e : p->ev) if (hipEventCreate(&e) != hipSuccess) { delete p; return nullptr; }
    return p;
}
extern "C" void gode_prof_destroy(void* prof) {
    GodeProf* p = (GodeProf*)prof;
    if (!p) return;
    if (g_prof == p) g_prof = nullptr;
    for (auto& e : p->ev) (void)hipEventDestroy(e);
    delete p;
}
extern "C" void gode_prof_enable(void* prof) { g_prof = (GodeProf*)prof; }
extern "C" void gode_prof_reset(void* prof) { if (prof) ((GodeProf*)prof)->count = 0; }
extern "C" int gode_prof_count(void* prof) { return prof ? ((GodeProf*)prof)->count : 0; }

extern "C" int gode_prof_read(void* prof, float* ms, int64_t* d, int64_t* rows, int64_t* extra, int max_n) {
    GodeProf* p = (GodeProf*)prof;
    if (!p || !ms) return GODE_E_NULLPTR;
    int n = p->count < max_n ? p->count : max_n;
    for (int i = 0; i < n; ++i) {
        hipError_t e = hipEventSynchronize(p->ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        e = hipEventElapsedTime(&ms[i], p->ev[2 * i], p->ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        if (d) d[i] = p->d[i];
        if (rows) rows[i] = p->rows[i];
        if (extra) extra[i] = p->extra[i];
    }
    return n;
}

extern "C" int gode_prof_kinds(void* prof, int32_t* kinds, int max_n) {
    GodeProf* p = (GodeProf*)prof;
    if (!p || !kinds) return GODE_E_NULLPTR;
    const int n = p->count < max_n ? p->count : max_n;
    for (int i = 0; i < n; ++i) kinds[i] = p->kind[i];
    return n;
}

int gode_prof_begin(hipStream_t s, int64_t d, int64_t rows, int64_t extra, int kind) {
    GodeProf* p = g_prof;
    if (!p || p->count >= p->capacity) return -1;
    // never inside a stream capture: the record would become a graph node that refers to an event this profile owns -
    // a replay after gode_prof_destroy() then touches a destroyed handle (hipErrorInvalidHandle on the next API call;
    // round 4: bench.py --gpus 2 at a launch-bound size, where the solves are captured inside the timed region)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return -1;
    const int i = p->count;
    p->d[i] = d; p->rows[i] = rows; p->extra[i] = extra; p->kind[i] = kind;
    (void)hipEventRecord(p->ev[2 * i], s);
    return i;
}
void gode_prof_end(hipStream_t s, int slot) {
    GodeProf* p = g_prof;
    if (!p || slot < 0) return;
    (void)hipEventRecord(p->ev[2 * slot + 1], s);
    p->count = slot + 1;
}
