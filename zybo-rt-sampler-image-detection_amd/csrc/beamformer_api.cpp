// beamformer_api.cpp -- the C-ABI of include/beamformer_hip.h: process-global state, table upload, launch glue.
//
// Mirrors the reference's process model: one table set per process, loaded once before the frame loop
// (PC/src/main.pyx:172-181), single-threaded callers.  Everything numerical runs in das_kernels.hip; this
// file only validates, copies and enqueues.  There is no CPU compute path: when no GPU is usable every entry
// point reports an error and poisons its output with NaN.
#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/beamformer_hip.h"
#include "das_kernels.h"

namespace {

using bf::Algo;

struct Sizes {
    int n_microphones = 256, n_samples = 256, res_x = 57, res_y = 32, n_taps = 8;  // PC/src/config.json:3-11
    int dirs() const { return res_x * res_y; }
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        // slack: the scalar-table kernels fetch 16 entries (or 4 mics x 8 taps) at a time and may run past the last row
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T) + 256);
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Everything a digest (bf::launch_digest) depends on; compared field by field.
struct DigestKey {
    bool valid = false;
    int mic_chunk = 0, row_stride = 0, lead = 0, algo = 0, dpw = 0, dir_begin = 0, dir_end = 0, n_mics = 0, nf = 0;
    bool operator==(const DigestKey& o) const
    {
        return valid && o.valid && mic_chunk == o.mic_chunk && row_stride == o.row_stride && lead == o.lead && algo == o.algo && dpw == o.dpw &&
               dir_begin == o.dir_begin && dir_end == o.dir_end && n_mics == o.n_mics && nf == o.nf;
    }
};

// Page-locked host staging buffer (the host-pointer entry points copy through it: DMA straight from / to pinned memory
// instead of the runtime's own staging of pageable memory).
struct PinnedBuf {
    float* p = nullptr;
    size_t cap = 0;   // floats
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), n * sizeof(float), hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        return e;
    }
};

// One loaded coefficient set (what a load_coefficients_* call leaves behind).
struct TableSet {
    bool loaded = false;
    int entries = 0;      // n of the load call = D * M
    int max_whole = 0;
    DevBuf<int32_t> whole;
    DevBuf<float> frac;
    DevBuf<float> taps;
    // Digests of this table (shifted-copies layouts: LDS offsets per (direction, mic), see bf::launch_digest), one per launch
    // geometry that has been used -- one-frame and batched calls, the direction shards of several ranks -- so that alternating
    // callers do not rebuild (and re-synchronise) on every call, and a digest that an earlier launch on another stream may still
    // be reading is never overwritten in place: a slot is reused only after a device-wide synchronisation.
    struct DigestSlot {
        DevBuf<int32_t> buf;
        DigestKey key;            // what the digest was built for
        bool direct = false;      // ... in the [D][M] layout of the direction-outer kernel variant (table without structure)
        unsigned long long used = 0;
    };
    static constexpr int kDigestSlots = 4;
    DigestSlot digests[kDigestSlots];
    unsigned long long digest_clock = 0;
    void invalidate_digests() { for (auto& d : digests) d.key = DigestKey{}; }   // (buffers stay allocated; keys never match again)
    void drop()
    {
        loaded = false; entries = 0; max_whole = 0;
        whole.release(); frac.release(); taps.release();
        for (auto& d : digests) { d.buf.release(); d.key = DigestKey{}; d.direct = false; d.used = 0; }
    }
};

enum Slot { SLOT_PAD = 0, SLOT_LERP, SLOT_FIR, SLOT_HYBRID, SLOT_TRUNC, SLOT_COUNT };

struct State {
    std::mutex mu;
    Sizes sz;
    bool sizes_from_env_done = false;
    int device = -1;
    bool device_ready = false;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    TableSet tab[SLOT_COUNT];
    std::vector<int> pad2_host;          // load_coefficients_pad2: per-mic delays (pad_and_sum.c:153-157)
    // scratch for the host-pointer entry points
    DevBuf<float> d_frame, d_image, d_out, d_init, d_one_taps, d_one_frac;
    DevBuf<int32_t> d_mics, d_one_whole;
    DevBuf<unsigned long long> d_counter;  // digest build: direction steps that change the delay
    PinnedBuf h_frame, h_image;          // host-pointer mimo_*: pinned staging of the frame in / the image out
    DevBuf<float> fd_work;               // partial planes of the bin-reducing GEMMs (bf::fd_workspace_floats)
    DevBuf<float> fd_chol_work;          // blocks of the 129..256-mic Cholesky / inverse (bf::fd_cholesky_workspace_floats)
    DevBuf<float> fd_tw;                 // twiddles of the MFMA DFT for (N, bin_lo, n_bins) = fd_tw_key
    long long fd_tw_key = -1;
    std::vector<int> mics_host;          // what d_mics currently holds
    std::vector<float> published;        // bf_publish_frame / get_data
    std::vector<int> disabled_mics;      // get_data's dead-microphone rows
    bool disabled_default = true;
    int steer_offset = 0;                // steer(): flat table offset of the listening beam (api.c:576-581)
    std::vector<int> listen_mics;        // load_pa(): microphones of the listening beam (api.c:553-567); empty before load_miso / load_pa
    int last_variant = -1;               // bf_last_das_variant
    int debug_override = -1;             // bf_set_debug: planner A/B switches at run time (tests); -1 = $BF_DEBUG
    std::string err;
};

State& S()
{
    static State s;
    return s;
}

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    S().err = buf;
    fprintf(stderr, "[beamformer_hip] error: %s\n", buf);
}

void poison(float* out, size_t n)
{
    if (!out) return;
    for (size_t i = 0; i < n; ++i) out[i] = std::numeric_limits<float>::quiet_NaN();
}

#define HIP_OK(expr)                                                                          \
    ([&]() -> bool {                                                                          \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) { set_error("%s -> %s", #expr, hipGetErrorString(e__)); return false; } \
        return true;                                                                          \
    }())

// The host-pointer calls (one 64 KB frame in, one 40 KB map out) end in a wait for the stream.  hipStreamSynchronize may block or yield, depending on the
// scheduling policy the runtime picks for the box (CPU count, other devices): the same call measured 58 us on one box and 206 us on another.  A call this
// short is polled first -- hipStreamQuery in a spin for up to 2 ms -- and only then handed to the blocking wait.
static hipError_t sync_short(hipStream_t stream)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    return hipStreamSynchronize(stream);
}

// ---- config.json reader: "KEY": number, anywhere in the file (keys are unique in PC/src/config.json)
bool json_number(const std::string& text, const char* key, double* out)
{
    const std::string pat = std::string("\"") + key + "\"";
    size_t pos = text.find(pat);
    if (pos == std::string::npos) return false;
    pos = text.find(':', pos + pat.size());
    if (pos == std::string::npos) return false;
    const char* s = text.c_str() + pos + 1;
    char* end = nullptr;
    const double v = strtod(s, &end);
    if (end == s) return false;
    *out = v;
    return true;
}

bool apply_sizes(int mics, int n, int x, int y, int t)
{
    if (mics < 1 || n < 1 || x < 1 || y < 1 || t < 1 || (long long)x * y > (1 << 24)) {
        set_error("bf_configure: invalid sizes N_MICROPHONES=%d N_SAMPLES=%d MAX_RES_X=%d MAX_RES_Y=%d N_TAPS=%d", mics, n, x, y, t);
        return false;
    }
    if (n > 1024) { set_error("bf_configure: N_SAMPLES=%d > 1024 is not supported by the gfx950 kernels", n); return false; }
    State& s = S();
    const Sizes old = s.sz;
    s.sz.n_microphones = mics; s.sz.n_samples = n; s.sz.res_x = x; s.sz.res_y = y; s.sz.n_taps = t;
    if (old.n_samples != n || old.res_x != x || old.res_y != y || old.n_taps != t)
        for (auto& t2 : s.tab) t2.drop();
    s.published.clear();
    return true;
}

bool configure_from_json(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("cannot open config file %s", path); return false; }
    std::string text;
    char buf[4096];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
    fclose(f);
    Sizes z = S().sz;
    double v;
    if (json_number(text, "N_MICROPHONES", &v)) z.n_microphones = (int)v;
    if (json_number(text, "N_SAMPLES", &v)) z.n_samples = (int)v;
    if (json_number(text, "MAX_RES_X", &v)) z.res_x = (int)v;
    if (json_number(text, "MAX_RES_Y", &v)) z.res_y = (int)v;
    if (json_number(text, "N_TAPS", &v)) z.n_taps = (int)v;
    return apply_sizes(z.n_microphones, z.n_samples, z.res_x, z.res_y, z.n_taps);
}

void sizes_from_env_once()
{
    State& s = S();
    if (s.sizes_from_env_done) return;
    s.sizes_from_env_done = true;
    if (const char* p = getenv("BF_CONFIG")) (void)configure_from_json(p);
}

// Lazy, per-process HIP bring-up (never at library load: callers fork first).
bool ensure_device()
{
    State& s = S();
    sizes_from_env_once();
    if (s.device_ready) return true;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1) {
        set_error("no usable HIP device (%s); this library has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return false;
    }
    int dev = s.device;
    if (dev < 0) { const char* p = getenv("BF_DEVICE"); dev = p ? atoi(p) : 0; }
    if (dev >= count) { set_error("HIP device %d requested but only %d present", dev, count); return false; }
    if (!HIP_OK(hipSetDevice(dev))) return false;
    hipDeviceProp_t prop;
    if (!HIP_OK(hipGetDeviceProperties(&prop, dev))) return false;
    if (strncmp(prop.gcnArchName, "gfx9", 4) != 0) {
        set_error("device %d is %s; this build carries gfx950 code only", dev, prop.gcnArchName);
        return false;
    }
    s.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (!HIP_OK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking))) return false;
    s.device = dev;
    s.device_ready = true;
    return true;
}

template <typename T>
bool upload(DevBuf<T>& dst, const T* src, size_t n)
{
    if (!HIP_OK(dst.reserve(n))) return false;
    return HIP_OK(hipMemcpy(dst.p, src, n * sizeof(T), hipMemcpyHostToDevice));
}

bool upload_mics(const int* adaptive, int n, int* max_row)
{
    State& s = S();
    int mx = -1;
    for (int i = 0; i < n; ++i) {
        if (adaptive[i] < 0) { set_error("adaptive_array[%d] = %d is negative", i, adaptive[i]); return false; }
        mx = std::max(mx, adaptive[i]);
    }
    *max_row = mx;
    if ((int)s.mics_host.size() == n && std::equal(adaptive, adaptive + n, s.mics_host.begin()) && s.d_mics.p) return true;
    // a new adaptive array: launches already enqueued on the caller's (non-blocking) streams may still read the old copy
    if (s.d_mics.p && !HIP_OK(hipDeviceSynchronize())) return false;
    if (!upload(s.d_mics, adaptive, (size_t)n)) return false;
    s.mics_host.assign(adaptive, adaptive + n);
    return true;
}

int slot_of(int algo)
{
    switch (algo) {
        case bf::ALGO_PAD: return SLOT_PAD;
        case bf::ALGO_LERP: return SLOT_LERP;
        case bf::ALGO_HYBRID: return SLOT_HYBRID;
        case bf::ALGO_FIR_NAIVE:
        case bf::ALGO_FIR_VEC: return SLOT_FIR;
        default: return -1;
    }
}

const char* loader_name(int slot)
{
    static const char* n[] = {"load_coefficients_pad", "load_coefficients_lerp", "load_coefficients_convolve",
                              "load_coefficients_convolve_hybrid", "load_coefficients2"};
    return n[slot];
}

// Fill the launch descriptor shared by every entry point.  `n` = active mics, table must hold D*n entries.
bool describe(int algo, int slot, int n, bf::DasLaunch* L)
{
    State& s = S();
    const TableSet& t = s.tab[slot];
    if (!t.loaded) { set_error("%s has not been called", loader_name(slot)); return false; }
    const int D = s.sz.dirs();
    const bool fir = (slot == SLOT_FIR);
    const long long want = (long long)D * n * (fir ? s.sz.n_taps : 1);
    if (n < 1 || want != t.entries) {
        set_error("%s loaded %d coefficients but MAX_RES_X*MAX_RES_Y*n%s = %d*%d*%d%s = %lld", loader_name(slot), t.entries,
                  fir ? "*N_TAPS" : "", s.sz.res_x, s.sz.res_y, n, fir ? "*T" : "", want);
        return false;
    }
    L->algo = algo;
    L->tab.whole = t.whole.p; L->tab.frac = t.frac.p; L->tab.taps = t.taps.p; L->tab.max_whole = t.max_whole;
    L->n_mics = n; L->n_samples = s.sz.n_samples; L->n_taps = s.sz.n_taps; L->n_dirs = D;
    L->dir_begin = 0; L->dir_end = D; L->image_stride = D; L->image_origin = 0; L->frames = 1;
    L->mics = s.d_mics.p;
    return true;
}

bool plan_or_error(bf::DasLaunch& L, bf::DasPlan* plan)
{
    static const int layout = [] { const char* e = getenv("BF_LAYOUT"); return e ? atoi(e) : -1; }();
    L.force_layout = layout;             // A/B switch for tests and profiling (see DasPlan::layout)
    static const int debug = [] { const char* e = getenv("BF_DEBUG"); return e ? atoi(e) : 0; }();
    L.debug = S().debug_override >= 0 ? S().debug_override : debug;
    const char* why = "";
    if (bf::plan_das(L, S().n_cus, plan, &why) != 0) { set_error("unsupported shape: %s", why); return false; }
    return true;
}

// Shifted-copies layout with scalar tables: make sure the table set carries a digest built for this plan.
bool ensure_digest(TableSet& t, bf::DasLaunch& L, bf::DasPlan& plan, hipStream_t stream)
{
    S().last_variant = plan.layout == 2 ? (plan.nf == 2 ? 7 : 4) : plan.layout;   // refined below for the digest-driven kernels
    const bool plain_fir = L.algo == bf::ALGO_FIR_NAIVE || L.algo == bf::ALGO_FIR_VEC;
    if (plan.layout != 2 || (plain_fir && plan.nf != 2)) return true;   // the plain FIRs have no whole-sample table (their pair kernel wants the taps regrouped)
    // everything the digest depends on: the plan's geometry, the algorithm and (grouped layouts) the direction range
    const DigestKey key{true, plan.mic_chunk, plan.row_stride, plan.lead, L.algo, plan.dpw, L.dir_begin, L.dir_end, L.n_mics, plan.nf};
    TableSet::DigestSlot* slot = nullptr;
    for (auto& d : t.digests)
        if (d.buf.p && d.key == key) slot = &d;
    if (slot == nullptr) {
        State& s = S();
        // an unused slot, else the least recently used one -- which launches still in flight on other streams may be reading
        TableSet::DigestSlot* victim = &t.digests[0];
        for (auto& d : t.digests) {
            if (!d.buf.p || !d.key.valid) { victim = &d; break; }
            if (d.used < victim->used) victim = &d;
        }
        if (victim->buf.p && !HIP_OK(hipDeviceSynchronize())) return false;
        victim->key = DigestKey{};
        if (!HIP_OK(victim->buf.reserve(bf::digest_elements(L, plan))) || !HIP_OK(s.d_counter.reserve(1))) return false;
        victim->direct = false;
        const bool plain = L.algo == bf::ALGO_PAD || L.algo == bf::ALGO_LERP;
        const bool grouped = plain || ((L.algo == bf::ALGO_HYBRID || plain_fir) && plan.nf == 2);
        if (grouped && !HIP_OK(hipMemsetAsync(s.d_counter.p, 0, sizeof(unsigned long long), stream))) return false;
        if (!HIP_OK(bf::launch_digest(L, plan, victim->buf.p, grouped ? s.d_counter.p : nullptr, false, stream))) return false;
        // once per (table, geometry): wait, so that a later launch on ANOTHER stream cannot overtake the digest's construction
        if (!HIP_OK(hipStreamSynchronize(stream))) return false;
        if (plain && plan.waves == 16) {
            // A table without structure (more than half of the direction steps change the delay) defeats the sweep: use the
            // direction-outer kernel variant and its [D][M] digest instead.
            unsigned long long reloads = 0;
            if (!HIP_OK(hipMemcpy(&reloads, s.d_counter.p, sizeof(reloads), hipMemcpyDeviceToHost))) return false;
            const long long steps = bf::digest_shareable_steps(L, plan);
            if (steps > 0 && 2 * (long long)reloads > steps) {
                L.tab.digest_direct = true;
                if (!plan_or_error(L, &plan)) return false;     // that variant reads at every step: four shifted copies, its own chunk size
                if (!HIP_OK(bf::launch_digest(L, plan, victim->buf.p, nullptr, true, stream)) || !HIP_OK(hipStreamSynchronize(stream))) return false;
                victim->direct = true;
            }
        }
        victim->key = key;                                      // (the key is the sweep plan's: what the caller's planning yields next time)
        slot = victim;
    } else if (slot->direct) {
        L.tab.digest_direct = true;
        if (!plan_or_error(L, &plan)) return false;
    }
    slot->used = ++t.digest_clock;
    L.tab.digest_direct = slot->direct;
    L.tab.digest = slot->buf.p;
    S().last_variant = (L.algo == bf::ALGO_PAD || L.algo == bf::ALGO_LERP) ? (slot->direct ? 3 : plan.nf == 2 ? (plan.interleaved ? 8 : 5) : plan.long_rows ? 6 : 2) : plan.nf == 2 ? 7 : 4;
    return true;
}

// mimo_* with host pointers: one frame in, one image out (PC/src/algorithms/pad_and_sum.c:100-143 and twins).
void run_mimo_host(int algo, int slot, const float* signals, float* image, const int* adaptive, int n)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    const size_t D = (size_t)s.sz.dirs();
    if (!signals || !image || !adaptive) { set_error("null argument"); poison(image, image ? D : 0); return; }
    bool ok = ensure_device();
    bf::DasLaunch L{};
    bf::DasPlan plan{};
    int max_row = 0;
    ok = ok && upload_mics(adaptive, n, &max_row);
    ok = ok && describe(algo, slot, n, &L);
    if (ok) {
        const size_t rows = (size_t)max_row + 1;
        L.m_total = (int)rows;
        const size_t frame_floats = rows * s.sz.n_samples;
        ok = HIP_OK(s.d_frame.reserve(frame_floats)) && HIP_OK(s.d_image.reserve(D)) && HIP_OK(s.h_frame.reserve(frame_floats)) && HIP_OK(s.h_image.reserve(D));
        if (ok) std::memcpy(s.h_frame.p, signals, frame_floats * sizeof(float));
        ok = ok && HIP_OK(hipMemcpyAsync(s.d_frame.p, s.h_frame.p, frame_floats * sizeof(float), hipMemcpyHostToDevice, s.stream));
        L.signals = s.d_frame.p; L.images = s.d_image.p; L.mics = s.d_mics.p;
        ok = ok && plan_or_error(L, &plan);
        ok = ok && ensure_digest(s.tab[slot], L, plan, s.stream);
        ok = ok && HIP_OK(bf::launch_das(L, plan, s.stream));
        ok = ok && HIP_OK(hipMemcpyAsync(s.h_image.p, s.d_image.p, D * sizeof(float), hipMemcpyDeviceToHost, s.stream));
        ok = ok && HIP_OK(sync_short(s.stream));
        if (ok) std::memcpy(image, s.h_image.p, D * sizeof(float));
    }
    if (!ok) poison(image, D);
}

// miso_*: one direction (table offset), raw out[N] (pad_and_sum.c:54-70 and twins).
void run_miso_host(int algo, const TableSet& t, const char* loader, bool fir, const float* signals, float* out,
                   const int* adaptive, int n, long long row_offset, const float* init)
{
    State& s = S();
    sizes_from_env_once();
    const size_t N = (size_t)s.sz.n_samples;
    if (!signals || !out || !adaptive) { set_error("null argument"); poison(out, out ? N : 0); return; }
    bool ok = ensure_device();
    bf::DasLaunch L{};
    bf::DasPlan plan{};
    int max_row = 0;
    ok = ok && upload_mics(adaptive, n, &max_row);
    if (ok) {
        if (!t.loaded) { set_error("%s has not been called", loader); ok = false; }
        const long long per = fir ? s.sz.n_taps : 1;
        if (ok && (row_offset < 0 || (row_offset + n) * per > t.entries)) {
            set_error("offset %lld + n %d exceeds the %d loaded coefficients", row_offset, n, t.entries);
            ok = false;
        }
        if (ok) {
            L.algo = algo;
            L.tab.whole = t.whole.p; L.tab.frac = t.frac.p; L.tab.taps = t.taps.p; L.tab.max_whole = t.max_whole;
            L.n_mics = n; L.n_samples = s.sz.n_samples; L.n_taps = s.sz.n_taps; L.n_dirs = 1;
            L.dir_begin = 0; L.dir_end = 1; L.image_stride = 1; L.image_origin = 0; L.frames = 1;
        }
    }
    if (ok) {
        const size_t rows = (size_t)max_row + 1;
        L.m_total = (int)rows;
        ok = HIP_OK(s.d_frame.reserve(rows * N)) && HIP_OK(s.d_out.reserve(N));
        ok = ok && HIP_OK(hipMemcpyAsync(s.d_frame.p, signals, rows * N * sizeof(float), hipMemcpyHostToDevice, s.stream));
        if (ok && init) {
            ok = HIP_OK(s.d_init.reserve(N)) && HIP_OK(hipMemcpyAsync(s.d_init.p, init, N * sizeof(float), hipMemcpyHostToDevice, s.stream));
        }
        L.signals = s.d_frame.p; L.images = nullptr; L.mics = s.d_mics.p;
        ok = ok && plan_or_error(L, &plan);
        ok = ok && HIP_OK(bf::launch_miso(L, plan, row_offset, init ? s.d_init.p : nullptr, s.d_out.p, s.stream));
        ok = ok && HIP_OK(hipMemcpyAsync(out, s.d_out.p, N * sizeof(float), hipMemcpyDeviceToHost, s.stream));
        ok = ok && HIP_OK(sync_short(s.stream));
    }
    if (!ok) poison(out, N);
}

// The single-signal helpers (pad_delay, lerp_delay, convolve_*_delay*): out (+)= delayed(signal).  Run as a
// one-mic, one-entry-table MISO launch with `out` as the initial accumulator.
void run_delay_host(int algo, const float* signal, float* out, bool accumulate, int whole, float h, const float* taps)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    const size_t N = (size_t)s.sz.n_samples;
    if (!signal || !out) { set_error("null argument"); poison(out, out ? N : 0); return; }
    bool ok = ensure_device();
    if (ok && (whole < 0)) { set_error("negative delay %d", whole); ok = false; }
    bf::DasLaunch L{};
    bf::DasPlan plan{};
    const int zero = 0;
    int max_row = 0;
    ok = ok && upload_mics(&zero, 1, &max_row);
    if (ok) {
        const int T = s.sz.n_taps;
        const int w = std::min(whole, s.sz.n_samples);
        ok = upload(s.d_one_whole, &w, 1) && upload(s.d_one_frac, &h, 1);
        if (ok && taps) ok = upload(s.d_one_taps, taps, (size_t)T);
        L.algo = algo;
        L.tab.whole = s.d_one_whole.p; L.tab.frac = s.d_one_frac.p; L.tab.taps = s.d_one_taps.p; L.tab.max_whole = w;
        L.n_mics = 1; L.m_total = 1; L.n_samples = s.sz.n_samples; L.n_taps = T; L.n_dirs = 1;
        L.dir_begin = 0; L.dir_end = 1; L.image_stride = 1; L.image_origin = 0; L.frames = 1;
    }
    if (ok) {
        ok = HIP_OK(s.d_frame.reserve(N)) && HIP_OK(s.d_out.reserve(N)) && HIP_OK(s.d_init.reserve(N));
        ok = ok && HIP_OK(hipMemcpyAsync(s.d_frame.p, signal, N * sizeof(float), hipMemcpyHostToDevice, s.stream));
        if (ok && accumulate) ok = HIP_OK(hipMemcpyAsync(s.d_init.p, out, N * sizeof(float), hipMemcpyHostToDevice, s.stream));
        L.signals = s.d_frame.p; L.mics = s.d_mics.p;
        ok = ok && plan_or_error(L, &plan);
        ok = ok && HIP_OK(bf::launch_miso(L, plan, 0, accumulate ? s.d_init.p : nullptr, s.d_out.p, s.stream));
        ok = ok && HIP_OK(hipMemcpyAsync(out, s.d_out.p, N * sizeof(float), hipMemcpyDeviceToHost, s.stream));
        ok = ok && HIP_OK(sync_short(s.stream));
    }
    if (!ok) poison(out, N);
}

// Integer table sanity shared by the loaders: no negative delays; values beyond N contribute nothing, so they
// are clamped to N (keeps the LDS zero prefix bounded).
bool sanitize_whole(std::vector<int32_t>& w, int n_samples, int* max_whole, const char* who)
{
    int mx = 0;
    for (size_t i = 0; i < w.size(); ++i) {
        if (w[i] < 0) { set_error("%s: negative delay %d at index %zu (the reference would write out of bounds)", who, w[i], i); return false; }
        if (w[i] > n_samples) w[i] = n_samples;
        mx = std::max(mx, w[i]);
    }
    *max_whole = mx;
    return true;
}

bool load_whole_only(int slot, const int* whole, int n, const char* who)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!whole || n < 1) { set_error("%s: null or empty table", who); return false; }
    if (!ensure_device()) return false;
    std::vector<int32_t> w(whole, whole + n);
    TableSet& t = s.tab[slot];
    int mx = 0;
    if (!sanitize_whole(w, s.sz.n_samples, &mx, who)) return false;
    if (!upload(t.whole, w.data(), w.size())) return false;
    t.loaded = true; t.entries = n; t.max_whole = mx;
    t.invalidate_digests();   // new table values: any digest built from the old ones is stale
    return true;
}

// hybrid_convolve_and_sum.c:124-157 (note its own pi constant and epsilon placement)
void hybrid_taps(float* h, double delay, int T)
{
    const double PI = 3.14159265359, eps = 1e-9;
    const double tau = 0.5 - delay + eps;
    double sum = 0.0;
    for (int i = 0; i < T; ++i) {
        double v = (double)i - ((double)T - 1.0) / 2.0 - tau;
        v = std::sin(v * PI) / (v * PI);
        const double n = (double)(i * 2 - T + 1);
        const double w = 0.42 + 0.5 * std::cos(PI * n / ((double)(T - 1)) + eps) + 0.08 * std::cos(2.0 * PI * n / ((double)(T - 1) + eps));
        v *= w;
        sum += v;
        h[i] = (float)v;
    }
    for (int i = 0; i < T; ++i) h[i] /= (float)sum;
}

template <typename F>
void parallel_for(size_t n, F body)
{
    unsigned hw = std::thread::hardware_concurrency();
    size_t workers = std::min<size_t>(hw ? hw : 1, 16);
    if (n < 65536 || workers < 2) { body(0, n); return; }
    std::vector<std::thread> pool;
    const size_t step = (n + workers - 1) / workers;
    for (size_t w = 0; w < workers; ++w) {
        const size_t lo = w * step, hi = std::min(n, lo + step);
        if (lo < hi) pool.emplace_back([=] { body(lo, hi); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace

// ============================================================================================== C-ABI
extern "C" {

// ---------------------------------------------------------------- configuration / errors

int bf_configure(int n_microphones, int n_samples, int max_res_x, int max_res_y, int n_taps)
{
    std::lock_guard<std::mutex> lock(S().mu);
    S().sizes_from_env_done = true;  // an explicit call wins over $BF_CONFIG
    return apply_sizes(n_microphones, n_samples, max_res_x, max_res_y, n_taps) ? 0 : -1;
}

int bf_configure_from_json(const char* path)
{
    std::lock_guard<std::mutex> lock(S().mu);
    S().sizes_from_env_done = true;
    return (path && configure_from_json(path)) ? 0 : -1;
}

void bf_get_config(int out[5])
{
    std::lock_guard<std::mutex> lock(S().mu);
    sizes_from_env_once();
    const Sizes& z = S().sz;
    out[0] = z.n_microphones; out[1] = z.n_samples; out[2] = z.res_x; out[3] = z.res_y; out[4] = z.n_taps;
}

const char* bf_last_error(void) { return S().err.c_str(); }
void bf_clear_error(void) { S().err.clear(); }

// Which kernel family the last delay-and-sum launch used: 0 strided, 1 quad + DPP, 2 shifted copies / sweep, 3 shifted copies /
// direction-outer (table without structure), 4 shifted copies / 8-tap FIR; -1 before the first launch.  For tests and tuning.
int bf_last_das_variant(void) { return S().last_variant; }
int bf_read_phase_stamps(unsigned long long* out16, int clear)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (!out16 || !ensure_device()) return -1;
    return HIP_OK(hipDeviceSynchronize()) && HIP_OK(bf::read_phase_stamps(out16, clear != 0)) ? 0 : -1;
}
void bf_set_debug(int flags) { std::lock_guard<std::mutex> lock(S().mu); S().debug_override = flags; }

int bf_gpu_available(void)
{
    int count = 0;
    return (hipGetDeviceCount(&count) == hipSuccess && count > 0) ? 1 : 0;
}

int bf_set_device(int device)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (S().device_ready && device != S().device) { set_error("bf_set_device: device %d already in use", S().device); return -1; }
    S().device = device;
    return 0;
}

// ---------------------------------------------------------------- pad

void load_coefficients_pad(int* whole_samples, int n) { (void)load_whole_only(SLOT_PAD, whole_samples, n, "load_coefficients_pad"); }
void unload_coefficients_pad(void) { std::lock_guard<std::mutex> lock(S().mu); S().tab[SLOT_PAD].drop(); }

void load_coefficients_pad2(int* whole_miso, int n)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (!whole_miso || n < 1) { set_error("load_coefficients_pad2: null or empty table"); return; }
    S().pad2_host.assign(whole_miso, whole_miso + n);
}
void unload_coefficients_pad2(void) { std::lock_guard<std::mutex> lock(S().mu); S().pad2_host.clear(); }

void mimo_pad(float* signals, float* image, int* adaptive_array, int n) { run_mimo_host(bf::ALGO_PAD, SLOT_PAD, signals, image, adaptive_array, n); }

void miso_pad(float* signals, float* out, int* adaptive_array, int n, int offset)
{
    std::lock_guard<std::mutex> lock(S().mu);
    run_miso_host(bf::ALGO_PAD, S().tab[SLOT_PAD], loader_name(SLOT_PAD), false, signals, out, adaptive_array, n, offset, nullptr);
}

void miso_pad2(float* signals, float* out, int* adaptive_array, int n, int offset)
{
    // pad_and_sum.c:77-92: delay looked up by MICROPHONE id in the pad2 table; `offset` is unused there too.
    (void)offset;
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    const size_t N = (size_t)s.sz.n_samples;
    if (!adaptive_array || n < 1) { set_error("miso_pad2: null or empty adaptive_array"); poison(out, N); return; }
    std::vector<int32_t> row((size_t)n);
    for (int m = 0; m < n; ++m) {
        const int mic = adaptive_array[m];
        if (mic < 0 || mic >= (int)s.pad2_host.size()) { set_error("miso_pad2: mic %d outside the %zu-entry pad2 table", mic, s.pad2_host.size()); poison(out, N); return; }
        row[(size_t)m] = s.pad2_host[(size_t)mic];
    }
    if (!ensure_device()) { poison(out, N); return; }
    static TableSet one;  // a private one-row table in slot order
    int mx = 0;
    if (!sanitize_whole(row, s.sz.n_samples, &mx, "miso_pad2") || !upload(one.whole, row.data(), row.size())) { poison(out, N); return; }
    one.loaded = true; one.entries = n; one.max_whole = mx;
    run_miso_host(bf::ALGO_PAD, one, "load_coefficients_pad2", false, signals, out, adaptive_array, n, 0, nullptr);
}

void pad_delay(float* signal, float* out, int pos_pad) { run_delay_host(bf::ALGO_PAD, signal, out, true, pos_pad, 0.f, nullptr); }

// ---------------------------------------------------------------- lerp

void load_coefficients_lerp(float* delays, int n)
{
    // lerp_and_sum.c:139-153: frac = modf((double)delay, &ip); h = 1.0 - (float)frac (rounded to float); whole = (int)ip
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!delays || n < 1) { set_error("load_coefficients_lerp: null or empty table"); return; }
    if (!ensure_device()) return;
    std::vector<int32_t> w((size_t)n);
    std::vector<float> h((size_t)n);
    for (int i = 0; i < n; ++i) {
        double ip;
        const float frac = (float)std::modf((double)delays[i], &ip);
        h[(size_t)i] = (float)(1.0 - (double)frac);
        w[(size_t)i] = (int)ip;
    }
    TableSet& t = s.tab[SLOT_LERP];
    int mx = 0;
    if (!sanitize_whole(w, s.sz.n_samples, &mx, "load_coefficients_lerp")) return;
    if (!upload(t.whole, w.data(), w.size()) || !upload(t.frac, h.data(), h.size())) return;
    t.loaded = true; t.entries = n; t.max_whole = mx;
    t.invalidate_digests();   // new table values: any digest built from the old ones is stale
}
void unload_coefficients_lerp(void) { std::lock_guard<std::mutex> lock(S().mu); S().tab[SLOT_LERP].drop(); }

void mimo_lerp(float* signals, float* image, int* adaptive_array, int n) { run_mimo_host(bf::ALGO_LERP, SLOT_LERP, signals, image, adaptive_array, n); }
void miso_lerp(float* signals, float* out, int* adaptive_array, int n, int offset)
{
    std::lock_guard<std::mutex> lock(S().mu);
    run_miso_host(bf::ALGO_LERP, S().tab[SLOT_LERP], loader_name(SLOT_LERP), false, signals, out, adaptive_array, n, offset, nullptr);
}
void lerp_delay(float* signal, float* out, float h, int pad) { run_delay_host(bf::ALGO_LERP, signal, out, true, pad, h, nullptr); }

int bf_get_lerp_tables(int* whole, float* h, int n)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    const TableSet& t = s.tab[SLOT_LERP];
    if (!t.loaded || n != t.entries) { set_error("bf_get_lerp_tables: %d requested, %d loaded", n, t.loaded ? t.entries : 0); return -1; }
    bool ok = HIP_OK(hipMemcpy(whole, t.whole.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    ok = ok && HIP_OK(hipMemcpy(h, t.frac.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return ok ? 0 : -1;
}

// ---------------------------------------------------------------- convolve (full FIR)

void load_coefficients_convolve(float* h, int n)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!h || n < 1) { set_error("load_coefficients_convolve: null or empty table"); return; }
    if (!ensure_device()) return;
    TableSet& t = s.tab[SLOT_FIR];
    if (!upload(t.taps, h, (size_t)n)) return;
    t.loaded = true; t.entries = n; t.max_whole = 0;
    t.invalidate_digests();   // new table values: any digest built from the old ones is stale
}
void unload_coefficients_convolve(void) { std::lock_guard<std::mutex> lock(S().mu); S().tab[SLOT_FIR].drop(); }

void mimo_convolve_naive(float* signals, float* image, int* adaptive_array, int n) { run_mimo_host(bf::ALGO_FIR_NAIVE, SLOT_FIR, signals, image, adaptive_array, n); }
void mimo_convolve_vectorized(float* signals, float* image, int* adaptive_array, int n) { run_mimo_host(bf::ALGO_FIR_VEC, SLOT_FIR, signals, image, adaptive_array, n); }

void miso_convolve_vectorized(float* signals, float* out, int* adaptive_array, int n, int offset)
{
    // convolve_and_sum.c:276-292: offset counts taps (d * n * N_TAPS)
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    const int T = s.sz.n_taps;
    if (offset % T != 0) { set_error("miso_convolve_vectorized: offset %d is not a multiple of N_TAPS=%d", offset, T); poison(out, (size_t)s.sz.n_samples); return; }
    run_miso_host(bf::ALGO_FIR_VEC, s.tab[SLOT_FIR], loader_name(SLOT_FIR), true, signals, out, adaptive_array, n, offset / T, nullptr);
}

void convolve_delay_naive_add(float* signal, float* h, float* out) { run_delay_host(bf::ALGO_FIR_NAIVE, signal, out, true, 0, 0.f, h); }
void convolve_delay_naive(float* signal, float* out, float* h) { run_delay_host(bf::ALGO_FIR_NAIVE, signal, out, true, 0, 0.f, h); }
void convolve_delay_vectorized(float* signal, float* h, float* out) { run_delay_host(bf::ALGO_FIR_NAIVE, signal, out, false, 0, 0.f, h); }
void convolve_delay_vectorized_add(float* signal, float* h, float* out) { run_delay_host(bf::ALGO_FIR_VEC, signal, out, true, 0, 0.f, h); }

// ---------------------------------------------------------------- hybrid

void load_coefficients_convolve_hybrid(float* delays, int n)
{
    // hybrid_convolve_and_sum.c:161-180
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!delays || n < 1) { set_error("load_coefficients_convolve_hybrid: null or empty table"); return; }
    if (!ensure_device()) return;
    const int T = s.sz.n_taps;
    std::vector<int32_t> w((size_t)n);
    std::vector<float> taps((size_t)n * T);
    parallel_for((size_t)n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            double ip;
            const double fraction = 1.0 - std::modf((double)delays[i], &ip);
            w[i] = (int)ip;
            hybrid_taps(taps.data() + i * T, fraction, T);
        }
    });
    TableSet& t = s.tab[SLOT_HYBRID];
    int mx = 0;
    if (!sanitize_whole(w, s.sz.n_samples, &mx, "load_coefficients_convolve_hybrid")) return;
    if (!upload(t.whole, w.data(), w.size()) || !upload(t.taps, taps.data(), taps.size())) return;
    t.loaded = true; t.entries = n; t.max_whole = mx;
    t.invalidate_digests();   // new table values: any digest built from the old ones is stale
}
void unload_coefficients_convolve_hybrid(void) { std::lock_guard<std::mutex> lock(S().mu); S().tab[SLOT_HYBRID].drop(); }

void mimo_convolve_hybrid(float* signals, float* image, int* adaptive_array, int n) { run_mimo_host(bf::ALGO_HYBRID, SLOT_HYBRID, signals, image, adaptive_array, n); }
void miso_convolve_hybrid(float* signals, float* out, int* adaptive_array, int n, int offset)
{
    std::lock_guard<std::mutex> lock(S().mu);
    run_miso_host(bf::ALGO_HYBRID, S().tab[SLOT_HYBRID], loader_name(SLOT_HYBRID), false, signals, out, adaptive_array, n, offset, nullptr);
}
void convolve_hybrid_delay_add(float* signal, float* h, int pad, float* out) { run_delay_host(bf::ALGO_HYBRID, signal, out, true, pad, 0.f, h); }

int bf_get_hybrid_tables(int* whole, float* taps, int n)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    const TableSet& t = s.tab[SLOT_HYBRID];
    if (!t.loaded || n != t.entries) { set_error("bf_get_hybrid_tables: %d requested, %d loaded", n, t.loaded ? t.entries : 0); return -1; }
    bool ok = HIP_OK(hipMemcpy(whole, t.whole.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    ok = ok && HIP_OK(hipMemcpy(taps, t.taps.p, (size_t)n * s.sz.n_taps * sizeof(float), hipMemcpyDeviceToHost));
    return ok ? 0 : -1;
}

// ---------------------------------------------------------------- api.h shims

void bf_publish_frame(const float* signals)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    const size_t n = (size_t)s.sz.n_microphones * s.sz.n_samples;
    if (!signals) { set_error("bf_publish_frame: null frame"); return; }
    s.published.assign(signals, signals + n);
}

static bool copy_published(float* out, bool mask_dead)
{
    State& s = S();
    sizes_from_env_once();
    const size_t n = (size_t)s.sz.n_microphones * s.sz.n_samples;
    if (s.published.size() != n) { set_error("get_data: no frame published (call bf_publish_frame first)"); poison(out, n); return false; }
    std::memcpy(out, s.published.data(), n * sizeof(float));
    if (mask_dead) {
        if (s.disabled_default) {
            // PC/src/api.c:835-856 zeroes the rows of the 122 microphones that are dead on the authors' arrays.
            static const short dead[] = {0, 1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30,
                31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64,
                83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95, 96, 98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111,
                112, 135, 137, 143, 145, 146, 147, 148, 149, 150, 151, 152, 153, 154, 159, 160, 162, 163, 164, 165, 166, 167, 169, 175,
                184, 192, 193, 194, 195, 196, 197, 198, 199, 200, 201};
            s.disabled_mics.assign(dead, dead + sizeof(dead) / sizeof(dead[0]));
            s.disabled_default = false;
        }
        for (int mic : s.disabled_mics)
            if (mic >= 0 && mic < s.sz.n_microphones) std::memset(out + (size_t)mic * s.sz.n_samples, 0, (size_t)s.sz.n_samples * sizeof(float));
    }
    return true;
}

void get_data(float* signals)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (signals) (void)copy_published(signals, true);
}

static void shim(int algo, int slot, float* image, int* adaptive_array, int n, bool mask_dead)
{
    std::vector<float> frame;
    {
        State& s = S();
        std::lock_guard<std::mutex> lock(s.mu);
        sizes_from_env_once();
        frame.resize((size_t)s.sz.n_microphones * s.sz.n_samples);
        if (!copy_published(frame.data(), mask_dead)) { poison(image, (size_t)s.sz.dirs()); return; }
    }
    run_mimo_host(algo, slot, frame.data(), image, adaptive_array, n);
}

void pad_mimo(float* image, int* adaptive_array, int n) { shim(bf::ALGO_PAD, SLOT_PAD, image, adaptive_array, n, true); }
void lerp_mimo(float* image, int* adaptive_array, int n) { shim(bf::ALGO_LERP, SLOT_LERP, image, adaptive_array, n, true); }
void convolve_mimo_naive(float* image, int* adaptive_array, int n) { shim(bf::ALGO_FIR_NAIVE, SLOT_FIR, image, adaptive_array, n, true); }
void convolve_mimo_vectorized(float* image, int* adaptive_array, int n) { shim(bf::ALGO_FIR_VEC, SLOT_FIR, image, adaptive_array, n, true); }

void load_coefficients2(int* whole_samples, int n) { (void)load_whole_only(SLOT_TRUNC, whole_samples, n, "load_coefficients2"); }
// api.c:1077-1087 copies the ring buffer WITHOUT the dead-microphone mask, then runs the pad algorithm
void mimo_truncated(float* image, int* adaptive_array, int n) { shim(bf::ALGO_PAD, SLOT_TRUNC, image, adaptive_array, n, false); }

void miso_steer_listen(float* out, int* adaptive_array, int n, int steer_offset)
{
    std::vector<float> frame;
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    frame.resize((size_t)s.sz.n_microphones * s.sz.n_samples);
    if (!copy_published(frame.data(), true)) { poison(out, (size_t)s.sz.n_samples); return; }
    run_miso_host(bf::ALGO_PAD, s.tab[SLOT_PAD], loader_name(SLOT_PAD), false, frame.data(), out, adaptive_array, n, steer_offset, nullptr);
}

// ---- api.h:6-9,41-45: process management around the path.  The receiver / playback children are live-hardware I/O and
// out of scope; what these keep is the STATE the path reads (listening offset, listening microphones).

int load(bool replay_mode)
{
    (void)replay_mode;
    std::lock_guard<std::mutex> lock(S().mu);
    set_error("load: the UDP receiver process (PC/src/api.c:874-939) is out of scope of this library; hand frames over with bf_publish_frame");
    return -1;
}

void stop_receiving(void)
{
    std::lock_guard<std::mutex> lock(S().mu);
    S().published.clear();
}

void signal_handler(void) {}

int load_miso(void)
{
    // miso_init_shared_memory (api.c:461-489) + the steer(0) that opens miso_loop (api.c:493): n = 1, adaptive_array all zero
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    s.listen_mics.assign(1, 0);
    s.steer_offset = 0;
    return 0;
}

void load_pa(int* adaptive_array, int n)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    if (!adaptive_array || n < 1) { set_error("load_pa: null or empty adaptive_array"); return; }
    s.listen_mics.assign(adaptive_array, adaptive_array + n);
}

void stop_miso(void)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    s.listen_mics.clear();
    s.steer_offset = 0;
}

void steer(int offset)
{
    std::lock_guard<std::mutex> lock(S().mu);
    S().steer_offset = offset;
}

int bf_get_steer(int* n_out)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (n_out) *n_out = (int)S().listen_mics.size();
    return S().steer_offset;
}

int bf_miso_listen_block(float* out, float mic_gain)
{
    // miso_loop's body (api.c:505-531): get_data; miso_pad(signals, out, adaptive_array, n, steer_offset); out[i] /= n; out[i] *= MIC_GAIN
    std::vector<float> frame;
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    const size_t N = (size_t)s.sz.n_samples;
    if (!out) { set_error("bf_miso_listen_block: null output"); return -1; }
    if (s.listen_mics.empty()) { set_error("bf_miso_listen_block: load_miso / load_pa has not been called"); poison(out, N); return -1; }
    frame.resize((size_t)s.sz.n_microphones * N);
    if (!copy_published(frame.data(), true)) { poison(out, N); return -1; }
    const std::string before = s.err;
    s.err.clear();
    std::vector<int> mics = s.listen_mics;     // run_miso_host uploads from a stable copy
    run_miso_host(bf::ALGO_PAD, s.tab[SLOT_PAD], loader_name(SLOT_PAD), false, frame.data(), out, mics.data(), (int)mics.size(), s.steer_offset, nullptr);
    if (!s.err.empty()) return -1;
    s.err = before;
    const float fn = (float)mics.size();
    for (size_t i = 0; i < N; ++i) {
        out[i] /= fn;
        out[i] *= mic_gain;
    }
    return 0;
}

// ---------------------------------------------------------------- device-resident batched path

int bf_das_device(int algo, const float* d_signals, int m_total, float* d_images, int image_stride, int frames,
                  const int* adaptive_array, int n, int dir_begin, int dir_end, void* stream)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    const int slot = slot_of(algo);
    if (slot < 0) { set_error("bf_das_device: unknown algo %d", algo); return -1; }
    if (!d_signals || !d_images || !adaptive_array || frames < 1) { set_error("bf_das_device: null argument or frames < 1"); return -1; }
    if (!ensure_device()) return -1;
    int max_row = 0;
    if (!upload_mics(adaptive_array, n, &max_row)) return -1;
    if (max_row >= m_total) { set_error("bf_das_device: adaptive_array names row %d but frames have %d rows", max_row, m_total); return -1; }
    bf::DasLaunch L{};
    if (!describe(algo, slot, n, &L)) return -1;
    if (dir_begin < 0 || dir_end > L.n_dirs || dir_begin >= dir_end) { set_error("bf_das_device: bad direction range [%d,%d) of %d", dir_begin, dir_end, L.n_dirs); return -1; }
    if (image_stride < dir_end - dir_begin) { set_error("bf_das_device: image_stride %d < %d directions", image_stride, dir_end - dir_begin); return -1; }
    L.signals = d_signals; L.images = d_images; L.m_total = m_total; L.frames = frames;
    L.dir_begin = dir_begin; L.dir_end = dir_end; L.image_stride = image_stride; L.image_origin = dir_begin;
    bf::DasPlan plan{};
    if (!plan_or_error(L, &plan)) return -1;
    if (!ensure_digest(s.tab[slot], L, plan, reinterpret_cast<hipStream_t>(stream))) return -1;
    return HIP_OK(bf::launch_das(L, plan, reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}

// ---------------------------------------------------------------- ingest (receiver.c:94-151)

static int ingest_common(const void* d_packets, int n_arrays, int rows, int columns, float* d_frame, hipStream_t stream)
{
    State& s = S();
    const int mics_out = n_arrays * rows * columns;
    if (n_arrays < 1 || rows < 1 || columns < 1 || mics_out > s.sz.n_microphones) {
        set_error("bf_ingest: n_arrays*rows*columns = %d does not fit N_MICROPHONES = %d", mics_out, s.sz.n_microphones);
        return -1;
    }
    const int stride = 8 + 4 * s.sz.n_microphones;   // sizeof(msg), receiver.h:51-59
    return HIP_OK(bf::launch_ingest(d_packets, stride, 8, s.sz.n_samples, mics_out, s.sz.n_microphones, rows, columns, d_frame, stream)) ? 0 : -1;
}

int bf_ingest_device(const void* d_packets, int n_arrays, int rows, int columns, float* d_frame, void* stream)
{
    std::lock_guard<std::mutex> lock(S().mu);
    sizes_from_env_once();
    if (!d_packets || !d_frame) { set_error("bf_ingest_device: null argument"); return -1; }
    if (!ensure_device()) return -1;
    return ingest_common(d_packets, n_arrays, rows, columns, d_frame, reinterpret_cast<hipStream_t>(stream));
}

int bf_ingest(const void* packets, int n_arrays, int rows, int columns, float* frame)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    if (!packets || !frame) { set_error("bf_ingest: null argument"); return -1; }
    const size_t out_n = (size_t)std::max(0, n_arrays * rows * columns) * s.sz.n_samples;
    if (!ensure_device()) { poison(frame, out_n); return -1; }
    const size_t in_bytes = (size_t)(8 + 4 * s.sz.n_microphones) * s.sz.n_samples;
    static DevBuf<unsigned char> d_in;
    bool ok = HIP_OK(d_in.reserve(in_bytes)) && HIP_OK(s.d_frame.reserve(out_n ? out_n : 1));
    ok = ok && HIP_OK(hipMemcpyAsync(d_in.p, packets, in_bytes, hipMemcpyHostToDevice, s.stream));
    ok = ok && ingest_common(d_in.p, n_arrays, rows, columns, s.d_frame.p, s.stream) == 0;
    ok = ok && HIP_OK(hipMemcpyAsync(frame, s.d_frame.p, out_n * sizeof(float), hipMemcpyDeviceToHost, s.stream));
    ok = ok && HIP_OK(sync_short(s.stream));
    if (!ok) poison(frame, out_n);
    return ok ? 0 : -1;
}

// ---------------------------------------------------------------- heat-map post-processing (visual.py)

int bf_heatmap_colorize_device(const float* d_power, int frames, float threshold, float amount, float exponent, unsigned char* d_small,
                               int* d_should_overlay, void* stream)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    if (!d_power || !d_small || !d_should_overlay || frames < 1) { set_error("bf_heatmap_colorize_device: null argument or frames < 1"); return -1; }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_colorize(d_power, frames, s.sz.res_x, s.sz.res_y, threshold, amount, exponent, d_small, d_should_overlay,
                                      reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}

int bf_heatmap_overlay_device(const unsigned char* d_small, int frames, int out_w, int out_h, unsigned char* d_prev, const unsigned char* d_camera,
                              unsigned char* d_out, float w_prev, float w_new, float w_cam, float w_heat, void* stream)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    if (!d_small || !d_prev || !d_out || frames < 1 || out_w < 1 || out_h < 1) { set_error("bf_heatmap_overlay_device: bad argument"); return -1; }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_overlay(d_small, frames, s.sz.res_x, s.sz.res_y, out_w, out_h, d_prev, d_camera, d_out, w_prev, w_new, w_cam, w_heat,
                                     reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}

int bf_power_center_device(const float* d_power, int frames, float* d_centers, float* d_workspace, void* stream)
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    if (!d_power || !d_centers || !d_workspace || frames < 1) { set_error("bf_power_center_device: bad argument"); return -1; }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_power_center(d_power, frames, s.sz.res_x, s.sz.res_y, d_centers, d_workspace, reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}

int bf_letterbox_bgr8_device(const unsigned char* d_frame, int h, int w, unsigned char* d_out, int out_h, int out_w, int new_h, int new_w, int top, int left, int value,
                             void* stream)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (!d_frame || !d_out || h < 1 || w < 1 || out_h < 1 || out_w < 1 || new_h < 1 || new_w < 1 || top < 0 || left < 0 || top + new_h > out_h || left + new_w > out_w ||
        value < 0 || value > 255) {
        set_error("bf_letterbox_bgr8_device: %d x %d -> %d x %d at (%d, %d) of %d x %d, border %d", h, w, new_h, new_w, top, left, out_h, out_w, value);
        return -1;
    }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_letterbox(d_frame, h, w, d_out, out_h, out_w, new_h, new_w, top, left, value, reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}

// ---------------------------------------------------------------- frequency-domain beamformers

#define FD_ENTER(cond, name)                                                     \
    State& s = S();                                                              \
    std::lock_guard<std::mutex> lock(s.mu);                                      \
    sizes_from_env_once();                                                       \
    if (!(cond)) { set_error(name ": null pointer or non-positive size"); return -1; } \
    if (!ensure_device()) return -1;                                             \
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

int bf_fd_steering_device(const double* d_tau, const double* d_freq, int n_dirs, int n_mics, int n_bins, float* d_are, float* d_aim, void* stream)
{
    FD_ENTER(d_tau && d_freq && d_are && d_aim && n_dirs > 0 && n_mics > 0 && n_bins > 0, "bf_fd_steering_device")
    return HIP_OK(bf::launch_fd_steering(d_tau, d_freq, n_dirs, n_mics, n_bins, d_are, d_aim, st)) ? 0 : -1;
}

int bf_fd_dft_device(const float* d_frames, int m_total, int frames, const int* adaptive_array, int n, int bin_lo, int n_bins, float* d_xre_mf,
                     float* d_xim_mf, float* d_xre_fm, float* d_xim_fm, void* stream)
{
    FD_ENTER(d_frames && adaptive_array && d_xre_mf && d_xim_mf && d_xre_fm && d_xim_fm && frames > 0 && n > 0 && n_bins > 0 && bin_lo >= 0,
             "bf_fd_dft_device")
    if (bin_lo + n_bins > s.sz.n_samples / 2 + 1) { set_error("bf_fd_dft_device: bins [%d,%d) exceed N_SAMPLES/2+1 = %d", bin_lo, bin_lo + n_bins, s.sz.n_samples / 2 + 1); return -1; }
    int max_row = 0;
    if (!upload_mics(adaptive_array, n, &max_row)) return -1;
    if (max_row >= m_total) { set_error("bf_fd_dft_device: adaptive_array names row %d but frames have %d rows", max_row, m_total); return -1; }
    const long long key = ((long long)s.sz.n_samples << 40) ^ ((long long)bin_lo << 20) ^ n_bins;
    if (s.fd_tw_key != key || !s.fd_tw.p) {
        if (!HIP_OK(s.fd_tw.reserve(bf::fd_twiddle_floats(s.sz.n_samples, n_bins)))) return -1;
        if (!HIP_OK(bf::launch_fd_twiddles(s.sz.n_samples, bin_lo, n_bins, s.fd_tw.p, st))) return -1;
        if (!HIP_OK(hipStreamSynchronize(st))) return -1;     // once per (N, bin range): see ensure_digest
        s.fd_tw_key = key;
    }
    return HIP_OK(bf::launch_fd_dft(d_frames, s.d_mics.p, m_total, s.sz.n_samples, frames, n, bin_lo, n_bins, s.fd_tw.p, d_xre_mf, d_xim_mf, d_xre_fm, d_xim_fm,
                                    st)) ? 0 : -1;
}

int bf_fd_das_power_device(const float* d_xre_mf, const float* d_xim_mf, const float* d_are, const float* d_aim, int frames, int n_mics, int n_dirs,
                           int n_bins, float* d_power, void* stream)
{
    FD_ENTER(d_xre_mf && d_xim_mf && d_are && d_aim && d_power && frames > 0 && n_mics > 0 && n_dirs > 0 && n_bins > 0, "bf_fd_das_power_device")
    const size_t work = bf::fd_workspace_floats(frames, n_dirs, n_bins);
    if (!HIP_OK(s.fd_work.reserve(work))) return -1;
    return HIP_OK(bf::launch_fd_das_power(d_xre_mf, d_xim_mf, d_are, d_aim, frames, n_mics, n_dirs, n_bins, d_power, s.fd_work.p, s.fd_work.cap, st)) ? 0 : -1;
}

int bf_fd_covariance_device(const float* d_xre_fm, const float* d_xim_fm, int frames, int n_mics, int n_bins, float* d_rre, float* d_rim, void* stream)
{
    FD_ENTER(d_xre_fm && d_xim_fm && d_rre && d_rim && frames > 0 && n_mics > 0 && n_bins > 0, "bf_fd_covariance_device")
    return HIP_OK(bf::launch_fd_covariance(d_xre_fm, d_xim_fm, frames, n_mics, n_bins, d_rre, d_rim, st)) ? 0 : -1;
}

int bf_fd_cholesky_inverse_device(const float* d_rre, const float* d_rim, int n_mics, int n_bins, float loading, float* d_lire_t, float* d_liim_t,
                                  int* d_status, void* stream)
{
    FD_ENTER(d_rre && d_rim && d_lire_t && d_liim_t && d_status && n_mics > 0 && n_bins > 0, "bf_fd_cholesky_inverse_device")
    if (n_mics > 256) { set_error("bf_fd_cholesky_inverse_device: %d mics; the blocked factorisation handles at most 256", n_mics); return -1; }
    const size_t work = bf::fd_cholesky_workspace_floats(n_mics, n_bins);
    if (work && !HIP_OK(s.fd_chol_work.reserve(work))) return -1;
    return HIP_OK(bf::launch_fd_cholesky_inverse(d_rre, d_rim, n_mics, n_bins, loading, d_lire_t, d_liim_t, d_status, s.fd_chol_work.p, s.fd_chol_work.cap, st))
               ? 0 : -1;
}

int bf_fd_mvdr_power_device(const float* d_lire_t, const float* d_liim_t, const float* d_are, const float* d_aim, int n_mics, int n_dirs, int n_bins,
                            float* d_power, void* stream)
{
    FD_ENTER(d_lire_t && d_liim_t && d_are && d_aim && d_power && n_mics > 0 && n_dirs > 0 && n_bins > 0, "bf_fd_mvdr_power_device")
    if (n_mics > 256) { set_error("bf_fd_mvdr_power_device: %d mics; at most 256", n_mics); return -1; }
    const size_t work = bf::fd_workspace_floats(1, n_dirs, n_bins);
    if (!HIP_OK(s.fd_work.reserve(work))) return -1;
    return HIP_OK(bf::launch_fd_mvdr_power(d_lire_t, d_liim_t, d_are, d_aim, n_mics, n_dirs, n_bins, d_power, s.fd_work.p, s.fd_work.cap, st)) ? 0 : -1;
}

// The 256-entry colour table of the colourise kernel (visual.py:26-49 generate_color_map("jet")), uint8 [256][3].
void bf_jet_lut(unsigned char* out768)
{
    struct Rgb { unsigned char r, g, b; };
    static const Rgb kJet[256] = {
#include "jet_lut.inc"
    };
    if (!out768) return;
    for (int i = 0; i < 256; ++i) { out768[3 * i] = kJet[i].r; out768[3 * i + 1] = kJet[i].g; out768[3 * i + 2] = kJet[i].b; }
}

// ---------------------------------------------------------------- detector post-processing

int bf_yolo_decode_device(const void* const raw[3], const int h[3], const int w[3], const int strides[3], const float* anchors, int batch, int nc,
                          int format, float conf_thres, float* d_boxes, float* d_scores, int* d_cls, void* stream)
{
    FD_ENTER(raw && raw[0] && raw[1] && raw[2] && h && w && strides && anchors && d_boxes && d_scores && d_cls && batch > 0 && nc > 0, "bf_yolo_decode_device")
    return HIP_OK(bf::launch_yolo_decode(raw, h, w, strides, anchors, batch, nc, format, conf_thres, d_boxes, d_scores, d_cls, st)) ? 0 : -1;
}

int bf_topk_candidates_device(const float* d_scores, const float* d_boxes, const int* d_cls, int batch, int total, int k, float* d_top_scores,
                              float* d_top_boxes, int* d_top_cls, int* d_counts, void* stream)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (!d_scores || !d_boxes || !d_cls || !d_top_scores || !d_top_boxes || !d_top_cls || !d_counts) { set_error("bf_topk_candidates_device: null pointer"); return -1; }
    if (k < 1 || k > 1024 || batch < 1 || total < 1) { set_error("bf_topk_candidates_device: batch %d, %d boxes, k = %d (1..1024)", batch, total, k); return -1; }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_topk_candidates(d_scores, d_boxes, d_cls, batch, total, k, d_top_scores, d_top_boxes, d_top_cls, d_counts,
                                             reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}

static int upsample_concat_checked(const char* who, int eb, const void* d_a, const void* d_b, void* d_out, int batch, int h, int w, int ca, int cb, void* stream)
{
    std::lock_guard<std::mutex> lock(S().mu);
    const int E = 16 / eb;
    if (!d_a || !d_b || !d_out || batch < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || ca < E || cb < E || (ca % E) || (cb % E)) {
        set_error("%s: batch %d, %d x %d (even), %d + %d channels (multiples of %d)", who, batch, h, w, ca, cb, E);
        return -1;
    }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_upsample_concat(d_a, d_b, d_out, batch, h, w, ca, cb, eb, reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}
int bf_upsample_concat_device(const void* d_a, const void* d_b, void* d_out, int batch, int h, int w, int ca, int cb, void* stream)
{
    return upsample_concat_checked("bf_upsample_concat_device", 2, d_a, d_b, d_out, batch, h, w, ca, cb, stream);
}
int bf_upsample_concat_f32_device(const void* d_a, const void* d_b, void* d_out, int batch, int h, int w, int ca, int cb, void* stream)
{
    return upsample_concat_checked("bf_upsample_concat_f32_device", 4, d_a, d_b, d_out, batch, h, w, ca, cb, stream);
}

static int sppf_pool_checked(const char* who, int eb, void* d_buf, int batch, int h, int w, int c, void* stream)
{
    std::lock_guard<std::mutex> lock(S().mu);
    const int E = 16 / eb;
    if (!d_buf || batch < 1 || h < 1 || w < 1 || c < E || (c % E) || (long long)h * w > 2048) {
        set_error("%s: batch %d, %d x %d, %d channels (a multiple of %d; at most 2048 pixels per plane)", who, batch, h, w, c, E);
        return -1;
    }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_sppf_pool(d_buf, batch, h, w, c, eb, reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}
int bf_sppf_pool_device(void* d_buf, int batch, int h, int w, int c, void* stream) { return sppf_pool_checked("bf_sppf_pool_device", 2, d_buf, batch, h, w, c, stream); }
int bf_sppf_pool_f32_device(void* d_buf, int batch, int h, int w, int c, void* stream) { return sppf_pool_checked("bf_sppf_pool_f32_device", 4, d_buf, batch, h, w, c, stream); }

static int preprocess_checked(const char* who, int eb, const void* d_frames, void* d_out, int batch, int h, int w, int cpad, void* stream)
{
    std::lock_guard<std::mutex> lock(S().mu);
    if (!d_frames || !d_out || batch < 1 || h < 1 || w < 1 || cpad < 3) { set_error("%s: batch %d, %d x %d, %d channels", who, batch, h, w, cpad); return -1; }
    if (!ensure_device()) return -1;
    return HIP_OK(bf::launch_preprocess_bgr8(d_frames, d_out, (long long)batch * h * w, cpad, eb, reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}
int bf_preprocess_bgr8_device(const void* d_frames, void* d_out, int batch, int h, int w, int cpad, void* stream)
{
    return preprocess_checked("bf_preprocess_bgr8_device", 2, d_frames, d_out, batch, h, w, cpad, stream);
}
int bf_preprocess_bgr8_f32_device(const void* d_frames, void* d_out, int batch, int h, int w, int cpad, void* stream)
{
    return preprocess_checked("bf_preprocess_bgr8_f32_device", 4, d_frames, d_out, batch, h, w, cpad, stream);
}

int bf_conv2d_use_dma_kernel(int enable) { return bf::conv_dma_switch(enable); }
int bf_conv2d_f32_mode(int mode) { return bf::conv_f32_mode(mode); }
int bf_fd_gemm_f32_mode(int mode) { return bf::gemm_f32_mode(mode); }

int bf_conv2d_weight_row(int kh, int kw, int c) { return kh > 0 && kw > 0 && c > 0 ? bf::conv_weight_row(2, kh, kw, c) : -1; }
int bf_conv2d_weight_row_f32(int kh, int kw, int c) { return kh > 0 && kw > 0 && c > 0 ? bf::conv_weight_row(4, kh, kw, c) : -1; }

static int conv2d_checked(const char* who, int eb, const void* d_x, const void* d_w, const float* d_bias, void* d_y, int batch, int h, int w, int c, int n, int kh,
                          int kw, int stride, int pad, int silu, int ldy, const void* d_res, int ldr, void* stream, bool cat = false, const void* d_x2 = nullptr,
                          int c1 = 0, int ld1 = 0, int ld2 = 0, int up1 = 0)
{
    std::lock_guard<std::mutex> lock(S().mu);
    const int E = 16 / eb;
    if (!d_x || !d_w || !d_y) { set_error("%s: null pointer", who); return -1; }
    if (batch < 1 || h < 1 || w < 1 || n < 1 || kh < 1 || kw < 1 || stride < 1 || pad < 0 || h + 2 * pad < kh || w + 2 * pad < kw) {
        set_error("%s: batch %d, %d x %d, window %d x %d, stride %d, pad %d", who, batch, h, w, kh, kw, stride, pad);
        return -1;
    }
    if (c < 4 || (c & (c - 1)) != 0 || ((kw * c) % E) != 0 || (c < E && ((stride & 1) || (pad & 1) || (w & 1)))) {
        set_error("%s: %d input channels, window width %d, stride %d, pad %d, width %d: channels must be a power of two >= 4 with "
                  "kw * c a multiple of %d; float16 with 4 channels needs even stride, pad and width", who, c, kw, stride, pad, w, E);
        return -1;
    }
    if (ldy < n || (d_res && ldr < n)) { set_error("%s: row strides %d / %d under %d output channels", who, ldy, ldr, n); return -1; }
    if ((reinterpret_cast<uintptr_t>(d_x) & 15) || (reinterpret_cast<uintptr_t>(d_w) & 15)) { set_error("%s: x and w must be 16-byte aligned", who); return -1; }
    if (cat) {
        if (c1 < E || c1 > c || (c1 % E) || (ld1 % E) || ld1 < c1 || (c1 < c && (!d_x2 || (ld2 % E) || ld2 < c - c1 || (reinterpret_cast<uintptr_t>(d_x2) & 15))) ||
            (up1 && ((h & 1) || (w & 1)))) {
            set_error("%s: sources of %d (pitch %d%s) + %d (pitch %d) channels for %d: whole 16-byte chunks of %d elements, 16-byte aligned; an upsampled "
                      "source needs even h and w", who, c1, ld1, up1 ? ", upsampled" : "", c - c1, ld2, c, E);
            return -1;
        }
        if (c1 == c) d_x2 = nullptr;
    }
    if (!ensure_device()) return -1;
    // (a plain dense source through the cat entry still takes the 1x1 path: ld1 = c selects it only when something differs -- force it with up1 / x2 / ld1)
    return HIP_OK(bf::launch_conv2d_nhwc(eb, d_x, d_w, d_bias, d_y, batch, h, w, c, n, kh, kw, stride, pad, silu, ldy, d_res, ldr, cat ? d_x2 : nullptr,
                                         cat ? c1 : c, cat ? ld1 : c, cat ? ld2 : 0, cat ? up1 : 0, reinterpret_cast<hipStream_t>(stream))) ? 0 : -1;
}

int bf_conv2d_nhwc_f16_device(const void* d_x, const void* d_w, const float* d_bias, void* d_y, int batch, int h, int w, int c, int n, int kh, int kw, int stride,
                              int pad, int silu, void* stream)
{
    return conv2d_checked("bf_conv2d_nhwc_f16_device", 2, d_x, d_w, d_bias, d_y, batch, h, w, c, n, kh, kw, stride, pad, silu, n, nullptr, 0, stream);
}

int bf_conv2d_nhwc_f16_into_device(const void* d_x, const void* d_w, const float* d_bias, void* d_y, int ldy, const void* d_res, int ldr, int batch, int h, int w,
                                   int c, int n, int kh, int kw, int stride, int pad, int silu, void* stream)
{
    return conv2d_checked("bf_conv2d_nhwc_f16_into_device", 2, d_x, d_w, d_bias, d_y, batch, h, w, c, n, kh, kw, stride, pad, silu, ldy, d_res, ldr, stream);
}

int bf_conv2d_nhwc_f32_device(const void* d_x, const void* d_w, const float* d_bias, void* d_y, int batch, int h, int w, int c, int n, int kh, int kw, int stride,
                              int pad, int silu, void* stream)
{
    return conv2d_checked("bf_conv2d_nhwc_f32_device", 4, d_x, d_w, d_bias, d_y, batch, h, w, c, n, kh, kw, stride, pad, silu, n, nullptr, 0, stream);
}

int bf_conv2d_nhwc_f32_into_device(const void* d_x, const void* d_w, const float* d_bias, void* d_y, int ldy, const void* d_res, int ldr, int batch, int h, int w,
                                   int c, int n, int kh, int kw, int stride, int pad, int silu, void* stream)
{
    return conv2d_checked("bf_conv2d_nhwc_f32_into_device", 4, d_x, d_w, d_bias, d_y, batch, h, w, c, n, kh, kw, stride, pad, silu, ldy, d_res, ldr, stream);
}

int bf_conv1x1_cat_nhwc_f16_device(const void* d_x1, int ld1, int c1, int up1, const void* d_x2, int ld2, const void* d_w, const float* d_bias, void* d_y, int ldy,
                                   const void* d_res, int ldr, int batch, int h, int w, int c, int n, int silu, void* stream)
{
    return conv2d_checked("bf_conv1x1_cat_nhwc_f16_device", 2, d_x1, d_w, d_bias, d_y, batch, h, w, c, n, 1, 1, 1, 0, silu, ldy, d_res, ldr, stream, true, d_x2, c1, ld1,
                          ld2, up1);
}

int bf_conv1x1_cat_nhwc_f32_device(const void* d_x1, int ld1, int c1, int up1, const void* d_x2, int ld2, const void* d_w, const float* d_bias, void* d_y, int ldy,
                                   const void* d_res, int ldr, int batch, int h, int w, int c, int n, int silu, void* stream)
{
    return conv2d_checked("bf_conv1x1_cat_nhwc_f32_device", 4, d_x1, d_w, d_bias, d_y, batch, h, w, c, n, 1, 1, 1, 0, silu, ldy, d_res, ldr, stream, true, d_x2, c1, ld1,
                          ld2, up1);
}

int bf_nms_device(const float* d_boxes, const float* d_scores, const int* d_cls, const int* d_counts, int batch, int k, float iou_thres, int max_det,
                  unsigned long long* d_mask, float* d_out, int* d_out_count, void* stream)
{
    FD_ENTER(d_boxes && d_scores && d_cls && d_counts && d_mask && d_out && d_out_count && batch > 0 && k > 0 && max_det > 0, "bf_nms_device")
    if (k > 4096) { set_error("bf_nms_device: k = %d candidates; at most 4096", k); return -1; }
    return HIP_OK(bf::launch_nms(d_boxes, d_scores, d_cls, d_counts, batch, k, iou_thres, max_det, d_mask, d_out, d_out_count, st)) ? 0 : -1;
}

int bf_plan_das(int algo, int n, int frames, int dir_begin, int dir_end, int max_whole, int n_cus, long long out[10])
{
    State& s = S();
    std::lock_guard<std::mutex> lock(s.mu);
    sizes_from_env_once();
    bf::DasLaunch L{};
    L.algo = algo; L.n_mics = n; L.m_total = n; L.n_samples = s.sz.n_samples; L.n_taps = s.sz.n_taps; L.n_dirs = s.sz.dirs();
    L.force_layout = -1;
    L.dir_begin = dir_begin; L.dir_end = dir_end; L.frames = frames; L.tab.max_whole = max_whole;
    bf::DasPlan p{};
    const char* why = "";
    if (bf::plan_das(L, n_cus > 0 ? n_cus : 256, &p, &why) != 0) { set_error("bf_plan_das: %s", why); return -1; }
    out[0] = p.nc; out[1] = p.lead; out[2] = p.row_stride; out[3] = p.mic_chunk; out[4] = p.n_chunks;
    out[5] = p.waves; out[6] = p.dpw; out[7] = p.tile_dirs; out[8] = p.n_tiles; out[9] = (long long)p.lds_bytes;
    return 0;
}

}  // extern "C"
