// Context, error text, stage / one-off timing, rocFFT plans, options.
// (part of the C ABI of libpsa_hip.so, include/psa_hip.h; shared declarations: api_internal.h)
#include "api_internal.h"

#include <unistd.h>

#include <filesystem>

namespace psa {

thread_local std::string g_error;


void set_error(const char* fmt, ...) {
    char    buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return PSA_OK;
    if (ptr) {
        PSA_HIP_CHECK(hipFree(ptr));
        ptr = nullptr;
        cap = 0;
    }
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e != hipSuccess) {
        ptr = nullptr;
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return PSA_ENOMEM;
    }
    cap = bytes;
    return PSA_OK;
}

void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
}

int enter(psa_ctx* c) {
    PSA_REQUIRE(c != nullptr, "null context");
    PSA_HIP_CHECK(hipSetDevice(c->device));
    return PSA_OK;
}

// ---- stage timing: event pairs on the context's stream ---------------------
TimingState& timing(psa_ctx* c) { return c->timing; }

int get_event(TimingState& ts, hipEvent_t* ev) {
    if (!ts.pool.empty()) {
        *ev = ts.pool.back();
        ts.pool.pop_back();
        return PSA_OK;
    }
    PSA_HIP_CHECK(hipEventCreate(ev));
    return PSA_OK;
}

int collect(psa_ctx* c, TimingState& ts) {
    if (ts.pending.empty()) return PSA_OK;
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (auto& p : ts.pending) {
        float ms = 0.f;
        PSA_HIP_CHECK(hipEventElapsedTime(&ms, p.e0, p.e1));
        ts.acc[p.stage] += ms;
        if (p.stage == PSA_T_PROJECT) {
            ts.k1_launches += 1;
            ts.k1_ms += ms;
        }
        ts.pool.push_back(p.e0);
        ts.pool.push_back(p.e1);
    }
    ts.pending.clear();
    return PSA_OK;
}

int upload(psa_ctx* c, DevBuf& b, const void* host, size_t bytes) {
    PSA_TRY(b.reserve(bytes ? bytes : 16));
    if (bytes) PSA_HIP_CHECK(hipMemcpyAsync(b.ptr, host, bytes, hipMemcpyHostToDevice, c->stream));
    return PSA_OK;
}

static int create_plan(psa_ctx* c, int64_t T, int64_t batch, FftPlan* p) {
    size_t len = (size_t)T;
    PSA_FFT_CHECK(rocfft_plan_create(&p->plan, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                                     rocfft_precision_single, 1, &len, (size_t)batch, nullptr));
    PSA_FFT_CHECK(rocfft_plan_get_work_buffer_size(p->plan, &p->work_bytes));
    PSA_FFT_CHECK(rocfft_execution_info_create(&p->info));
    PSA_FFT_CHECK(rocfft_execution_info_set_stream(p->info, c->stream));
    return PSA_OK;
}

// the primer's plan (T, 3) -- one k-vector -- joins the cache; its kernels serve every batch of that length
static void join_primer(psa_ctx* c) {
    if (!c->fft_primer.joinable()) return;
    HostTimer ht(&c->oneoff_ms[0]);                    // whatever of the build is still outstanding
    c->fft_primer.join();
    if (c->primed.plan) {
        auto key = std::make_pair(c->primed_T, (int64_t)3);
        if (c->plans.find(key) == c->plans.end()) {
            c->plans.emplace(key, c->primed);
        } else {
            (void)rocfft_execution_info_destroy(c->primed.info);
            (void)rocfft_plan_destroy(c->primed.plan);
        }
        c->primed = FftPlan{};
    }
}

void prime_fft(psa_ctx* c, int64_t T) {
    if (!c->opt_fft_prime || c->fft_primer.joinable() || T < 2 || c->primed_T == T) return;
    for (auto& kv : c->plans)
        if (kv.first.first == T) return;
    c->primed_T = T;
    c->fft_primer = std::thread([c, T] {
        (void)hipSetDevice(c->device);
        FftPlan p;
        if (create_plan(c, T, 3, &p) == PSA_OK) c->primed = p;
    });
}

int get_plan(psa_ctx* c, int64_t T, int64_t batch, FftPlan** out) {
    join_primer(c);
    auto key = std::make_pair(T, batch);
    auto it = c->plans.find(key);
    if (it == c->plans.end()) {
        HostTimer ht(&c->oneoff_ms[0]);
        FftPlan   p;
        PSA_TRY(create_plan(c, T, batch, &p));
        it = c->plans.emplace(key, p).first;
    }
    *out = &it->second;
    return PSA_OK;
}

int run_fft(psa_ctx* c, float2* data, int64_t T, int64_t batch) {
    FftPlan* p = nullptr;
    PSA_TRY(get_plan(c, T, batch, &p));
    if (p->work_bytes) {
        PSA_TRY(c->d_fft_work.reserve(p->work_bytes));
        PSA_FFT_CHECK(rocfft_execution_info_set_work_buffer(p->info, c->d_fft_work.ptr, p->work_bytes));
    }
    void* bufs[1] = {data};
    PSA_FFT_CHECK(rocfft_execute(p->plan, bufs, nullptr, p->info));
    return PSA_OK;
}

int check_slot(psa_ctx* c, int slot) {
    PSA_REQUIRE(slot >= 0 && slot < PSA_NUM_SLOTS, "bad data slot %d", slot);
    PSA_REQUIRE(c->slot[slot].valid, "data slot %d holds no array", slot);
    return PSA_OK;
}

int validate_groups(int64_t N, const int32_t* group_idx, const int64_t* group_off, int32_t G) {
    PSA_REQUIRE(G >= 1, "need at least one atom group");
    if (!group_idx) {
        PSA_REQUIRE(G == 1, "group_idx NULL means one group of all atoms (G must be 1)");
        return PSA_OK;
    }
    PSA_REQUIRE(group_off != nullptr, "group_off is NULL");
    PSA_REQUIRE(group_off[0] == 0, "group_off[0] must be 0");
    for (int g = 0; g < G; ++g)
        PSA_REQUIRE(group_off[g + 1] >= group_off[g], "group_off must be non-decreasing");
    for (int64_t i = 0; i < group_off[G]; ++i)
        PSA_REQUIRE(group_idx[i] >= 0 && group_idx[i] < N, "Atom indices in basis out of bounds.");
    return PSA_OK;
}

}  // namespace psa

using namespace psa;

extern "C" {

int psa_abi_version(void) { return PSA_HIP_ABI_VERSION; }

const char* psa_last_error(void) { return g_error.c_str(); }

int psa_device_count(int* count) {
    PSA_REQUIRE(count != nullptr, "null count");
    PSA_HIP_CHECK(hipGetDeviceCount(count));
    return PSA_OK;
}

int psa_host_alloc(size_t bytes, void** out) {
    PSA_REQUIRE(out != nullptr && bytes > 0, "bad argument");
    PSA_HIP_CHECK(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return PSA_OK;
}

int psa_host_free(void* p) {
    if (p) PSA_HIP_CHECK(hipHostFree(p));
    return PSA_OK;
}

int psa_create(int device, psa_ctx** out) {
    PSA_REQUIRE(out != nullptr, "null out");
    int n = 0;
    PSA_HIP_CHECK(hipGetDeviceCount(&n));
    PSA_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    PSA_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    PSA_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    PSA_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "libpsa_hip is built for gfx950 (MI355X) only; device %d is %s", device,
                prop.gcnArchName);
    static std::once_flag fft_once;
    std::call_once(fft_once, [] {
        // rocFFT keeps run-time compiled kernels in a cache file only when it is told where: give it a
        // per-user one (a second process then skips the compilation: 58 -> 23 ms for T = 65536,
        // tools/fft_plan_timing.py); an explicit ROCFFT_RTC_CACHE_PATH wins
        if (!std::getenv("ROCFFT_RTC_CACHE_PATH")) {
            std::string dir;
            if (const char* e = std::getenv("PSA_CACHE_DIR")) dir = e;
            else if (const char* x = std::getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/psa_amd";
            else if (const char* h = std::getenv("HOME")) dir = std::string(h) + "/.cache/psa_amd";
            if (!dir.empty()) {
                std::error_code ec;
                std::filesystem::create_directories(dir, ec);
                if (!ec && ::access(dir.c_str(), W_OK) == 0)
                    ::setenv("ROCFFT_RTC_CACHE_PATH", (dir + "/rocfft_rtc_cache.db").c_str(), 0);
            }
        }
        (void)rocfft_setup();
    });
    psa_ctx* c = new psa_ctx();
    c->device = device;
    c->compute_units = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        return PSA_EHIP;
    }
    *out = c;
    return PSA_OK;
}

int psa_destroy(psa_ctx* c) {
    if (!c) return PSA_OK;
    {
        Guard g(c);
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->fft_primer.joinable()) c->fft_primer.join();
        if (c->primed.plan) {
            (void)rocfft_execution_info_destroy(c->primed.info);
            (void)rocfft_plan_destroy(c->primed.plan);
        }
        if (c->comm) (void)ncclCommDestroy(c->comm);
        for (auto& kv : c->plans) {
            (void)rocfft_execution_info_destroy(kv.second.info);
            (void)rocfft_plan_destroy(kv.second.plan);
        }
        for (auto& p : c->timing.pending) {
            (void)hipEventDestroy(p.e0);
            (void)hipEventDestroy(p.e1);
        }
        for (auto ev : c->timing.pool) (void)hipEventDestroy(ev);
        for (auto& s : c->slot) s.buf.release();
        for (auto& ps : c->planes) ps->buf.release();
        c->planes.clear();
        stager_release(c);
        if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
        if (c->d2h_ready) (void)hipEventDestroy(c->d2h_ready);
        for (DevBuf* b : {&c->d_kvec, &c->d_mean_all, &c->d_idx, &c->d_mean_g, &c->d_phase, &c->d_qwork,
                          &c->d_fft_work, &c->d_tables, &c->d_absmax, &c->d_slab, &c->d_out, &c->d_aux, &c->d_sync,
                          &c->d_qrows, &c->d_stage, &c->d_bin, &c->d_upload_max, &c->d_zeros, &c->d_kmap, &c->d_cols, &c->d_inten})
            b->release();
        (void)hipStreamDestroy(c->stream);
    }
    delete c;
    return PSA_OK;
}

int psa_synchronize(psa_ctx* c) {
    PSA_TRY(enter(c));
    Guard g(c);
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_set_k1(psa_ctx* c, int selector) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(selector == PSA_K1_AUTO || selector == PSA_K1_WAVE || selector == PSA_K1_MFMA32 ||
                    selector == PSA_K1_SPLIT_BF16,
                "unknown K1 selector %d", selector);
    Guard g(c);
    c->k1_selector = selector;
    return PSA_OK;
}

int psa_set_option(psa_ctx* c, int option, int64_t value) {
    PSA_TRY(enter(c));
    Guard g(c);
    switch (option) {
        case PSA_OPT_PLANES:
            c->opt_planes = value != 0;
            if (!c->opt_planes) {
                PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
                for (auto& ps : c->planes) ps->buf.release();
                c->planes.clear();
            }
            return PSA_OK;
        case PSA_OPT_PLANES_BUDGET:
            PSA_REQUIRE(value >= 0, "negative plane budget");
            c->opt_planes_budget = value;
            return PSA_OK;
        case PSA_OPT_PLANES_EAGER: c->opt_planes_eager = value != 0; return PSA_OK;
        case PSA_OPT_PLANES_MIN_K:
            PSA_REQUIRE(value >= 1, "PSA_OPT_PLANES_MIN_K must be >= 1");
            c->opt_planes_min_k = value;
            return PSA_OK;
        case PSA_OPT_FOLD_PAIRS: c->opt_fold_pairs = value != 0; return PSA_OK;
        case PSA_OPT_FFT_PRIME: c->opt_fft_prime = value != 0; return PSA_OK;
        case PSA_OPT_K1_LOADER_WAVES: c->opt_k1_loader_waves = value != 0; return PSA_OK;
        case PSA_OPT_K1_WIDE: c->opt_k1_wide = value != 0; return PSA_OK;
    }
    set_error("unknown option %d", option);
    return PSA_EINVAL;
}

int psa_device_info(psa_ctx* c, char* name, int name_len, int* compute_units, int64_t* hbm_bytes) {
    PSA_TRY(enter(c));
    hipDeviceProp_t prop;
    PSA_HIP_CHECK(hipGetDeviceProperties(&prop, c->device));
    if (name && name_len > 0) {
        std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name[0] ? prop.name : "AMD GPU",
                      prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return PSA_OK;
}

int psa_last_timings(psa_ctx* c, double* ms) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(ms != nullptr, "null output");
    Guard guard(c);
    TimingState& ts = timing(c);
    PSA_TRY(collect(c, ts));
    for (int i = 0; i < PSA_T_COUNT; ++i) {
        ms[i] = ts.acc[i];
        ts.acc[i] = 0.0;
    }
    return PSA_OK;
}

int psa_oneoff_stats(psa_ctx* c, double* ms) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(ms != nullptr, "null output");
    Guard guard(c);
    for (int i = 0; i < 4; ++i) {
        ms[i] = c->oneoff_ms[i];
        c->oneoff_ms[i] = 0.0;
    }
    return PSA_OK;
}

int psa_k1_stats(psa_ctx* c, int64_t* launches, double* total_ms) {
    PSA_TRY(enter(c));
    Guard guard(c);
    TimingState& ts = timing(c);
    PSA_TRY(collect(c, ts));
    if (launches) *launches = ts.k1_launches;
    if (total_ms) *total_ms = ts.k1_ms;
    ts.k1_launches = 0;
    ts.k1_ms = 0.0;
    return PSA_OK;
}

}  // extern "C"
