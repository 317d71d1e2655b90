// C ABI of libpsa_hip.so (declared in include/psa_hip.h).  In file order:
//   context, error text, stage / one-off timing, rocFFT plans, magnitude passes
//   the plane cache (get_planes), projection geometry (make_geom), phase table + projection launch
//   the host -> device staging pipeline (CopyPool, staged_upload)
//   extern "C": context and options, trajectory residency, synthetic fill, mean, displacements
//   the hot path: psa_sed_project / _project_upload / _finalize / _calculate (pipelined) / _single_bin
//   slab access, result intensity / chiral phase, timings, diagnostics
//   sharding over RCCL: communicator, gather (k rows), frame sharding (psa_sed_fs_*)
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "k1_f16.h"

namespace psa {

static thread_local std::string g_error;

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

namespace {

struct HostTimer {                                    // adds its lifetime to one of ctx->oneoff_ms
    double*                               into;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit HostTimer(double* into_) : into(into_) {}
    ~HostTimer() { *into += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

struct Guard {
    std::lock_guard<std::mutex> lk;
    explicit Guard(psa_ctx* c) : lk(c->mu) {}
};

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

struct StageTimer {
    psa_ctx*     c;
    TimingState& ts;
    int          stage;
    hipEvent_t   e0 = nullptr, e1 = nullptr;
    bool         ok = false;
    StageTimer(psa_ctx* c_, int stage_) : c(c_), ts(timing(c_)), stage(stage_) {
        if (get_event(ts, &e0) == PSA_OK && get_event(ts, &e1) == PSA_OK &&
            hipEventRecord(e0, c->stream) == hipSuccess)
            ok = true;
    }
    ~StageTimer() {
        if (ok && hipEventRecord(e1, c->stream) == hipSuccess) ts.pending.push_back({stage, e0, e1});
    }
};

int upload(psa_ctx* c, DevBuf& b, const void* host, size_t bytes) {
    PSA_TRY(b.reserve(bytes ? bytes : 16));
    if (bytes) PSA_HIP_CHECK(hipMemcpyAsync(b.ptr, host, bytes, hipMemcpyHostToDevice, c->stream));
    return PSA_OK;
}

int get_plan(psa_ctx* c, int64_t T, int64_t batch, FftPlan** out) {
    auto key = std::make_pair(T, batch);
    auto it = c->plans.find(key);
    if (it == c->plans.end()) {
        HostTimer ht(&c->oneoff_ms[0]);
        FftPlan   p;
        size_t  len = (size_t)T;
        PSA_FFT_CHECK(rocfft_plan_create(&p.plan, rocfft_placement_inplace,
                                         rocfft_transform_type_complex_forward,
                                         rocfft_precision_single, 1, &len, (size_t)batch, nullptr));
        PSA_FFT_CHECK(rocfft_plan_get_work_buffer_size(p.plan, &p.work_bytes));
        PSA_FFT_CHECK(rocfft_execution_info_create(&p.info));
        PSA_FFT_CHECK(rocfft_execution_info_set_stream(p.info, c->stream));
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

// largest magnitude of a resident array: one HBM pass + a 4-byte read-back, once per upload
int slot_absmax(psa_ctx* c, int slot) {
    DataSlot& s = c->slot[slot];
    if (s.absmax_known) return PSA_OK;
    HostTimer ht(&c->oneoff_ms[1]);
    PSA_TRY(c->d_absmax.reserve(sizeof(unsigned)));
    PSA_TRY(launch_absmax_bits(c, s.buf.as<float>(), s.T * s.N * 3, c->d_absmax.as<unsigned>()));
    PSA_HIP_CHECK(hipMemcpyAsync(&s.absmax_bits, c->d_absmax.ptr, sizeof(unsigned), hipMemcpyDeviceToHost,
                                 c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    s.absmax_known = true;
    return PSA_OK;
}

// largest magnitude among the atoms of an index list: the maximum over the 32-atom column blocks
// they sit in (one HBM pass per upload, then a host loop over the list)
int group_absmax(psa_ctx* c, int slot, const int32_t* h_idx, int64_t n_g, unsigned* bits) {
    DataSlot& s = c->slot[slot];
    if (!s.blocks_known) {
        HostTimer    ht(&c->oneoff_ms[1]);
        const size_t n_blocks = (size_t)((s.N + 31) / 32);
        PSA_TRY(c->d_absmax.reserve(n_blocks * sizeof(unsigned)));
        PSA_TRY(launch_absmax_blocks(c, s.buf.as<float>(), nullptr, s.T, s.N, c->d_absmax.as<unsigned>()));
        s.block_absmax.resize(n_blocks);
        PSA_HIP_CHECK(hipMemcpyAsync(s.block_absmax.data(), c->d_absmax.ptr, n_blocks * sizeof(unsigned),
                                     hipMemcpyDeviceToHost, c->stream));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
        s.blocks_known = true;
    }
    unsigned m = 0;
    for (int64_t i = 0; i < n_g; ++i) m = std::max(m, s.block_absmax[(size_t)h_idx[i] >> 5]);
    *bits = m;
    return PSA_OK;
}

// the same for slot - mean (displacement mode; the mean is in d_mean_all): per-block table cached for
// (slot contents, mean); h_idx null = all atoms
int displaced_absmax(psa_ctx* c, int slot, const float* mean_host, const int32_t* h_idx, int64_t n_g, unsigned* bits) {
    DataSlot&    s = c->slot[slot];
    const size_t n_mean = (size_t)s.N * 3, n_blocks = (size_t)((s.N + 31) / 32);
    const bool   fresh = c->disp_abs_source == s.generation && c->disp_abs_mean.size() == n_mean &&
                       c->disp_block_absmax.size() == n_blocks &&
                       std::memcmp(c->disp_abs_mean.data(), mean_host, n_mean * sizeof(float)) == 0;
    if (!fresh) {
        HostTimer ht(&c->oneoff_ms[1]);
        PSA_TRY(c->d_absmax.reserve(n_blocks * sizeof(unsigned)));
        PSA_TRY(launch_absmax_blocks(c, s.buf.as<float>(), c->d_mean_all.as<float>(), s.T, s.N, c->d_absmax.as<unsigned>()));
        c->disp_block_absmax.resize(n_blocks);
        PSA_HIP_CHECK(hipMemcpyAsync(c->disp_block_absmax.data(), c->d_absmax.ptr, n_blocks * sizeof(unsigned),
                                     hipMemcpyDeviceToHost, c->stream));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
        c->disp_abs_mean.assign(mean_host, mean_host + n_mean);
        c->disp_abs_source = s.generation;
    }
    unsigned m = 0;
    if (h_idx)
        for (int64_t i = 0; i < n_g; ++i) m = std::max(m, c->disp_block_absmax[(size_t)h_idx[i] >> 5]);
    else
        for (unsigned b : c->disp_block_absmax) m = std::max(m, b);
    *bits = m;
    return PSA_OK;
}

uint64_t hash_idx(const int32_t* p, int64_t n) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    for (int64_t i = 0; i < n; ++i) h = (h ^ (uint32_t)p[i]) * 1099511628211ull;
    return h;
}

size_t planes_bytes_held(psa_ctx* c) {
    size_t b = 0;
    for (auto& ps : c->planes) b += ps->buf.cap;
    return b;
}

// plane sets built from contents a slot no longer holds
void drop_stale_planes(psa_ctx* c) {
    auto& v = c->planes;
    v.erase(std::remove_if(v.begin(), v.end(),
                           [&](const std::unique_ptr<PlaneSet>& ps) {
                               const DataSlot& s = c->slot[ps->slot];
                               if (s.valid && s.generation == ps->generation) return false;
                               ps->buf.release();
                               return true;
                           }),
            v.end());
}

// least recently used set that the call in progress has not touched; false if there is none
bool evict_one_plane_set(psa_ctx* c) {
    int victim = -1;
    for (size_t i = 0; i < c->planes.size(); ++i)
        if (c->planes[i]->last_use < c->plane_call_mark &&
            (victim < 0 || c->planes[i]->last_use < c->planes[victim]->last_use))
            victim = (int)i;
    if (victim < 0) return false;
    c->planes[victim]->buf.release();
    c->planes.erase(c->planes.begin() + victim);
    return true;
}

// The group's split planes (k1_planes.hip): found in the cache, or built now if the policy
// (PSA_OPT_PLANES*) and HBM allow; *out stays nullptr otherwise and the caller projects with the
// kernels that split on the fly.  h_idx / d_idx: the group's index list on the host / device
// (nullptr: all atoms in order).
// mean_host non-null: planes of slot - mean (displacement mode; the mean is also in d_mean_all).
int get_planes(psa_ctx* c, int slot, const int* d_idx, const int32_t* h_idx, int64_t n_g, int64_t K_local,
               const float* mean_host, PlaneSet** out) {
    *out = nullptr;
    if (c->k1_selector != PSA_K1_AUTO || !c->opt_planes) return PSA_OK;
    DataSlot& s = c->slot[slot];
    drop_stale_planes(c);
    const bool     all = h_idx == nullptr, displaced = mean_host != nullptr;
    const uint64_t h = (all ? 0 : hash_idx(h_idx, n_g)) ^ (displaced ? 0x9E3779B97F4A7C15ull : 0);
    const size_t   n_mean = (size_t)s.N * 3;
    for (auto& ps : c->planes)
        if (ps->slot == slot && ps->all_atoms == all && ps->n_g == n_g && ps->displaced == displaced &&
            (all || (ps->idx_hash == h && std::memcmp(ps->idx.data(), h_idx, (size_t)n_g * sizeof(int32_t)) == 0)) &&
            (!displaced || (ps->mean.size() == n_mean && std::memcmp(ps->mean.data(), mean_host, n_mean * sizeof(float)) == 0))) {
            ps->last_use = ++c->plane_tick;
            *out = ps.get();
            return PSA_OK;
        }
    // nothing cached: short k-lists are not worth a set of their own (their "3 x bf16" kernel streams the
    // float32 array at the same HBM-bound rate: 4.17 vs 4.10 ms at 16 k-vectors) -- but they use one that exists
    if (K_local < c->opt_planes_min_k) return PSA_OK;
    if (!all && !c->opt_planes_eager) {              // an index list seen for the first time: not yet
        auto& seen = c->seen_groups;
        if (std::find(seen.begin(), seen.end(), h) == seen.end()) {
            seen.push_back(h);
            if (seen.size() > 256) seen.erase(seen.begin());
            return PSA_OK;
        }
    }
    unsigned bits = 0;
    if (displaced) {
        PSA_TRY(displaced_absmax(c, slot, mean_host, h_idx, n_g, &bits));
    } else if (all) {
        PSA_TRY(slot_absmax(c, slot));
        bits = s.absmax_bits;
    } else {
        PSA_TRY(group_absmax(c, slot, h_idx, n_g, &bits));
    }
    const float vscale = k1_f16_vscale(bits);
    if (!(vscale > 0.f)) return PSA_OK;              // NaN / Inf in the data: the bf16 kernel propagates them
    const int     A_pad = k1_pair_atom_pad(n_g);
    const int64_t n_fg = (s.T + 15) / 16;
    const size_t  bytes = plane_bytes(n_fg, A_pad / K1_BA);
    size_t        free_b = 0, total_b = 0;
    PSA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = c->opt_planes_budget > 0 ? (size_t)c->opt_planes_budget : (size_t)(0.45 * (double)total_b);
    if (bytes > budget) return PSA_OK;
    while (planes_bytes_held(c) + bytes > budget)
        if (!evict_one_plane_set(c)) return PSA_OK;
    const size_t reserve = (size_t)2 << 30;          // leave room for slabs, FFT work buffers, results
    while (free_b < bytes + reserve) {
        if (!evict_one_plane_set(c)) return PSA_OK;
        PSA_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    }
    auto ps = std::make_unique<PlaneSet>();
    if (ps->buf.reserve(bytes) != PSA_OK) {
        (void)hipGetLastError();
        return PSA_OK;
    }
    {
        HostTimer ht(&c->oneoff_ms[2]);                 // timed: the stream is drained once per set
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
        PSA_TRY(launch_split_planes(c, s.buf.as<float>(), displaced ? c->d_mean_all.as<float>() : nullptr, d_idx, ps->buf.ptr,
                                    s.T, s.N, (int)n_g, A_pad, vscale));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    ps->slot = slot;
    ps->generation = s.generation;
    ps->all_atoms = all;
    if (!all) ps->idx.assign(h_idx, h_idx + n_g);
    ps->idx_hash = h;
    ps->displaced = displaced;
    if (displaced) ps->mean.assign(mean_host, mean_host + n_mean);
    ps->T = s.T;
    ps->n_fg = n_fg;
    ps->n_g = (int)n_g;
    ps->A_pad = A_pad;
    ps->vscale = vscale;
    ps->last_use = ++c->plane_tick;
    *out = ps.get();
    c->planes.push_back(std::move(ps));
    return PSA_OK;
}

// h_idx: the group's index list on the host (nullptr: all atoms in order); ps: its split planes, if any
// force: 0 = the product rule; 3 = "3 x bf16" wherever it can serve (needs no scale: the streaming
// upload projects frames before the whole array has been seen); -1 = the float32 kernel
int make_geom(psa_ctx* c, int slot, int64_t K_local, int64_t n_g, const int* d_idx, const int32_t* h_idx,
              bool disp, const PlaneSet* ps, int force, ProjGeom* g) {
    g->T = c->slot[slot].T;
    g->q_stride = g->T;
    g->N_tot = c->slot[slot].N;
    PSA_REQUIRE(n_g < (1ll << 30) && K_local < (1ll << 29), "group or k-list too large");
    g->n_g = (int)n_g;
    g->A_pad = (int)((n_g + 31) / 32 * 32);
    g->K = (int)K_local;
    // product path: split-precision matrix-core kernels -- "2 x f16" from the group's cached planes,
    // or splitting on the fly (more than 16 k-vectors, no NaN/Inf), "3 x bf16" for every other
    // group; exact-fp32 MFMA kernel for displacement mode when no displacement array could be made
    g->split = 0;
    const bool autosel = c->k1_selector == PSA_K1_AUTO;
    if (ps && force == 0) {
        g->split = 4;
        g->vscale = ps->vscale;
        g->m_blk = k1_planes_block_rows((int)K_local);
        g->A_pad = ps->A_pad;
        g->M_pad = (int)((2 * K_local + g->m_blk - 1) / g->m_blk * g->m_blk);
        return PSA_OK;
    }
    if (force == 0 && autosel && k1_pair_eligible(d_idx, g->N_tot, n_g, K_local, disp)) {
        unsigned bits = 0;
        if (h_idx) {
            PSA_TRY(group_absmax(c, slot, h_idx, n_g, &bits));
        } else {
            PSA_TRY(slot_absmax(c, slot));
            bits = c->slot[slot].absmax_bits;
        }
        g->vscale = k1_f16_vscale(bits);
        if (g->vscale > 0.f) g->split = 2;
    }
    if (g->split == 0 && force >= 0 && (autosel || c->k1_selector == PSA_K1_SPLIT_BF16) &&
        k1_split_eligible(d_idx, g->N_tot, n_g, disp))
        g->split = 3;
    if (g->split == 2) {
        g->m_blk = k1_pair_block_rows((int)K_local);
        g->A_pad = k1_pair_atom_pad(n_g);
    } else {
        g->m_blk = g->split ? k1_split_block_rows((int)K_local) : k1_mfma_block_rows((int)K_local);
    }
    g->M_pad = (int)((2 * K_local + g->m_blk - 1) / g->m_blk * g->m_blk);
    return PSA_OK;
}

// phase table of one group in the image its projection kernel wants (+ the group's mean positions
// for the subtract-while-staging kernels)
int prepare_phase(psa_ctx* c, const int* d_idx, const ProjGeom& g, bool disp, int64_t k_first = 0) {
    const float* d_kvec = c->d_kvec.as<float>() + 3 * k_first;         // the launch's k-vectors within the uploaded list
    const bool f16 = g.split == 2 || g.split == 4, bf16 = g.split == 3;
    PSA_TRY(c->d_phase.reserve(f16    ? pf16_table_bytes(g.M_pad, g.A_pad)
                               : bf16 ? pb_table_bytes(g.M_pad, g.A_pad)
                                      : p_table_floats(g.M_pad, g.A_pad) * sizeof(float)));
    StageTimer st(c, PSA_T_PHASE);
    if (f16)
        PSA_TRY(launch_phase_table_f16(c, d_kvec, c->d_mean_all.as<float>(), d_idx, c->d_phase.ptr, g));
    else if (bf16)
        PSA_TRY(launch_phase_table_split(c, d_kvec, c->d_mean_all.as<float>(), d_idx, c->d_phase.ptr, g));
    else
        PSA_TRY(launch_phase_table(c, d_kvec, c->d_mean_all.as<float>(), d_idx, c->d_phase.as<float>(), g));
    if (disp) {
        PSA_TRY(c->d_mean_g.reserve((size_t)g.A_pad * 3 * sizeof(float)));
        PSA_TRY(launch_gather_mean(c, c->d_mean_all.as<float>(), d_idx, c->d_mean_g.as<float>(), g));
    }
    return PSA_OK;
}

// projection of frames [t_begin, t_begin + t_count) of one group into columns t_begin.. of q
// (K_local,3,q_stride); the phase table is in place
int launch_projection(psa_ctx* c, int slot, const int* d_idx, ProjGeom g, bool disp, const PlaneSet* ps, float2* d_q,
                      int64_t q_stride, int64_t t_begin, int64_t t_count) {
    const DataSlot& s = c->slot[slot];
    PSA_REQUIRE(t_begin >= 0 && t_count > 0 && t_begin + t_count <= s.T && q_stride >= t_begin + t_count,
                "frame range [%lld,%lld) outside the slot", (long long)t_begin, (long long)(t_begin + t_count));
    g.T = t_count;
    g.q_stride = q_stride;
    StageTimer   st(c, PSA_T_PROJECT);
    const float* d_v = s.buf.as<float>() + (size_t)t_begin * 3 * (size_t)s.N;
    d_q += t_begin;
    if (g.split == 4) {
        PSA_REQUIRE(ps != nullptr && t_begin % 16 == 0, "planes are cut in groups of 16 frames");
        const int64_t fg0 = t_begin / 16;
        const _Float16* pl = ps->buf.as<_Float16>() + (size_t)fg0 * (size_t)(ps->A_pad / K1_BA) * PL_STAGE_ELEMS;
        return launch_k1_planes(c, pl, c->d_phase.ptr, d_q, g, ps->n_fg - fg0);
    }
    if (g.split == 2) return launch_k1_pair(c, d_v, c->d_phase.ptr, d_idx, d_q, g);
    if (g.split == 3) return launch_k1_split(c, d_v, c->d_phase.ptr, d_idx, d_q, g);
    if (c->k1_selector == PSA_K1_WAVE)
        return launch_k1_wave(c, d_v, c->d_phase.as<float>(), d_idx, c->d_mean_g.as<float>(), d_q, g, disp);
    return launch_k1_mfma(c, d_v, c->d_phase.as<float>(), d_idx, c->d_mean_g.as<float>(), d_q, g, disp);
}

// phase table + projection of one group over all frames of the slot into q (K_local,3,T); no FFT
int project_group(psa_ctx* c, int slot, const int* d_idx, const ProjGeom& g, bool disp, const PlaneSet* ps, float2* d_q) {
    PSA_TRY(prepare_phase(c, d_idx, g, disp));
    return launch_projection(c, slot, d_idx, g, disp, ps, d_q, c->slot[slot].T, 0, c->slot[slot].T);
}

// ---- host -> device staging pipeline -------------------------------------------------------
// A few host threads copy the (pageable or memory-mapped) source into one of two page-locked
// buffers while hipMemcpyAsync drains the other over PCIe on a copy stream of its own.
class CopyPool {
    std::vector<std::thread> threads_;
    std::mutex               m_;
    std::condition_variable  go_, done_;
    const char*              src_ = nullptr;
    char*                    dst_ = nullptr;
    size_t                   bytes_ = 0;
    uint64_t                 gen_ = 0;
    int                      pending_ = 0;

    void worker(int i, int n) {
        uint64_t seen = 0;
        for (;;) {
            const char* src;
            char*       dst;
            size_t      bytes;
            {
                std::unique_lock<std::mutex> lk(m_);
                go_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                src = src_, dst = dst_, bytes = bytes_;
            }
            const size_t per = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
            const size_t lo = std::min(bytes, per * (size_t)i), hi = std::min(bytes, lo + per);
            if (hi > lo) std::memcpy(dst + lo, src + lo, hi - lo);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }

public:
    explicit CopyPool(int n) {
        for (int i = 0; i < n; ++i) threads_.emplace_back([this, i, n] { worker(i, n); });
        for (auto& t : threads_) t.detach();
    }
    void copy(void* dst, const void* src, size_t bytes) {
        std::unique_lock<std::mutex> lk(m_);
        src_ = (const char*)src, dst_ = (char*)dst, bytes_ = bytes;
        pending_ = (int)threads_.size();
        ++gen_;
        go_.notify_all();
        done_.wait(lk, [&] { return pending_ == 0; });
    }
};

CopyPool& copy_pool() {
    static CopyPool* pool = [] {                     // lives as long as the process: its threads sleep on a condition
        int n = 6;
        if (const char* e = std::getenv("PSA_UPLOAD_THREADS")) n = std::atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0) n = std::min(n, hw);
        return new CopyPool(std::max(1, n));
    }();
    return *pool;
}
std::mutex g_copy_pool_mutex;                        // one upload at a time feeds the pool

int stager_init(psa_ctx* c, size_t chunk_bytes) {
    Stager& st = c->stager;
    if (std::getenv("PSA_UPLOAD_NO_STAGING")) {                  // (tests: the path a locked-memory limit takes)
        set_error("page-locked staging disabled");
        return PSA_ENOMEM;
    }
    if (!st.copy_stream) PSA_HIP_CHECK(hipStreamCreateWithFlags(&st.copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i)
        if (!st.freed[i]) PSA_HIP_CHECK(hipEventCreateWithFlags(&st.freed[i], hipEventDisableTiming));
    if (st.cap < chunk_bytes) {
        for (int i = 0; i < 2; ++i) {
            if (st.pin[i]) PSA_HIP_CHECK(hipHostFree(st.pin[i]));
            st.pin[i] = nullptr;
        }
        st.cap = 0;
        for (int i = 0; i < 2; ++i) PSA_HIP_CHECK(hipHostMalloc(&st.pin[i], chunk_bytes, hipHostMallocDefault));
        st.cap = chunk_bytes;
    }
    return PSA_OK;
}

void stager_release(psa_ctx* c) {
    Stager& st = c->stager;
    for (int i = 0; i < 2; ++i) {
        if (st.pin[i]) (void)hipHostFree(st.pin[i]);
        if (st.freed[i]) (void)hipEventDestroy(st.freed[i]);
        st.pin[i] = nullptr, st.freed[i] = nullptr;
    }
    if (st.copy_stream) (void)hipStreamDestroy(st.copy_stream);
    st.copy_stream = nullptr;
    st.cap = 0;
}

// (T, N, 3) float32 rows of `host` into `dev`, in chunks of whole frames.  After a chunk's copy has
// been queued on the copy stream, on_chunk(first frame, frames, event) may queue work that waits
// for `event`.  Returns when every byte is on the device.
int staged_upload(psa_ctx* c, float* dev, const float* host, int64_t T, int64_t N,
                  const std::function<int(int64_t, int64_t, hipEvent_t)>& on_chunk) {
    const size_t row = (size_t)N * 3 * sizeof(float);
    size_t       target = (size_t)64 << 20;
    if (const char* e = std::getenv("PSA_UPLOAD_CHUNK_MIB")) target = (size_t)std::max(1, std::atoi(e)) << 20;
    int64_t frames = (int64_t)(target / row);
    if (frames >= 64) frames = frames / 64 * 64;               // whole projection tiles
    else if ((size_t)64 * row <= ((size_t)256 << 20)) frames = 64;
    frames = std::min(std::max<int64_t>(frames, 1), T);        // (very wide rows: fewer frames per chunk)
    if (stager_init(c, (size_t)frames * row) != PSA_OK) {
        // no page-locked memory to be had (locked-memory limit): plain copies from the pageable source,
        // the chunk callback still runs behind each of them
        (void)hipGetLastError();
        stager_release(c);
        HostTimer  ht(&c->oneoff_ms[3]);
        hipEvent_t ev = nullptr;
        PSA_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        int rc = PSA_OK;
        for (int64_t t0 = 0; t0 < T && rc == PSA_OK; t0 += frames) {
            const int64_t nt = std::min(frames, T - t0);
            if (hipMemcpyAsync((char*)dev + (size_t)t0 * row, (const char*)host + (size_t)t0 * row, (size_t)nt * row,
                               hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipEventRecord(ev, c->stream) != hipSuccess)
                rc = PSA_EHIP;
            else if (on_chunk)
                rc = on_chunk(t0, nt, ev);
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PSA_OK) rc = PSA_EHIP;
        (void)hipEventDestroy(ev);
        if (rc == PSA_EHIP) set_error("host -> device copy failed: %s", hipGetErrorString(hipGetLastError()));
        return rc;
    }
    Stager&                     st = c->stager;
    HostTimer                   ht(&c->oneoff_ms[3]);
    std::lock_guard<std::mutex> pool_lock(g_copy_pool_mutex);
    std::vector<hipEvent_t>     landed;
    int                         rc = PSA_OK;
    int64_t                     i = 0;
    for (int64_t t0 = 0; t0 < T && rc == PSA_OK; t0 += frames, ++i) {
        const int64_t nt = std::min(frames, T - t0);
        const int     b = (int)(i & 1);
        if (i >= 2 && hipEventSynchronize(st.freed[b]) != hipSuccess) rc = PSA_EHIP;
        if (rc != PSA_OK) break;
        copy_pool().copy(st.pin[b], (const char*)host + (size_t)t0 * row, (size_t)nt * row);
        if (hipMemcpyAsync((char*)dev + (size_t)t0 * row, st.pin[b], (size_t)nt * row, hipMemcpyHostToDevice,
                           st.copy_stream) != hipSuccess ||
            hipEventRecord(st.freed[b], st.copy_stream) != hipSuccess) {
            rc = PSA_EHIP;
            break;
        }
        if (on_chunk) rc = on_chunk(t0, nt, st.freed[b]);
    }
    if (hipStreamSynchronize(st.copy_stream) != hipSuccess && rc == PSA_OK) rc = PSA_EHIP;
    if (rc == PSA_EHIP) set_error("host -> device staging failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
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

}  // namespace
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
    std::call_once(fft_once, [] { (void)rocfft_setup(); });
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
                          &c->d_qrows, &c->d_stage, &c->d_bin, &c->d_upload_max})
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

// ---- trajectory residency ----------------------------------------------------
static int data_alloc_locked(psa_ctx* c, int slot, int64_t T, int64_t N) {
    PSA_REQUIRE(slot >= 0 && slot < PSA_NUM_SLOTS, "bad data slot %d", slot);
    PSA_REQUIRE(T > 0 && N > 0, "empty trajectory (T=%lld, N=%lld)", (long long)T, (long long)N);
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    DataSlot& s = c->slot[slot];
    s.valid = false;
    s.absmax_known = false;
    s.blocks_known = false;
    ++s.generation;
    // 1 KiB of zeros behind the array: the split projection kernels pad the atom axis (to 32 atoms,
    // 64 in k1_pair.hip) and read up to 63 atoms past the final row, multiplied by zero phase columns
    const size_t bytes = (size_t)T * N * 3 * sizeof(float);
    PSA_TRY(s.buf.reserve(bytes + 1024));
    PSA_HIP_CHECK(hipMemsetAsync((char*)s.buf.ptr + bytes, 0, 1024, c->stream));
    s.T = T;
    s.N = N;
    s.valid = true;
    drop_stale_planes(c);
    return PSA_OK;
}

int psa_data_alloc(psa_ctx* c, int slot, int64_t T, int64_t N) {
    PSA_TRY(enter(c));
    Guard g(c);
    return data_alloc_locked(c, slot, T, N);
}

int psa_data_upload(psa_ctx* c, int slot, const float* host, int64_t T, int64_t N) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(host != nullptr, "null host array");
    Guard g(c);
    PSA_TRY(data_alloc_locked(c, slot, T, N));
    c->slot[slot].valid = false;                       // until every frame has landed
    StageTimer st(c, PSA_T_H2D);
    PSA_TRY(staged_upload(c, c->slot[slot].buf.as<float>(), host, T, N, nullptr));
    c->slot[slot].valid = true;
    return PSA_OK;
}

int psa_data_download(psa_ctx* c, int slot, float* host, int64_t t0, int64_t nt) {
    PSA_TRY(enter(c));
    Guard g(c);
    PSA_TRY(check_slot(c, slot));
    PSA_REQUIRE(host != nullptr, "null host array");
    const DataSlot& s = c->slot[slot];
    PSA_REQUIRE(t0 >= 0 && nt >= 0 && t0 + nt <= s.T, "frame range [%lld,%lld) outside [0,%lld)",
                (long long)t0, (long long)(t0 + nt), (long long)s.T);
    const size_t row = (size_t)s.N * 3 * sizeof(float);
    PSA_HIP_CHECK(hipMemcpyAsync(host, (const char*)s.buf.ptr + (size_t)t0 * row, (size_t)nt * row,
                                 hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_data_release(psa_ctx* c, int slot) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(slot >= 0 && slot < PSA_NUM_SLOTS, "bad data slot %d", slot);
    Guard g(c);
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->slot[slot].buf.release();
    c->slot[slot].valid = false;
    c->slot[slot].T = c->slot[slot].N = 0;
    ++c->slot[slot].generation;
    if (slot == PSA_SLOT_POSITIONS) {                      // the displacements derived from it go too
        c->slot[PSA_NUM_SLOTS].buf.release();
        c->slot[PSA_NUM_SLOTS].valid = false;
        ++c->slot[PSA_NUM_SLOTS].generation;
    }
    drop_stale_planes(c);
    return PSA_OK;
}

int psa_data_shape(psa_ctx* c, int slot, int64_t* T, int64_t* N) {
    PSA_TRY(enter(c));
    Guard g(c);
    PSA_TRY(check_slot(c, slot));
    if (T) *T = c->slot[slot].T;
    if (N) *N = c->slot[slot].N;
    return PSA_OK;
}

int psa_data_fill_synthetic(psa_ctx* c, int slot, uint64_t seed, int64_t t_offset, int n_modes, const float* amp,
                            const int32_t* mode_comp, const float* ct, const float* st,
                            const float* ca, const float* sa) {
    PSA_TRY(enter(c));
    Guard g(c);
    PSA_TRY(check_slot(c, slot));
    PSA_REQUIRE(n_modes >= 0 && n_modes <= 16, "n_modes must be in [0,16]");
    PSA_REQUIRE(t_offset >= 0, "negative frame offset");
    if (n_modes > 0)
        PSA_REQUIRE(amp && mode_comp && ct && st && ca && sa, "null mode table");
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    c->slot[slot].absmax_known = false;
    c->slot[slot].blocks_known = false;
    ++c->slot[slot].generation;
    // one packed upload: amp | comp | ct | st | ca | sa
    const size_t nm = (size_t)n_modes;
    const size_t o_amp = 0, o_comp = o_amp + nm * 4, o_ct = o_comp + nm * 4, o_st = o_ct + nm * T * 4,
                 o_ca = o_st + nm * T * 4, o_sa = o_ca + nm * N * 4, total = o_sa + nm * N * 4;
    PSA_TRY(c->d_tables.reserve(total ? total : 16));
    char* base = (char*)c->d_tables.ptr;
    if (nm) {
        PSA_HIP_CHECK(hipMemcpyAsync(base + o_amp, amp, nm * 4, hipMemcpyHostToDevice, c->stream));
        PSA_HIP_CHECK(hipMemcpyAsync(base + o_comp, mode_comp, nm * 4, hipMemcpyHostToDevice, c->stream));
        PSA_HIP_CHECK(hipMemcpyAsync(base + o_ct, ct, nm * T * 4, hipMemcpyHostToDevice, c->stream));
        PSA_HIP_CHECK(hipMemcpyAsync(base + o_st, st, nm * T * 4, hipMemcpyHostToDevice, c->stream));
        PSA_HIP_CHECK(hipMemcpyAsync(base + o_ca, ca, nm * N * 4, hipMemcpyHostToDevice, c->stream));
        PSA_HIP_CHECK(hipMemcpyAsync(base + o_sa, sa, nm * N * 4, hipMemcpyHostToDevice, c->stream));
    }
    drop_stale_planes(c);
    PSA_TRY(launch_fill_synthetic(c, c->slot[slot].buf.as<float>(), T, N, seed, t_offset, n_modes,
                                  (const float*)(base + o_amp), (const int*)(base + o_comp),
                                  (const float*)(base + o_ct), (const float*)(base + o_st),
                                  (const float*)(base + o_ca), (const float*)(base + o_sa)));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_mean_positions(psa_ctx* c, int slot, float* mean_host) {
    PSA_TRY(enter(c));
    Guard g(c);
    PSA_TRY(check_slot(c, slot));
    PSA_REQUIRE(mean_host != nullptr, "null output");
    const DataSlot& s = c->slot[slot];
    PSA_TRY(c->d_mean_all.reserve((size_t)s.N * 3 * sizeof(float)));
    PSA_TRY(launch_mean_over_frames(c, s.buf.as<float>(), s.T, s.N, c->d_mean_all.as<float>()));
    PSA_HIP_CHECK(hipMemcpyAsync(mean_host, c->d_mean_all.ptr, (size_t)s.N * 3 * sizeof(float),
                                 hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

// Displacement mode on the fast kernels: positions - mean as an array of its own (what the reference
// builds as a temporary, sed_calculator.py:70-72), cached while positions and mean stay the same.
// *slot_io becomes the internal slot and *disp false; if HBM has no room for the second array the
// call proceeds with the subtract-while-staging float32 kernel.
int materialise_displacements(psa_ctx* c, int* slot_io, bool* disp, const float* mean_host) {
    if (!*disp || c->k1_selector == PSA_K1_MFMA32 || c->k1_selector == PSA_K1_WAVE) return PSA_OK;
    DataSlot&       src = c->slot[*slot_io];
    DataSlot&       dst = c->slot[PSA_NUM_SLOTS];
    const size_t    n_mean = (size_t)src.N * 3;
    const bool fresh = dst.valid && c->disp_source == src.generation && dst.T == src.T && dst.N == src.N &&
                       c->disp_mean.size() == n_mean &&
                       std::memcmp(c->disp_mean.data(), mean_host, n_mean * sizeof(float)) == 0;
    if (!fresh) {
        const size_t bytes = (size_t)src.T * src.N * 3 * sizeof(float);
        dst.valid = false;
        if (dst.buf.reserve(bytes + 1024) != PSA_OK) {          // no room: keep the float32 path
            (void)hipGetLastError();
            return PSA_OK;
        }
        PSA_HIP_CHECK(hipMemsetAsync((char*)dst.buf.ptr + bytes, 0, 1024, c->stream));
        PSA_TRY(launch_subtract_mean(c, src.buf.as<float>(), c->d_mean_all.as<float>(), dst.buf.as<float>(), src.T, src.N));
        dst.T = src.T;
        dst.N = src.N;
        dst.valid = true;
        dst.absmax_known = dst.blocks_known = false;
        ++dst.generation;
        c->disp_mean.assign(mean_host, mean_host + n_mean);
        c->disp_source = src.generation;
    }
    *slot_io = PSA_NUM_SLOTS;
    *disp = false;
    return PSA_OK;
}

// Where one group's data comes from.  In order: its cached split planes -- of the velocities, or of
// positions - mean built straight from the positions (no float32 displacement array) -- else the
// float32 slot, which in displacement mode is the materialised positions - mean array (or, when HBM
// has no room for it, the positions themselves with the subtract-while-staging kernel).
// *slot_io / *disp_io come in as the caller's slot and PSA_F_DISPLACEMENTS and go out as what the
// projection has to be launched with.
static int group_source(psa_ctx* c, int* slot_io, bool* disp_io, const float* mean_host, const int* d_idx,
                        const int32_t* h_idx, int64_t n_g, int64_t K, PlaneSet** ps) {
    *ps = nullptr;
    PSA_TRY(get_planes(c, *slot_io, d_idx, h_idx, n_g, K, *disp_io ? mean_host : nullptr, ps));
    if (*ps) {
        *disp_io = false;                                         // the planes already hold slot - mean
        return PSA_OK;
    }
    return materialise_displacements(c, slot_io, disp_io, mean_host);
}

// ---- the hot path ---------------------------------------------------------------
namespace {

struct ProjectArgs {
    int            slot;
    const float*   mean_pos_all;
    const float*   k_vectors;
    int64_t        K_local, K_total, k_offset;
    const int32_t* group_idx;
    const int64_t* group_off;
    int32_t        G, flags;
};

int check_project_args(psa_ctx* c, const ProjectArgs& a, int64_t N) {
    PSA_REQUIRE(a.mean_pos_all != nullptr, "null mean_pos_all");
    PSA_REQUIRE(a.K_local >= 0 && a.K_total >= 1 && a.k_offset >= 0 && a.k_offset + a.K_local <= a.K_total,
                "k range [%lld,%lld) outside [0,%lld)", (long long)a.k_offset, (long long)(a.k_offset + a.K_local),
                (long long)a.K_total);
    PSA_REQUIRE(a.K_local == 0 || a.k_vectors != nullptr, "null k_vectors");
    PSA_TRY(validate_groups(N, a.group_idx, a.group_off, a.G));
    PSA_REQUIRE((a.flags & PSA_F_INTENSITY) || a.G == 1, "complex output needs exactly one atom group (got %d)", a.G);
    (void)c;
    return PSA_OK;
}

// result slab (k-major) of a calculation over T frames; returns the rows of this call
int begin_result(psa_ctx* c, int64_t T, int64_t K_total, int64_t k_offset, bool intensity, char** rows, size_t* row_bytes) {
    *row_bytes = intensity ? (size_t)T * sizeof(float) : (size_t)T * 3 * sizeof(float2);
    PSA_TRY(c->d_slab.reserve(*row_bytes * (size_t)K_total));
    c->res_T = T;
    c->res_K = K_total;
    c->res_intensity = intensity;
    c->slab_valid = true;
    c->out_valid = false;
    c->plane_call_mark = c->plane_tick + 1;
    *rows = (char*)c->d_slab.ptr + *row_bytes * (size_t)k_offset;
    return PSA_OK;
}

int upload_project_inputs(psa_ctx* c, const ProjectArgs& a, int64_t N) {
    StageTimer st(c, PSA_T_H2D);
    PSA_TRY(upload(c, c->d_kvec, a.k_vectors, (size_t)a.K_local * 3 * sizeof(float)));
    PSA_TRY(upload(c, c->d_mean_all, a.mean_pos_all, (size_t)N * 3 * sizeof(float)));
    if (a.group_idx) PSA_TRY(upload(c, c->d_idx, a.group_idx, (size_t)a.group_off[a.G] * sizeof(int32_t)));
    return PSA_OK;
}

// groups [g_first, G) on the resident slot: project, FFT, epilogue
int project_groups(psa_ctx* c, const ProjectArgs& a, int slot_in, bool disp_in, int g_first, bool* first, char* rows,
                   float2* d_q) {
    const int64_t T = c->slot[slot_in].T, N = c->slot[slot_in].N;
    const bool    intensity = (a.flags & PSA_F_INTENSITY) != 0;
    for (int gi = g_first; gi < a.G; ++gi) {
        const int64_t n_g = a.group_idx ? (a.group_off[gi + 1] - a.group_off[gi]) : N;
        if (n_g == 0) continue;                                   // sed_calculator.py:64-65, 319-321
        const int*     d_idx = a.group_idx ? c->d_idx.as<int>() + a.group_off[gi] : nullptr;
        const int32_t* h_idx = a.group_idx ? a.group_idx + a.group_off[gi] : nullptr;
        PlaneSet*      ps = nullptr;
        int            slot = slot_in;
        bool           disp = disp_in;
        PSA_TRY(group_source(c, &slot, &disp, a.mean_pos_all, d_idx, h_idx, n_g, a.K_local, &ps));
        // the phase table holds 8 bytes per (k-vector, atom): very long k-lists (a 500 x 500 grid) are
        // projected in blocks whose table stays under 2 GiB (the reference chunks k for the same reason,
        // sed_calculator.py:268-272); ordinary lists are one block
        const int64_t per_k = 8 * ((n_g + 63) / 64 * 64);
        int64_t       table = (int64_t)2 << 30;
        if (const char* e = std::getenv("PSA_PHASE_TABLE_MIB")) table = (int64_t)std::max(1, std::atoi(e)) << 20;
        int64_t kb = std::max<int64_t>(64, (table / per_k) / 64 * 64);
        if (a.K_local <= kb + 64) kb = a.K_local;
        for (int64_t k0 = 0; k0 < a.K_local; k0 += kb) {
            const int64_t nk = std::min(kb, a.K_local - k0);
            ProjGeom      g;
            PSA_TRY(make_geom(c, slot, nk, n_g, d_idx, h_idx, disp, ps, 0, &g));
            PSA_TRY(prepare_phase(c, d_idx, g, disp, k0));
            PSA_TRY(launch_projection(c, slot, d_idx, g, disp, ps, d_q + (size_t)k0 * 3 * (size_t)T, T, 0, T));
        }
        {
            StageTimer st(c, PSA_T_FFT);
            PSA_TRY(run_fft(c, d_q, T, 3 * a.K_local));
        }
        if (intensity) {
            StageTimer st(c, PSA_T_EPILOGUE);
            PSA_TRY(launch_intensity_accumulate(c, d_q, (float*)rows, T, a.K_local, *first));
        }
        *first = false;
    }
    return PSA_OK;
}

}  // namespace

int psa_sed_project(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                    int64_t K_local, int64_t K_total, int64_t k_offset, const int32_t* group_idx,
                    const int64_t* group_off, int32_t G, int32_t flags) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(check_slot(c, slot));
    const ProjectArgs a{slot, mean_pos_all, k_vectors, K_local, K_total, k_offset, group_idx, group_off, G, flags};
    const int64_t     T = c->slot[slot].T, N = c->slot[slot].N;
    const bool        intensity = (flags & PSA_F_INTENSITY) != 0;
    bool              disp = (flags & PSA_F_DISPLACEMENTS) != 0;
    PSA_TRY(check_project_args(c, a, N));
    char*  rows = nullptr;
    size_t row_bytes = 0;
    PSA_TRY(begin_result(c, T, K_total, k_offset, intensity, &rows, &row_bytes));
    if (K_local == 0) return PSA_OK;
    PSA_TRY(upload_project_inputs(c, a, N));

    float2* d_q = intensity ? nullptr : (float2*)rows;
    if (intensity) {
        PSA_TRY(c->d_qwork.reserve((size_t)K_local * 3 * T * sizeof(float2)));
        d_q = c->d_qwork.as<float2>();
    }
    bool first = true;
    PSA_TRY(project_groups(c, a, slot, disp, 0, &first, rows, d_q));
    if (first)   // every group empty: the rows are zero
        PSA_HIP_CHECK(hipMemsetAsync(rows, 0, row_bytes * (size_t)K_local, c->stream));
    return PSA_OK;
}

// Upload and project, overlapped (psa_hip.h).  The first non-empty group is projected chunk by
// chunk behind the copies, with a kernel that needs nothing from frames not yet seen: "3 x bf16"
// (no scale), or the float32 kernel that subtracts the mean while staging in displacement mode.
int psa_sed_project_upload(psa_ctx* c, int slot, const float* host, int64_t T, int64_t N, const float* mean_pos_all,
                           const float* k_vectors, int64_t K, const int32_t* group_idx, const int64_t* group_off,
                           int32_t G, int32_t flags) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(host != nullptr, "null host array");
    PSA_REQUIRE(K >= 1, "need at least one k-vector");
    Guard             guard(c);
    const ProjectArgs a{slot, mean_pos_all, k_vectors, K, K, 0, group_idx, group_off, G, flags};
    PSA_REQUIRE(slot >= 0 && slot < PSA_NUM_SLOTS, "bad data slot %d", slot);
    PSA_REQUIRE(T > 0 && N > 0, "empty trajectory (T=%lld, N=%lld)", (long long)T, (long long)N);
    PSA_TRY(check_project_args(c, a, N));
    PSA_TRY(data_alloc_locked(c, slot, T, N));
    c->slot[slot].valid = false;
    const bool intensity = (flags & PSA_F_INTENSITY) != 0;
    const bool disp = (flags & PSA_F_DISPLACEMENTS) != 0;
    char*      rows = nullptr;
    size_t     row_bytes = 0;
    PSA_TRY(begin_result(c, T, K, 0, intensity, &rows, &row_bytes));
    PSA_TRY(upload_project_inputs(c, a, N));
    float2* d_q = intensity ? nullptr : (float2*)rows;
    if (intensity) {
        PSA_TRY(c->d_qwork.reserve((size_t)K * 3 * T * sizeof(float2)));
        d_q = c->d_qwork.as<float2>();
    }
    // the rocFFT plan (run-time compiled on first use of a length) is built beside the upload
    int         plan_rc = PSA_OK;
    std::string plan_err;
    std::thread planner([&] {
        (void)hipSetDevice(c->device);
        FftPlan* p = nullptr;
        plan_rc = get_plan(c, T, 3 * K, &p);
        if (plan_rc != PSA_OK) plan_err = g_error;
    });
    struct JoinOnExit {                                           // no path leaves with the thread running
        std::thread& t;
        ~JoinOnExit() {
            if (t.joinable()) t.join();
        }
    } join_planner{planner};
    int g0 = 0;                                                   // first non-empty group
    while (g0 < G && group_idx && group_off[g0 + 1] == group_off[g0]) ++g0;
    int rc = PSA_OK;
    // the array's largest magnitude (scale of the f16 kernels on later calls) is folded chunk by chunk
    // behind the copies too: no extra pass over the array after the upload
    PSA_TRY(c->d_upload_max.reserve(sizeof(unsigned)));
    PSA_HIP_CHECK(hipMemsetAsync(c->d_upload_max.ptr, 0, sizeof(unsigned), c->stream));
    const size_t row_floats = (size_t)N * 3;
    auto fold_max = [&](int64_t t0, int64_t nt) {
        return launch_absmax_bits(c, c->slot[slot].buf.as<float>() + (size_t)t0 * row_floats, nt * (int64_t)row_floats,
                                  c->d_upload_max.as<unsigned>(), false);
    };
    if (g0 < G) {
        const int64_t  n_g = group_idx ? (group_off[g0 + 1] - group_off[g0]) : N;
        const int*     d_idx = group_idx ? c->d_idx.as<int>() + group_off[g0] : nullptr;
        const int32_t* h_idx = group_idx ? group_idx + group_off[g0] : nullptr;
        ProjGeom       g;
        rc = make_geom(c, slot, K, n_g, d_idx, h_idx, disp, nullptr, disp ? -1 : 3, &g);
        if (rc == PSA_OK) rc = prepare_phase(c, d_idx, g, disp);
        if (rc == PSA_OK) {
            StageTimer st(c, PSA_T_H2D);
            rc = staged_upload(c, c->slot[slot].buf.as<float>(), host, T, N,
                               [&](int64_t t0, int64_t nt, hipEvent_t landed) -> int {
                                   PSA_HIP_CHECK(hipStreamWaitEvent(c->stream, landed, 0));
                                   PSA_TRY(fold_max(t0, nt));
                                   return launch_projection(c, slot, d_idx, g, disp, nullptr, d_q, T, t0, nt);
                               });
        }
    } else {
        StageTimer st(c, PSA_T_H2D);
        rc = staged_upload(c, c->slot[slot].buf.as<float>(), host, T, N,
                           [&](int64_t t0, int64_t nt, hipEvent_t landed) -> int {
                               PSA_HIP_CHECK(hipStreamWaitEvent(c->stream, landed, 0));
                               return fold_max(t0, nt);
                           });
    }
    planner.join();
    if (rc == PSA_OK && plan_rc != PSA_OK) {
        g_error = plan_err;
        rc = plan_rc;
    }
    PSA_TRY(rc);
    c->slot[slot].valid = true;
    PSA_HIP_CHECK(hipMemcpyAsync(&c->slot[slot].absmax_bits, c->d_upload_max.ptr, sizeof(unsigned), hipMemcpyDeviceToHost,
                                 c->stream));
    bool first = true;
    if (g0 < G) {
        {
            StageTimer st(c, PSA_T_FFT);
            PSA_TRY(run_fft(c, d_q, T, 3 * K));
        }
        if (intensity) {
            StageTimer st(c, PSA_T_EPILOGUE);
            PSA_TRY(launch_intensity_accumulate(c, d_q, (float*)rows, T, K, true));
        }
        first = false;
        // remaining groups on the now resident array, by the ordinary rule
        if (g0 + 1 < G) PSA_TRY(project_groups(c, a, slot, disp, g0 + 1, &first, rows, d_q));
    }
    if (first) PSA_HIP_CHECK(hipMemsetAsync(rows, 0, row_bytes * (size_t)K, c->stream));
    if (!c->slot[slot].absmax_known) {                           // (a later group's geometry may have asked already)
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));            // the read-back above has landed
        c->slot[slot].absmax_known = true;
    }
    return PSA_OK;
}

static size_t result_bytes(const psa_ctx* c) {
    return c->res_intensity ? (size_t)c->res_T * c->res_K * sizeof(float) : (size_t)c->res_T * c->res_K * 3 * sizeof(float2);
}

int psa_sed_finalize(psa_ctx* c, void* out_host, size_t out_bytes) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->slab_valid) {
        set_error("psa_sed_finalize before psa_sed_project");
        return PSA_ESTATE;
    }
    const int64_t T = c->res_T, K = c->res_K;
    const size_t  bytes = result_bytes(c);
    PSA_REQUIRE(out_host == nullptr || out_bytes == bytes,
                "result is %zu bytes (T=%lld, K=%lld, %s), the caller's buffer %zu", bytes, (long long)T, (long long)K,
                c->res_intensity ? "float32 intensity" : "complex64 x 3", out_bytes);
    PSA_TRY(c->d_out.reserve(bytes));
    {
        StageTimer st(c, PSA_T_TRANSPOSE);
        if (c->res_intensity)
            PSA_TRY(launch_transpose_f32(c, c->d_slab.as<float>(), c->d_out.as<float>(), T, K));
        else
            PSA_TRY(launch_scale_transpose_c64(c, c->d_slab.as<float2>(), c->d_out.as<float2>(), T, K, K, 0));
    }
    c->out_valid = true;
    if (out_host) {
        StageTimer st(c, PSA_T_D2H);
        PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_out.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return PSA_OK;
}

// Blocks of k-vectors when a complex result is produced block by block so that the D2H copy of one
// block runs while the next is projected.  r = (D2H time per k-vector) / (projection time per
// k-vector) = (24 T / 55 GB/s) / (N_g T / 4.3e13 units/s) = 1.9e4 / N_g decides the shape:
//   r < 1  (projection-bound, e.g. configuration 3): what stays exposed is the LAST block's copy ->
//          one large block (efficient projection) and a last block of one 64-vector M block;
//   r >= 1 (copy-bound, e.g. the 2500-point grid on 8192 atoms): what stays exposed is the FIRST
//          block's projection -> a first block of 128, then blocks of 512.
// Lists shorter than 192 are not split (every block is at least one M block of 64).
static std::vector<int64_t> pipeline_blocks(int64_t K, int64_t n_g) {
    std::vector<int64_t> b;
    if (K < 192) {
        b.push_back(K);
    } else if (1.9e4 / (double)std::max<int64_t>(n_g, 1) < 1.0) {
        b.push_back(K - 64);
        b.push_back(64);
    } else {
        b.push_back(128);
        for (int64_t k0 = 128; k0 < K; k0 += 512) b.push_back(std::min<int64_t>(512, K - k0));
        if (b.back() < 64 && b.size() > 2) {             // fold a sliver into its neighbour
            b[b.size() - 2] += b.back();
            b.pop_back();
        }
    }
    return b;
}

// Complex result of one group, all K on this device, straight to the host: per block of k-vectors
// project -> FFT -> scale/transpose into its columns of (T, K, 3) -> 2-D D2H on a copy stream
// (full PCIe rate at >= 1.5-KB rows: tools/probes/d2h_2d.hip), overlapped with the next block.
static int calculate_pipelined(psa_ctx* c, const ProjectArgs& a, void* out_host) {
    int       slot = a.slot;
    const int64_t K = a.K_total;
    PSA_TRY(check_slot(c, slot));
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    bool          disp = (a.flags & PSA_F_DISPLACEMENTS) != 0;
    PSA_TRY(check_project_args(c, a, N));
    char*  rows = nullptr;
    size_t row_bytes = 0;
    PSA_TRY(begin_result(c, T, K, 0, false, &rows, &row_bytes));
    PSA_TRY(upload_project_inputs(c, a, N));
    PSA_TRY(c->d_out.reserve(result_bytes(c)));
    if (!c->d2h_stream) PSA_HIP_CHECK(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    if (!c->d2h_ready) PSA_HIP_CHECK(hipEventCreateWithFlags(&c->d2h_ready, hipEventDisableTiming));
    const int64_t  n_g = a.group_idx ? (a.group_off[1] - a.group_off[0]) : N;
    const int*     d_idx = a.group_idx ? c->d_idx.as<int>() : nullptr;
    const int32_t* h_idx = a.group_idx ? a.group_idx : nullptr;
    if (n_g == 0) {
        std::memset(out_host, 0, result_bytes(c));
        PSA_HIP_CHECK(hipMemsetAsync(c->d_out.ptr, 0, result_bytes(c), c->stream));
        PSA_HIP_CHECK(hipMemsetAsync(rows, 0, row_bytes * (size_t)K, c->stream));
        c->out_valid = true;
        return PSA_OK;
    }
    PlaneSet* ps = nullptr;
    PSA_TRY(group_source(c, &slot, &disp, a.mean_pos_all, d_idx, h_idx, n_g, K, &ps));
    const size_t pitch = (size_t)K * 3 * sizeof(float2);
    int64_t      k0 = 0;
    for (const int64_t nk : pipeline_blocks(K, n_g)) {
        float2* d_q = (float2*)(rows + row_bytes * (size_t)k0);
        ProjGeom      g;
        PSA_TRY(make_geom(c, slot, nk, n_g, d_idx, h_idx, disp, ps, 0, &g));
        PSA_TRY(prepare_phase(c, d_idx, g, disp, k0));
        PSA_TRY(launch_projection(c, slot, d_idx, g, disp, ps, d_q, T, 0, T));
        {
            StageTimer st(c, PSA_T_FFT);
            PSA_TRY(run_fft(c, d_q, T, 3 * nk));
        }
        {
            StageTimer st(c, PSA_T_TRANSPOSE);
            PSA_TRY(launch_scale_transpose_c64(c, d_q, c->d_out.as<float2>(), T, nk, K, k0));
        }
        PSA_HIP_CHECK(hipEventRecord(c->d2h_ready, c->stream));
        PSA_HIP_CHECK(hipStreamWaitEvent(c->d2h_stream, c->d2h_ready, 0));
        const size_t off = (size_t)k0 * 3 * sizeof(float2), width = (size_t)nk * 3 * sizeof(float2);
        PSA_HIP_CHECK(hipMemcpy2DAsync((char*)out_host + off, pitch, (const char*)c->d_out.ptr + off, pitch, width, (size_t)T,
                                       hipMemcpyDeviceToHost, c->d2h_stream));
        k0 += nk;
    }
    c->out_valid = true;
    PSA_HIP_CHECK(hipStreamSynchronize(c->d2h_stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_sed_calculate(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                      int64_t K, const int32_t* group_idx, const int64_t* group_off, int32_t G,
                      int32_t flags, void* out_host, size_t out_bytes) {
    PSA_REQUIRE(K >= 1, "need at least one k-vector");
    if (out_host && !(flags & PSA_F_INTENSITY) && G == 1 && K >= 192 && c && c->k1_selector == PSA_K1_AUTO) {
        PSA_TRY(enter(c));
        Guard guard(c);
        PSA_TRY(check_slot(c, slot));
        const size_t bytes = (size_t)c->slot[slot].T * K * 3 * sizeof(float2);
        PSA_REQUIRE(out_bytes == bytes, "result is %zu bytes (T=%lld, K=%lld, complex64 x 3), the caller's buffer %zu", bytes,
                    (long long)c->slot[slot].T, (long long)K, out_bytes);
        const ProjectArgs a{slot, mean_pos_all, k_vectors, K, K, 0, group_idx, group_off, G, flags};
        return calculate_pipelined(c, a, out_host);
    }
    PSA_TRY(psa_sed_project(c, slot, mean_pos_all, k_vectors, K, K, 0, group_idx, group_off, G, flags));
    return psa_sed_finalize(c, out_host, out_bytes);
}

// one (k, omega) bin of one group: K = 1 projection + one DFT dot (psa_hip.h)
int psa_sed_single_bin(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vector, const int32_t* idx,
                       int64_t n_g, int32_t flags, int64_t i_w, float* out_c64x3) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(check_slot(c, slot));
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    PSA_REQUIRE(mean_pos_all && k_vector && out_c64x3, "null argument");
    PSA_REQUIRE(i_w >= 0 && i_w < T, "frequency bin %lld outside [0,%lld)", (long long)i_w, (long long)T);
    if (idx) {
        PSA_REQUIRE(n_g >= 0, "negative group size");
        for (int64_t i = 0; i < n_g; ++i)
            PSA_REQUIRE(idx[i] >= 0 && idx[i] < N, "Atom indices in basis out of bounds.");
    } else {
        n_g = N;
    }
    if (n_g == 0) {
        std::memset(out_c64x3, 0, 6 * sizeof(float));
        return PSA_OK;
    }
    bool disp = (flags & PSA_F_DISPLACEMENTS) != 0;
    PSA_TRY(upload(c, c->d_kvec, k_vector, 3 * sizeof(float)));
    PSA_TRY(upload(c, c->d_mean_all, mean_pos_all, (size_t)N * 3 * sizeof(float)));
    if (idx) PSA_TRY(upload(c, c->d_idx, idx, (size_t)n_g * sizeof(int32_t)));
    const int* d_idx = idx ? c->d_idx.as<int>() : nullptr;
    c->plane_call_mark = c->plane_tick + 1;
    PlaneSet* ps = nullptr;
    PSA_TRY(group_source(c, &slot, &disp, mean_pos_all, d_idx, idx, n_g, 1, &ps));
    ProjGeom g;
    PSA_TRY(make_geom(c, slot, 1, n_g, d_idx, idx, disp, ps, 0, &g));
    PSA_TRY(c->d_qwork.reserve((size_t)3 * T * sizeof(float2)));
    PSA_TRY(project_group(c, slot, d_idx, g, disp, ps, c->d_qwork.as<float2>()));
    PSA_TRY(c->d_bin.reserve(3 * sizeof(float2)));
    {
        StageTimer st(c, PSA_T_FFT);
        PSA_TRY(launch_dft_bin(c, c->d_qwork.as<float2>(), T, i_w, c->d_bin.as<float2>()));
    }
    PSA_HIP_CHECK(hipMemcpyAsync(out_c64x3, c->d_bin.ptr, 3 * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

static int slab_rows(psa_ctx* c, int64_t row0, int64_t nrows, size_t* off, size_t* bytes) {
    if (!c->slab_valid) {
        set_error("no slab: call psa_sed_project first");
        return PSA_ESTATE;
    }
    PSA_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= c->res_K, "slab rows [%lld,%lld) outside [0,%lld)",
                (long long)row0, (long long)(row0 + nrows), (long long)c->res_K);
    const size_t row_bytes = c->res_intensity ? (size_t)c->res_T * sizeof(float)
                                              : (size_t)c->res_T * 3 * sizeof(float2);
    *off = row_bytes * (size_t)row0;
    *bytes = row_bytes * (size_t)nrows;
    return PSA_OK;
}

int psa_slab_read(psa_ctx* c, int64_t row0, int64_t nrows, void* host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    size_t off = 0, bytes = 0;
    PSA_TRY(slab_rows(c, row0, nrows, &off, &bytes));
    PSA_REQUIRE(host != nullptr || bytes == 0, "null host buffer");
    if (bytes)
        PSA_HIP_CHECK(hipMemcpyAsync(host, (const char*)c->d_slab.ptr + off, bytes, hipMemcpyDeviceToHost,
                                     c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_slab_write(psa_ctx* c, int64_t row0, int64_t nrows, const void* host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    size_t off = 0, bytes = 0;
    PSA_TRY(slab_rows(c, row0, nrows, &off, &bytes));
    PSA_REQUIRE(host != nullptr || bytes == 0, "null host buffer");
    if (bytes)
        PSA_HIP_CHECK(hipMemcpyAsync((char*)c->d_slab.ptr + off, host, bytes, hipMemcpyHostToDevice,
                                     c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->out_valid = false;
    return PSA_OK;
}

int psa_result_intensity(psa_ctx* c, float* out_host, size_t out_bytes) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->out_valid || c->res_intensity) {
        set_error("psa_result_intensity needs a finalized complex result");
        return PSA_ESTATE;
    }
    const int64_t n = c->res_T * c->res_K;
    PSA_REQUIRE(out_host == nullptr || out_bytes == (size_t)n * sizeof(float),
                "result is (%lld,%lld) float32 = %zu bytes, the caller's buffer %zu", (long long)c->res_T,
                (long long)c->res_K, (size_t)n * sizeof(float), out_bytes);
    PSA_TRY(c->d_aux.reserve((size_t)n * sizeof(float)));
    {
        StageTimer st(c, PSA_T_EPILOGUE);
        PSA_TRY(launch_result_intensity(c, c->d_out.as<float2>(), c->d_aux.as<float>(), n));
    }
    if (out_host) {
        PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_aux.ptr, (size_t)n * sizeof(float),
                                     hipMemcpyDeviceToHost, c->stream));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return PSA_OK;
}

int psa_result_chiral_phase(psa_ctx* c, int c1, int c2, float* out_host, size_t out_bytes) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->out_valid || c->res_intensity) {
        set_error("psa_result_chiral_phase needs a finalized complex result");
        return PSA_ESTATE;
    }
    PSA_REQUIRE(c1 >= 0 && c1 < 3 && c2 >= 0 && c2 < 3, "component indices must be 0..2");
    const int64_t n = c->res_T * c->res_K;
    PSA_REQUIRE(out_host == nullptr || out_bytes == (size_t)n * sizeof(float),
                "result is (%lld,%lld) float32 = %zu bytes, the caller's buffer %zu", (long long)c->res_T,
                (long long)c->res_K, (size_t)n * sizeof(float), out_bytes);
    PSA_TRY(c->d_aux.reserve((size_t)n * sizeof(float)));
    PSA_TRY(launch_result_chiral_c(c, c->d_out.as<float2>(), c->d_aux.as<float>(), n, c1, c2));
    if (out_host) {
        PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_aux.ptr, (size_t)n * sizeof(float),
                                     hipMemcpyDeviceToHost, c->stream));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
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

// ---- diagnostics ------------------------------------------------------------------
int psa_debug_phase_table(psa_ctx* c, const float* mean_pos_all, const float* k_vectors, int64_t K,
                          const int32_t* idx, int64_t n_g, int64_t N, void* out_host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_REQUIRE(mean_pos_all && k_vectors && out_host && K >= 1 && n_g >= 1 && N >= 1, "bad argument");
    if (idx)
        for (int64_t i = 0; i < n_g; ++i)
            PSA_REQUIRE(idx[i] >= 0 && idx[i] < N, "Atom indices in basis out of bounds.");
    else
        PSA_REQUIRE(n_g == N, "identity group must cover all atoms");
    ProjGeom g;
    g.n_g = (int)n_g;
    g.A_pad = (int)((n_g + 31) / 32 * 32);
    g.K = (int)K;
    g.m_blk = 32;
    g.M_pad = (int)((2 * K + 31) / 32 * 32);
    PSA_TRY(upload(c, c->d_kvec, k_vectors, (size_t)K * 3 * sizeof(float)));
    PSA_TRY(upload(c, c->d_mean_all, mean_pos_all, (size_t)N * 3 * sizeof(float)));
    if (idx) PSA_TRY(upload(c, c->d_idx, idx, (size_t)n_g * sizeof(int32_t)));
    PSA_TRY(c->d_phase.reserve(p_table_floats(g.M_pad, g.A_pad) * sizeof(float)));
    PSA_TRY(launch_phase_table(c, c->d_kvec.as<float>(), c->d_mean_all.as<float>(),
                               idx ? c->d_idx.as<int>() : nullptr, c->d_phase.as<float>(), g));
    std::vector<float> P(p_table_floats(g.M_pad, g.A_pad));
    PSA_HIP_CHECK(hipMemcpyAsync(P.data(), c->d_phase.ptr, P.size() * sizeof(float),
                                 hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    float* o = (float*)out_host;
    for (int64_t k = 0; k < K; ++k)
        for (int64_t a = 0; a < n_g; ++a) {
            o[2 * (k * n_g + a) + 0] = P[p_tile_index((int)(2 * k), (int)a, g.m_blk, g.A_pad / K1_BA)];
            o[2 * (k * n_g + a) + 1] = P[p_tile_index((int)(2 * k + 1), (int)a, g.m_blk, g.A_pad / K1_BA)];
        }
    return PSA_OK;
}

static int debug_project(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors, int64_t K,
                         const int32_t* idx, int64_t n_g, int32_t flags, int64_t t_begin, int64_t t_count, void* out_host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(check_slot(c, slot));
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    PSA_REQUIRE(mean_pos_all && k_vectors && out_host && K >= 1 && n_g >= 1, "bad argument");
    if (t_count < 0) t_begin = 0, t_count = T;
    if (idx)
        for (int64_t i = 0; i < n_g; ++i)
            PSA_REQUIRE(idx[i] >= 0 && idx[i] < N, "Atom indices in basis out of bounds.");
    else
        PSA_REQUIRE(n_g == N, "identity group must cover all atoms");
    PSA_TRY(upload(c, c->d_kvec, k_vectors, (size_t)K * 3 * sizeof(float)));
    PSA_TRY(upload(c, c->d_mean_all, mean_pos_all, (size_t)N * 3 * sizeof(float)));
    if (idx) PSA_TRY(upload(c, c->d_idx, idx, (size_t)n_g * sizeof(int32_t)));
    ProjGeom g;
    bool disp = (flags & PSA_F_DISPLACEMENTS) != 0;
    const int* d_idx = idx ? c->d_idx.as<int>() : nullptr;
    c->plane_call_mark = c->plane_tick + 1;
    PlaneSet* ps = nullptr;
    PSA_TRY(group_source(c, &slot, &disp, mean_pos_all, d_idx, idx, n_g, K, &ps));
    PSA_TRY(make_geom(c, slot, K, n_g, d_idx, idx, disp, ps, 0, &g));
    const size_t bytes = (size_t)K * 3 * T * sizeof(float2);
    PSA_TRY(c->d_qwork.reserve(bytes));
    if (t_count != T) PSA_HIP_CHECK(hipMemsetAsync(c->d_qwork.ptr, 0, bytes, c->stream));
    PSA_TRY(prepare_phase(c, d_idx, g, disp));
    if (t_count > 0) PSA_TRY(launch_projection(c, slot, d_idx, g, disp, ps, c->d_qwork.as<float2>(), T, t_begin, t_count));
    PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_qwork.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_debug_project_only(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                           int64_t K, const int32_t* idx, int64_t n_g, int32_t flags, void* out_host) {
    return debug_project(c, slot, mean_pos_all, k_vectors, K, idx, n_g, flags, 0, -1, out_host);
}

int psa_debug_project_frames(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors, int64_t K,
                             const int32_t* idx, int64_t n_g, int32_t flags, int64_t t_begin, int64_t t_count,
                             void* out_host) {
    PSA_REQUIRE(t_begin >= 0 && t_count >= 0, "negative frame range");
    return debug_project(c, slot, mean_pos_all, k_vectors, K, idx, n_g, flags, t_begin, t_count, out_host);
}

int psa_debug_plane_cache(psa_ctx* c, int64_t* n_sets, int64_t* bytes) {
    PSA_TRY(enter(c));
    Guard guard(c);
    drop_stale_planes(c);
    if (n_sets) *n_sets = (int64_t)c->planes.size();
    if (bytes) *bytes = (int64_t)planes_bytes_held(c);
    return PSA_OK;
}

// ---- k-point sharding over RCCL ------------------------------------------------------
int psa_comm_unique_id(void* out) {
    PSA_REQUIRE(out != nullptr, "null output");
    static_assert(sizeof(ncclUniqueId) <= PSA_UNIQUE_ID_BYTES, "ncclUniqueId larger than the ABI slot");
    ncclUniqueId id;
    PSA_NCCL_CHECK(ncclGetUniqueId(&id));
    std::memset(out, 0, PSA_UNIQUE_ID_BYTES);
    std::memcpy(out, &id, sizeof(id));
    return PSA_OK;
}

int psa_comm_init(psa_ctx* c, const void* unique_id, int rank, int nranks) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(unique_id != nullptr && nranks >= 1 && rank >= 0 && rank < nranks, "bad rank/nranks");
    Guard guard(c);
    if (c->comm) {
        PSA_NCCL_CHECK(ncclCommDestroy(c->comm));
        c->comm = nullptr;
    }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    PSA_NCCL_CHECK(ncclCommInitRank(&c->comm, nranks, id, rank));
    c->rank = rank;
    c->nranks = nranks;
    return PSA_OK;
}

int psa_comm_destroy(psa_ctx* c) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (c->comm) {
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
        PSA_NCCL_CHECK(ncclCommDestroy(c->comm));
        c->comm = nullptr;
    }
    c->rank = 0;
    c->nranks = 1;
    return PSA_OK;
}

int psa_sed_gather(psa_ctx* c, int root, const int64_t* k_offsets, const int64_t* k_counts) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->slab_valid) {
        set_error("psa_sed_gather before psa_sed_project");
        return PSA_ESTATE;
    }
    if (c->nranks == 1) return PSA_OK;
    PSA_REQUIRE(c->comm != nullptr, "no communicator: call psa_comm_init first");
    PSA_REQUIRE(root >= -1 && root < c->nranks && k_offsets && k_counts, "bad gather arguments");
    const size_t row_floats = c->res_intensity ? (size_t)c->res_T : (size_t)c->res_T * 6;
    for (int r = 0; r < c->nranks; ++r)
        PSA_REQUIRE(k_offsets[r] >= 0 && k_counts[r] >= 0 && k_offsets[r] + k_counts[r] <= c->res_K,
                    "rank %d row range outside the slab", r);
    StageTimer st(c, PSA_T_GATHER);
    float* slab = c->d_slab.as<float>();
    const int me = c->rank;
    // direct peer-to-peer exchange: every transfer rides its own xGMI link, no ring.  A failing
    // send/recv must not leave the group open: the loop stops, the group is closed, then we report.
    PSA_NCCL_CHECK(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int r = 0; r < c->nranks && bad == ncclSuccess; ++r) {
        if (r == me) continue;
        const bool i_receive = (root < 0 || root == me) && k_counts[r] > 0;
        const bool i_send = (root < 0 || root == r) && k_counts[me] > 0;
        if (i_receive)
            bad = ncclRecv(slab + row_floats * (size_t)k_offsets[r], row_floats * (size_t)k_counts[r], ncclFloat, r, c->comm,
                           c->stream);
        if (i_send && bad == ncclSuccess)
            bad = ncclSend(slab + row_floats * (size_t)k_offsets[me], row_floats * (size_t)k_counts[me], ncclFloat, r,
                           c->comm, c->stream);
    }
    const ncclResult_t closed = ncclGroupEnd();
    PSA_NCCL_CHECK(bad);
    PSA_NCCL_CHECK(closed);
    return PSA_OK;
}

// ---- frame sharding (psa_hip.h) ----------------------------------------------------------
namespace {

// my rows of the group in flight, (fs_rows_nk, 3, fs_T_total) complex64: the slab rows themselves
// for complex output, a work buffer when |.|^2 is accumulated over groups
float2* fs_my_rows(psa_ctx* c) {
    return c->fs_intensity ? c->d_qrows.as<float2>() : c->d_slab.as<float2>() + (size_t)c->fs_rows_k0 * 3 * c->fs_T_total;
}

int fs_check(psa_ctx* c) {
    if (c->fs_T_total <= 0) {
        set_error("no frame-sharded projection in flight: call psa_sed_fs_project first");
        return PSA_ESTATE;
    }
    return PSA_OK;
}

// columns [t0, t0 + nt) of my rows <- a contiguous (rows, nt) block
int fs_place(psa_ctx* c, const void* src, int64_t t0, int64_t nt, hipMemcpyKind kind) {
    if (nt == 0 || c->fs_rows_nk == 0) return PSA_OK;
    PSA_HIP_CHECK(hipMemcpy2DAsync(fs_my_rows(c) + t0, (size_t)c->fs_T_total * sizeof(float2), src, (size_t)nt * sizeof(float2),
                                   (size_t)nt * sizeof(float2), (size_t)c->fs_rows_nk * 3, kind, c->stream));
    return PSA_OK;
}

}  // namespace

int psa_sed_fs_project(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors, int64_t K_total,
                       const int32_t* idx, int64_t n_g, int32_t flags, int64_t T_total, int64_t k_offset, int64_t k_count) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(check_slot(c, slot));
    const int64_t T_local = c->slot[slot].T, N = c->slot[slot].N;
    const bool    intensity = (flags & PSA_F_INTENSITY) != 0;
    bool          disp = (flags & PSA_F_DISPLACEMENTS) != 0;
    PSA_REQUIRE(mean_pos_all && k_vectors && K_total >= 1, "bad argument");
    PSA_REQUIRE(T_total >= T_local, "the slot holds %lld frames of a %lld-frame trajectory?", (long long)T_local,
                (long long)T_total);
    PSA_REQUIRE(k_offset >= 0 && k_count >= 0 && k_offset + k_count <= K_total, "k rows [%lld,%lld) outside [0,%lld)",
                (long long)k_offset, (long long)(k_offset + k_count), (long long)K_total);
    if (idx) {
        for (int64_t i = 0; i < n_g; ++i)
            PSA_REQUIRE(idx[i] >= 0 && idx[i] < N, "Atom indices in basis out of bounds.");
    } else {
        n_g = N;
    }
    char*  rows = nullptr;
    size_t row_bytes = 0;
    PSA_TRY(begin_result(c, T_total, K_total, k_offset, intensity, &rows, &row_bytes));
    c->fs_T_total = T_total;
    c->fs_K_total = K_total;
    c->fs_rows_k0 = k_offset;
    c->fs_rows_nk = k_count;
    c->fs_intensity = intensity;
    c->fs_T_local = T_local;
    if (intensity) PSA_TRY(c->d_qrows.reserve((size_t)std::max<int64_t>(k_count, 1) * 3 * T_total * sizeof(float2)));
    PSA_TRY(c->d_qwork.reserve((size_t)K_total * 3 * T_local * sizeof(float2)));
    if (n_g == 0) {                                            // an empty group projects to zero
        PSA_HIP_CHECK(hipMemsetAsync(c->d_qwork.ptr, 0, (size_t)K_total * 3 * T_local * sizeof(float2), c->stream));
        return PSA_OK;
    }
    PSA_TRY(upload(c, c->d_kvec, k_vectors, (size_t)K_total * 3 * sizeof(float)));
    PSA_TRY(upload(c, c->d_mean_all, mean_pos_all, (size_t)N * 3 * sizeof(float)));
    if (idx) PSA_TRY(upload(c, c->d_idx, idx, (size_t)n_g * sizeof(int32_t)));
    const int* d_idx = idx ? c->d_idx.as<int>() : nullptr;
    PlaneSet*  ps = nullptr;
    PSA_TRY(group_source(c, &slot, &disp, mean_pos_all, d_idx, idx, n_g, K_total, &ps));
    ProjGeom g;
    PSA_TRY(make_geom(c, slot, K_total, n_g, d_idx, idx, disp, ps, 0, &g));
    return project_group(c, slot, d_idx, g, disp, ps, c->d_qwork.as<float2>());
}

int psa_sed_fs_exchange(psa_ctx* c, const int64_t* t_offsets, const int64_t* t_counts, const int64_t* k_offsets,
                        const int64_t* k_counts) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(fs_check(c));
    PSA_REQUIRE(t_offsets && t_counts && k_offsets && k_counts, "null range table");
    const int     me = c->rank, n = c->nranks;
    const int64_t T_local = c->fs_T_local;
    int64_t       t_sum = 0;
    for (int r = 0; r < n; ++r) {
        PSA_REQUIRE(t_offsets[r] >= 0 && t_counts[r] >= 0 && t_offsets[r] + t_counts[r] <= c->fs_T_total &&
                        k_offsets[r] >= 0 && k_counts[r] >= 0 && k_offsets[r] + k_counts[r] <= c->fs_K_total,
                    "rank %d: frame or row range outside the calculation", r);
        t_sum += t_counts[r];
    }
    PSA_REQUIRE(t_sum == c->fs_T_total && t_counts[me] == T_local, "frame ranges do not tile the trajectory");
    PSA_REQUIRE(k_offsets[me] == c->fs_rows_k0 && k_counts[me] == c->fs_rows_nk, "row range differs from psa_sed_fs_project's");
    PSA_REQUIRE(n == 1 || c->comm != nullptr, "no communicator: call psa_comm_init first");
    StageTimer    st(c, PSA_T_GATHER);
    const size_t  my_rows = (size_t)c->fs_rows_nk * 3;
    const float2* q = c->d_qwork.as<float2>();
    if (n > 1) {
        PSA_TRY(c->d_stage.reserve(std::max<size_t>(16, my_rows * (size_t)(c->fs_T_total - T_local) * sizeof(float2))));
        // every pair of ranks trades one block over its own link: my frames of your rows for your
        // frames of my rows
        PSA_NCCL_CHECK(ncclGroupStart());
        ncclResult_t bad = ncclSuccess;
        size_t       land = 0;
        for (int r = 0; r < n && bad == ncclSuccess; ++r) {
            if (r == me) continue;
            const size_t in = my_rows * (size_t)t_counts[r], out = (size_t)k_counts[r] * 3 * (size_t)T_local;
            if (in) bad = ncclRecv(c->d_stage.as<float2>() + land, 2 * in, ncclFloat, r, c->comm, c->stream);
            if (out && bad == ncclSuccess)
                bad = ncclSend(q + (size_t)k_offsets[r] * 3 * (size_t)T_local, 2 * out, ncclFloat, r, c->comm, c->stream);
            land += in;
        }
        const ncclResult_t closed = ncclGroupEnd();
        PSA_NCCL_CHECK(bad);
        PSA_NCCL_CHECK(closed);
        land = 0;
        for (int r = 0; r < n; ++r) {
            if (r == me) continue;
            PSA_TRY(fs_place(c, c->d_stage.as<float2>() + land, t_offsets[r], t_counts[r], hipMemcpyDeviceToDevice));
            land += my_rows * (size_t)t_counts[r];
        }
    }
    return fs_place(c, q + (size_t)c->fs_rows_k0 * 3 * (size_t)T_local, t_offsets[me], T_local, hipMemcpyDeviceToDevice);
}

int psa_sed_fs_read(psa_ctx* c, int64_t k0, int64_t nk, void* host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(fs_check(c));
    PSA_REQUIRE(k0 >= 0 && nk >= 0 && k0 + nk <= c->fs_K_total && (host || nk == 0), "bad row range");
    const size_t row = (size_t)3 * c->fs_T_local * sizeof(float2);
    if (nk)
        PSA_HIP_CHECK(hipMemcpyAsync(host, (const char*)c->d_qwork.ptr + row * (size_t)k0, row * (size_t)nk,
                                     hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_sed_fs_write(psa_ctx* c, int64_t t0, int64_t nt, const void* host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(fs_check(c));
    PSA_REQUIRE(t0 >= 0 && nt >= 0 && t0 + nt <= c->fs_T_total && (host || nt == 0), "bad frame range");
    PSA_TRY(fs_place(c, host, t0, nt, hipMemcpyHostToDevice));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

int psa_sed_fs_finish(psa_ctx* c, int32_t first_group) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(fs_check(c));
    if (c->fs_rows_nk == 0) return PSA_OK;
    {
        StageTimer st(c, PSA_T_FFT);
        PSA_TRY(run_fft(c, fs_my_rows(c), c->fs_T_total, 3 * c->fs_rows_nk));
    }
    if (c->fs_intensity) {
        StageTimer st(c, PSA_T_EPILOGUE);
        PSA_TRY(launch_intensity_accumulate(c, c->d_qrows.as<float2>(),
                                            c->d_slab.as<float>() + (size_t)c->fs_rows_k0 * c->fs_T_total, c->fs_T_total,
                                            c->fs_rows_nk, first_group != 0));
    }
    return PSA_OK;
}

// One small grouped point-to-point round in the pattern psa_sed_gather / psa_sed_fs_exchange use
// (every pair of ranks trades a stamped block), checked on arrival: run once after psa_comm_init so
// that a communicator that formed but cannot move data is found before a calculation depends on it.
int psa_comm_selftest(psa_ctx* c) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (c->nranks == 1 || !c->comm) return PSA_OK;
    const int           n = c->nranks, me = c->rank, words = 256;
    std::vector<float>  out((size_t)n * words), in((size_t)n * words, -1.f);
    for (int r = 0; r < n; ++r)
        for (int i = 0; i < words; ++i) out[(size_t)r * words + i] = (float)(me * 1000 + r) + 0.001f * (float)i;
    PSA_TRY(c->d_stage.reserve(2 * out.size() * sizeof(float)));
    float* d_out = c->d_stage.as<float>();
    float* d_in = d_out + out.size();
    PSA_HIP_CHECK(hipMemcpyAsync(d_out, out.data(), out.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PSA_HIP_CHECK(hipMemsetAsync(d_in, 0xff, in.size() * sizeof(float), c->stream));
    PSA_NCCL_CHECK(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int r = 0; r < n && bad == ncclSuccess; ++r) {
        if (r == me) continue;
        bad = ncclRecv(d_in + (size_t)r * words, words, ncclFloat, r, c->comm, c->stream);
        if (bad == ncclSuccess) bad = ncclSend(d_out + (size_t)r * words, words, ncclFloat, r, c->comm, c->stream);
    }
    const ncclResult_t closed = ncclGroupEnd();
    PSA_NCCL_CHECK(bad);
    PSA_NCCL_CHECK(closed);
    PSA_HIP_CHECK(hipMemcpyAsync(in.data(), d_in, in.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int r = 0; r < n; ++r) {
        if (r == me) continue;
        for (int i = 0; i < words; ++i)
            if (in[(size_t)r * words + i] != (float)(r * 1000 + me) + 0.001f * (float)i) {
                set_error("RCCL self-test: block from rank %d arrived damaged (word %d)", r, i);
                return PSA_ERCCL;
            }
    }
    return PSA_OK;
}

int psa_comm_barrier(psa_ctx* c) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (c->nranks == 1 || !c->comm) {
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
        return PSA_OK;
    }
    PSA_TRY(c->d_sync.reserve(sizeof(float)));
    PSA_NCCL_CHECK(ncclAllReduce(c->d_sync.ptr, c->d_sync.ptr, 1, ncclFloat, ncclSum, c->comm, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

}  // extern "C"
