// C ABI of libpsa_hip.so (declared in include/psa_hip.h): context, trajectory residency,
// the project -> FFT -> epilogue pipeline, k-shard gather over RCCL.
#include <algorithm>
#include <cstring>

#include "psa_ctx.h"

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
        FftPlan p;
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
        const size_t n_blocks = (size_t)((s.N + 31) / 32);
        PSA_TRY(c->d_absmax.reserve(n_blocks * sizeof(unsigned)));
        PSA_TRY(launch_absmax_blocks(c, s.buf.as<float>(), s.T, s.N, c->d_absmax.as<unsigned>()));
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

// h_idx: the group's index list on the host (nullptr: all atoms in order)
int make_geom(psa_ctx* c, int slot, int64_t K_local, int64_t n_g, const int* d_idx, const int32_t* h_idx,
              bool disp, ProjGeom* g) {
    g->T = c->slot[slot].T;
    g->N_tot = c->slot[slot].N;
    PSA_REQUIRE(n_g < (1ll << 30) && K_local < (1ll << 29), "group or k-list too large");
    g->n_g = (int)n_g;
    g->A_pad = (int)((n_g + 31) / 32 * 32);
    g->K = (int)K_local;
    // product path: split-precision matrix-core kernels for velocity data -- "2 x f16" for whole-
    // trajectory groups with 2K > 64 (unless the array holds NaN/Inf), "3 x bf16" for every other
    // group; exact-fp32 MFMA kernel for displacement mode
    g->split = 0;
    const bool autosel = c->k1_selector == PSA_K1_AUTO;
    if (autosel && k1_pair_eligible(d_idx, g->N_tot, n_g, K_local, disp)) {
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
    if (g->split == 0 && (autosel || c->k1_selector == PSA_K1_SPLIT_BF16) &&
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

// phase table + projection of one group into q (K_local,3,T); no FFT
int project_group(psa_ctx* c, int slot, const int* d_idx, const ProjGeom& g, bool disp, float2* d_q) {
    const bool split = g.split == 3;
    PSA_TRY(c->d_phase.reserve(g.split == 2 ? pf16_table_bytes(g.M_pad, g.A_pad)
                               : split     ? pb_table_bytes(g.M_pad, g.A_pad)
                                           : p_table_floats(g.M_pad, g.A_pad) * sizeof(float)));
    {
        StageTimer st(c, PSA_T_PHASE);
        if (g.split == 2)
            PSA_TRY(launch_phase_table_f16(c, c->d_kvec.as<float>(), c->d_mean_all.as<float>(), d_idx, c->d_phase.ptr, g));
        else if (split)
            PSA_TRY(launch_phase_table_split(c, c->d_kvec.as<float>(), c->d_mean_all.as<float>(), d_idx,
                                             c->d_phase.ptr, g));
        else
            PSA_TRY(launch_phase_table(c, c->d_kvec.as<float>(), c->d_mean_all.as<float>(), d_idx,
                                       c->d_phase.as<float>(), g));
        if (disp) {
            PSA_TRY(c->d_mean_g.reserve((size_t)g.A_pad * 3 * sizeof(float)));
            PSA_TRY(launch_gather_mean(c, c->d_mean_all.as<float>(), d_idx, c->d_mean_g.as<float>(), g));
        }
    }
    {
        StageTimer st(c, PSA_T_PROJECT);
        const float* d_v = c->slot[slot].buf.as<float>();
        if (g.split == 2)
            PSA_TRY(launch_k1_pair(c, d_v, c->d_phase.ptr, d_idx, d_q, g));
        else if (split)
            PSA_TRY(launch_k1_split(c, d_v, c->d_phase.ptr, d_idx, d_q, g));
        else if (c->k1_selector == PSA_K1_WAVE)
            PSA_TRY(launch_k1_wave(c, d_v, c->d_phase.as<float>(), d_idx, c->d_mean_g.as<float>(), d_q,
                                   g, disp));
        else
            PSA_TRY(launch_k1_mfma(c, d_v, c->d_phase.as<float>(), d_idx, c->d_mean_g.as<float>(), d_q,
                                   g, disp));
    }
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
        for (DevBuf* b : {&c->d_kvec, &c->d_mean_all, &c->d_idx, &c->d_mean_g, &c->d_phase, &c->d_qwork,
                          &c->d_fft_work, &c->d_tables, &c->d_absmax, &c->d_slab, &c->d_out, &c->d_aux, &c->d_sync})
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
int psa_data_alloc(psa_ctx* c, int slot, int64_t T, int64_t N) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(slot >= 0 && slot < PSA_NUM_SLOTS, "bad data slot %d", slot);
    PSA_REQUIRE(T > 0 && N > 0, "empty trajectory (T=%lld, N=%lld)", (long long)T, (long long)N);
    Guard g(c);
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
    return PSA_OK;
}

int psa_data_upload(psa_ctx* c, int slot, const float* host, int64_t T, int64_t N) {
    PSA_REQUIRE(host != nullptr, "null host array");
    PSA_TRY(psa_data_alloc(c, slot, T, N));
    Guard g(c);
    StageTimer st(c, PSA_T_H2D);
    // stream in 256 MiB pieces: pageable source, keeps the staging footprint bounded
    const size_t total = (size_t)T * N * 3 * sizeof(float), piece = 256ull << 20;
    for (size_t o = 0; o < total; o += piece) {
        const size_t n = std::min(piece, total - o);
        PSA_HIP_CHECK(hipMemcpyAsync((char*)c->slot[slot].buf.ptr + o, (const char*)host + o, n,
                                     hipMemcpyHostToDevice, c->stream));
    }
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
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
    }
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

int psa_data_fill_synthetic(psa_ctx* c, int slot, uint64_t seed, int n_modes, const float* amp,
                            const int32_t* mode_comp, const float* ct, const float* st,
                            const float* ca, const float* sa) {
    PSA_TRY(enter(c));
    Guard g(c);
    PSA_TRY(check_slot(c, slot));
    PSA_REQUIRE(n_modes >= 0 && n_modes <= 16, "n_modes must be in [0,16]");
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
    PSA_TRY(launch_fill_synthetic(c, c->slot[slot].buf.as<float>(), T, N, seed, n_modes,
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

// ---- the hot path ---------------------------------------------------------------
int psa_sed_project(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                    int64_t K_local, int64_t K_total, int64_t k_offset, const int32_t* group_idx,
                    const int64_t* group_off, int32_t G, int32_t flags) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(check_slot(c, slot));
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    const bool intensity = (flags & PSA_F_INTENSITY) != 0;
    bool       disp = (flags & PSA_F_DISPLACEMENTS) != 0;
    PSA_REQUIRE(mean_pos_all != nullptr, "null mean_pos_all");
    PSA_REQUIRE(K_local >= 0 && K_total >= 1 && k_offset >= 0 && k_offset + K_local <= K_total,
                "k range [%lld,%lld) outside [0,%lld)", (long long)k_offset,
                (long long)(k_offset + K_local), (long long)K_total);
    PSA_REQUIRE(K_local == 0 || k_vectors != nullptr, "null k_vectors");
    PSA_TRY(validate_groups(N, group_idx, group_off, G));
    PSA_REQUIRE(intensity || G == 1, "complex output needs exactly one atom group (got %d)", G);

    // result slab, k-major
    const size_t row_bytes = intensity ? (size_t)T * sizeof(float) : (size_t)T * 3 * sizeof(float2);
    PSA_TRY(c->d_slab.reserve(row_bytes * (size_t)K_total));
    c->res_T = T;
    c->res_K = K_total;
    c->res_intensity = intensity;
    c->slab_valid = true;
    c->out_valid = false;
    if (K_local == 0) return PSA_OK;

    {
        StageTimer st(c, PSA_T_H2D);
        PSA_TRY(upload(c, c->d_kvec, k_vectors, (size_t)K_local * 3 * sizeof(float)));
        PSA_TRY(upload(c, c->d_mean_all, mean_pos_all, (size_t)N * 3 * sizeof(float)));
        if (group_idx)
            PSA_TRY(upload(c, c->d_idx, group_idx, (size_t)group_off[G] * sizeof(int32_t)));
    }
    PSA_TRY(materialise_displacements(c, &slot, &disp, mean_pos_all));

    char*   rows = (char*)c->d_slab.ptr + row_bytes * (size_t)k_offset;
    float2* d_q = intensity ? nullptr : (float2*)rows;
    if (intensity) {
        PSA_TRY(c->d_qwork.reserve((size_t)K_local * 3 * T * sizeof(float2)));
        d_q = c->d_qwork.as<float2>();
    }
    bool first = true;
    for (int gi = 0; gi < G; ++gi) {
        const int64_t n_g = group_idx ? (group_off[gi + 1] - group_off[gi]) : N;
        if (n_g == 0) continue;                                   // sed_calculator.py:64-65, 319-321
        const int* d_idx = group_idx ? c->d_idx.as<int>() + group_off[gi] : nullptr;
        ProjGeom g;
        PSA_TRY(make_geom(c, slot, K_local, n_g, d_idx, group_idx ? group_idx + group_off[gi] : nullptr, disp, &g));
        PSA_TRY(project_group(c, slot, d_idx, g, disp, d_q));
        {
            StageTimer st(c, PSA_T_FFT);
            PSA_TRY(run_fft(c, d_q, T, 3 * K_local));
        }
        if (intensity) {
            StageTimer st(c, PSA_T_EPILOGUE);
            PSA_TRY(launch_intensity_accumulate(c, d_q, (float*)rows, T, K_local, first));
        }
        first = false;
    }
    if (first)   // every group empty: the rows are zero
        PSA_HIP_CHECK(hipMemsetAsync(rows, 0, row_bytes * (size_t)K_local, c->stream));
    return PSA_OK;
}

int psa_sed_finalize(psa_ctx* c, void* out_host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->slab_valid) {
        set_error("psa_sed_finalize before psa_sed_project");
        return PSA_ESTATE;
    }
    const int64_t T = c->res_T, K = c->res_K;
    const size_t  bytes = c->res_intensity ? (size_t)T * K * sizeof(float) : (size_t)T * K * 3 * sizeof(float2);
    PSA_TRY(c->d_out.reserve(bytes));
    {
        StageTimer st(c, PSA_T_TRANSPOSE);
        if (c->res_intensity)
            PSA_TRY(launch_transpose_f32(c, c->d_slab.as<float>(), c->d_out.as<float>(), T, K));
        else
            PSA_TRY(launch_scale_transpose_c64(c, c->d_slab.as<float2>(), c->d_out.as<float2>(), T, K));
    }
    c->out_valid = true;
    if (out_host) {
        StageTimer st(c, PSA_T_D2H);
        PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_out.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return PSA_OK;
}

int psa_sed_calculate(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                      int64_t K, const int32_t* group_idx, const int64_t* group_off, int32_t G,
                      int32_t flags, void* out_host) {
    PSA_REQUIRE(K >= 1, "need at least one k-vector");
    PSA_TRY(psa_sed_project(c, slot, mean_pos_all, k_vectors, K, K, 0, group_idx, group_off, G, flags));
    return psa_sed_finalize(c, out_host);
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

int psa_result_intensity(psa_ctx* c, float* out_host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->out_valid || c->res_intensity) {
        set_error("psa_result_intensity needs a finalized complex result");
        return PSA_ESTATE;
    }
    const int64_t n = c->res_T * c->res_K;
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

int psa_result_chiral_phase(psa_ctx* c, int c1, int c2, float* out_host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->out_valid || c->res_intensity) {
        set_error("psa_result_chiral_phase needs a finalized complex result");
        return PSA_ESTATE;
    }
    PSA_REQUIRE(c1 >= 0 && c1 < 3 && c2 >= 0 && c2 < 3, "component indices must be 0..2");
    const int64_t n = c->res_T * c->res_K;
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

int psa_debug_project_only(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                           int64_t K, const int32_t* idx, int64_t n_g, int32_t flags, void* out_host) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(check_slot(c, slot));
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    PSA_REQUIRE(mean_pos_all && k_vectors && out_host && K >= 1 && n_g >= 1, "bad argument");
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
    PSA_TRY(materialise_displacements(c, &slot, &disp, mean_pos_all));
    PSA_TRY(make_geom(c, slot, K, n_g, idx ? c->d_idx.as<int>() : nullptr, idx, disp, &g));
    const size_t bytes = (size_t)K * 3 * T * sizeof(float2);
    PSA_TRY(c->d_qwork.reserve(bytes));
    PSA_TRY(project_group(c, slot, idx ? c->d_idx.as<int>() : nullptr, g, disp, c->d_qwork.as<float2>()));
    PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_qwork.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
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
    // direct peer-to-peer exchange: every transfer rides its own xGMI link, no ring
    PSA_NCCL_CHECK(ncclGroupStart());
    for (int r = 0; r < c->nranks; ++r) {
        if (r == me) continue;
        const bool i_receive = (root < 0 || root == me) && k_counts[r] > 0;
        const bool i_send = (root < 0 || root == r) && k_counts[me] > 0;
        if (i_receive)
            PSA_NCCL_CHECK(ncclRecv(slab + row_floats * (size_t)k_offsets[r],
                                    row_floats * (size_t)k_counts[r], ncclFloat, r, c->comm, c->stream));
        if (i_send)
            PSA_NCCL_CHECK(ncclSend(slab + row_floats * (size_t)k_offsets[me],
                                    row_floats * (size_t)k_counts[me], ncclFloat, r, c->comm, c->stream));
    }
    PSA_NCCL_CHECK(ncclGroupEnd());
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
