// Trajectory residency: host -> device staging pipeline, uploads, magnitude passes, mean over frames,
// displacement array.
// (part of the C ABI of libpsa_hip.so, include/psa_hip.h; shared declarations: api_internal.h)
#include "api_internal.h"

namespace psa {

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

int data_alloc_locked(psa_ctx* c, int slot, int64_t T, int64_t N) {
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

}  // namespace psa

using namespace psa;

extern "C" {

// ---- trajectory residency ----------------------------------------------------
// np.mean(x, axis=0, dtype=float32) of a C-contiguous (T, cols) float32 array ON THE HOST, bit for bit
// (sed_calculator.py:205): NumPy adds the rows one after the other into a float32 row of sums and
// divides by T in float32 -- every column is its own sequential chain, so the columns are split over
// threads and each thread streams its share of every row.  One pass at memory bandwidth (a 25.8 GB
// positions array: ~0.5 s with 8 threads) where the single-threaded NumPy call takes seconds; used in
// velocity mode, where the positions are not uploaded (displacement mode: mean_over_frames_kernel).
int psa_host_mean_frames(const float* x, int64_t T, int64_t cols, float* mean_out, int threads) {
    PSA_REQUIRE(x != nullptr && mean_out != nullptr && T >= 1 && cols >= 1, "bad argument");
    int n = threads > 0 ? threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    n = (int)std::min<int64_t>(n, std::max<int64_t>(1, cols / 256));          // at least 1 KiB of every row per thread
    const float count = (float)T;
    auto work = [=](int64_t c0, int64_t c1) {
        std::vector<float> acc((size_t)(c1 - c0), 0.f);
        float*             a = acc.data();
        const int64_t      w = c1 - c0;
        for (int64_t t = 0; t < T; ++t) {
            const float* row = x + t * cols + c0;
            for (int64_t j = 0; j < w; ++j) a[j] += row[j];                     // independent chains: vectorises as it stands
        }
        for (int64_t j = 0; j < w; ++j) mean_out[c0 + j] = a[j] / count;
    };
    std::vector<std::thread> pool;
    const int64_t            per = (cols + n - 1) / n;
    for (int i = 1; i < n; ++i) {
        const int64_t c0 = std::min<int64_t>(cols, i * per), c1 = std::min<int64_t>(cols, c0 + per);
        if (c1 > c0) pool.emplace_back(work, c0, c1);
    }
    work(0, std::min<int64_t>(cols, per));
    for (auto& t : pool) t.join();
    return PSA_OK;
}

int psa_data_alloc(psa_ctx* c, int slot, int64_t T, int64_t N) {
    PSA_TRY(enter(c));
    Guard g(c);
    PSA_TRY(data_alloc_locked(c, slot, T, N));
    prime_fft(c, T);
    return PSA_OK;
}

int psa_data_upload(psa_ctx* c, int slot, const float* host, int64_t T, int64_t N) {
    PSA_TRY(enter(c));
    PSA_REQUIRE(host != nullptr, "null host array");
    Guard g(c);
    PSA_TRY(data_alloc_locked(c, slot, T, N));
    prime_fft(c, T);                                   // the FFT kernels of this length compile beside the upload
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

}  // extern "C"
