// Sharding one calculation over the GPUs of a node (one process per GPU): RCCL communicator, gather of
// k rows, frame sharding with an all-to-all before the FFT, self-test, barrier.
// (part of the C ABI of libpsa_hip.so, include/psa_hip.h; shared declarations: api_internal.h)
#include "api_internal.h"

using namespace psa;

extern "C" {

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
