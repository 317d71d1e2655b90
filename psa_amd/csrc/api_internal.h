// Internal declarations shared by the api_*.hip translation units (the C ABI of libpsa_hip.so,
// include/psa_hip.h):
//   api_core.hip     context, error text, timing, rocFFT plans, options
//   api_data.hip     trajectory residency: staging pipeline, uploads, magnitude passes, mean, displacements
//   api_project.hip  the hot path: plane cache, geometry, projection, project / finalize / calculate, diagnostics
//   api_shard.hip    sharding over RCCL: communicator, k-row gather, frame sharding
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <unordered_map>

#include "k1_f16.h"

namespace psa {

extern thread_local std::string g_error;                 // text of the calling thread's last failure

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

int          enter(psa_ctx* c);                          // null check + hipSetDevice
TimingState& timing(psa_ctx* c);
int          get_event(TimingState& ts, hipEvent_t* ev);
int          collect(psa_ctx* c, TimingState& ts);

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

int upload(psa_ctx* c, DevBuf& b, const void* host, size_t bytes);
int get_plan(psa_ctx* c, int64_t T, int64_t batch, FftPlan** out);
void prime_fft(psa_ctx* c, int64_t T);               // background build of the length's kernels (api_core.hip)
int run_fft(psa_ctx* c, float2* data, int64_t T, int64_t batch);
int check_slot(psa_ctx* c, int slot);
int validate_groups(int64_t N, const int32_t* group_idx, const int64_t* group_off, int32_t G);

// --- api_data.hip
int  slot_absmax(psa_ctx* c, int slot);
int  group_absmax(psa_ctx* c, int slot, const int32_t* h_idx, int64_t n_g, unsigned* bits);
int  displaced_absmax(psa_ctx* c, int slot, const float* mean_host, const int32_t* h_idx, int64_t n_g, unsigned* bits);
int  stager_init(psa_ctx* c, size_t chunk_bytes);
void stager_release(psa_ctx* c);
int  staged_upload(psa_ctx* c, float* dev, const float* host, int64_t T, int64_t N,
                   const std::function<int(int64_t, int64_t, hipEvent_t)>& on_chunk);
int  data_alloc_locked(psa_ctx* c, int slot, int64_t T, int64_t N);
int  materialise_displacements(psa_ctx* c, int* slot_io, bool* disp, const float* mean_host);

// --- api_project.hip
size_t planes_bytes_held(psa_ctx* c);
void   drop_stale_planes(psa_ctx* c);
int    get_planes(psa_ctx* c, int slot, const int* d_idx, const int32_t* h_idx, int64_t n_g, int64_t K_local,
                  const float* mean_host, PlaneSet** out);
int    make_geom(psa_ctx* c, int slot, int64_t K_local, int64_t n_g, const int* d_idx, const int32_t* h_idx, bool disp,
                 const PlaneSet* ps, int force, ProjGeom* g);
int    prepare_phase(psa_ctx* c, const int* d_idx, const ProjGeom& g, bool disp, int64_t k_first = 0);
int    launch_projection(psa_ctx* c, int slot, const int* d_idx, ProjGeom g, bool disp, const PlaneSet* ps, float2* d_q,
                         int64_t q_stride, int64_t t_begin, int64_t t_count);
int    project_group(psa_ctx* c, int slot, const int* d_idx, const ProjGeom& g, bool disp, const PlaneSet* ps, float2* d_q);
int    group_source(psa_ctx* c, int* slot_io, bool* disp_io, const float* mean_host, const int* d_idx, const int32_t* h_idx,
                    int64_t n_g, int64_t K, PlaneSet** ps);
void   fold_pairs(const float* k, int64_t K, std::vector<int32_t>* kmap, std::vector<int32_t>* unique_idx);
int    install_kmap(psa_ctx* c, const std::vector<int32_t>& kmap);
int    begin_result(psa_ctx* c, int64_t T, int64_t K_total, int64_t k_offset, bool intensity, char** rows, size_t* row_bytes);

}  // namespace psa
