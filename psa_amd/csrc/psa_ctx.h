// Internal declarations shared by the translation units of libpsa_hip.so.
// Public surface: include/psa_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "psa_hip.h"

namespace psa {

void set_error(const char* fmt, ...);

#define PSA_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) {                                                          \
            psa::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                           hipGetErrorString(e_));                                       \
            return PSA_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define PSA_FFT_CHECK(expr)                                                              \
    do {                                                                                 \
        rocfft_status s_ = (expr);                                                       \
        if (s_ != rocfft_status_success) {                                               \
            psa::set_error("%s:%d: %s -> rocfft_status %d", __FILE__, __LINE__, #expr,   \
                           (int)s_);                                                     \
            return PSA_EFFT;                                                             \
        }                                                                                \
    } while (0)

#define PSA_NCCL_CHECK(expr)                                                             \
    do {                                                                                 \
        ncclResult_t r_ = (expr);                                                        \
        if (r_ != ncclSuccess) {                                                         \
            psa::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                           ncclGetErrorString(r_));                                      \
            return PSA_ERCCL;                                                            \
        }                                                                                \
    } while (0)

#define PSA_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            psa::set_error(__VA_ARGS__);                                                 \
            return PSA_EINVAL;                                                           \
        }                                                                                \
    } while (0)

#define PSA_TRY(expr)                                                                    \
    do {                                                                                 \
        int rc_ = (expr);                                                                \
        if (rc_ != PSA_OK) return rc_;                                                   \
    } while (0)

// grow-only device allocation
struct DevBuf {
    void*  ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return static_cast<T*>(ptr); }
};

struct FftPlan {
    rocfft_plan           plan = nullptr;
    rocfft_execution_info info = nullptr;
    size_t                work_bytes = 0;
};

struct DataSlot {
    DevBuf  buf;
    int64_t T = 0, N = 0;
    bool    valid = false;
    uint64_t generation = 0;             // bumped whenever the contents change
    // largest |x| of the array (float bits), computed on first use after the contents change
    bool     absmax_known = false;
    unsigned absmax_bits = 0;
    // ... and per column block of 32 atoms (host copy), for index-list groups
    bool                  blocks_known = false;
    std::vector<unsigned> block_absmax;
};

// stage timing: event pairs recorded on the context's stream, resolved lazily
struct Timed {
    int        stage;
    hipEvent_t e0, e1;
};
struct TimingState {
    std::vector<hipEvent_t> pool;
    std::vector<Timed>      pending;
    double                  acc[8] = {};
    int64_t                 k1_launches = 0;
    double                  k1_ms = 0.0;
};

// P' (phase table) lives in HBM as the LDS tile images the projection kernel DMAs in:
//   [M block][atom stage][row in block][K1_PROW floats: 32 atoms + 4 pad]
constexpr int K1_BA   = 32;   // atoms per LDS stage
constexpr int K1_VROW = 96;   // floats per staged V row (32 atoms x 3 components)
constexpr int K1_PROW = 36;   // floats per staged P' row (odd number of 16-byte slots)

__host__ __device__ inline size_t p_tile_index(int m, int a, int m_blk, int n_stage) {
    return ((size_t)(m / m_blk) * n_stage + (a / K1_BA)) * ((size_t)m_blk * K1_PROW) +
           (size_t)(m % m_blk) * K1_PROW + (a % K1_BA);
}
inline size_t p_table_floats(int M_pad, int A_pad) { return (size_t)M_pad * (A_pad / K1_BA) * K1_PROW; }

// geometry of one projection launch (see k1_mfma.hip)
struct ProjGeom {
    int64_t T = 0;        // frames of this launch
    int64_t q_stride = 0; // frames per row of the q slab the launch writes into (>= T)
    int64_t N_tot = 0;    // atoms in the resident array
    int     n_g = 0;      // atoms in this group
    int     A_pad = 0;    // n_g rounded up to the atom stage (32)
    int     K = 0;        // k-vectors (rows of the output)
    int     M_pad = 0;    // 2K rounded up to the variant's M block
    int     m_blk = 0;    // rows of P per workgroup (variant)
    int     split = 0;    // 0: float32 kernels; 2: "2 x f16" kernel (k1_pair.hip); 3: "3 x bf16" (k1_split.hip);
                          // 4: "2 x f16" from the group's cached split planes (k1_planes.hip)
    float   vscale = 0.f; // split == 2: power of two applied to d (from the slot's largest magnitude)
};

// A group's data as cached split planes (k1_f16.h plane_index): built from one generation of one
// slot, for one atom list (or all atoms), with one power-of-two scale.
struct PlaneSet {
    DevBuf               buf;
    int                  slot = 0;
    uint64_t             generation = 0;
    bool                 all_atoms = true;
    std::vector<int32_t> idx;            // the atom list (compacted order), empty when all_atoms
    uint64_t             idx_hash = 0;
    bool                 displaced = false;   // the planes hold slot - mean (displacement mode)
    std::vector<float>   mean;                // ... with this mean (N,3)
    int64_t              T = 0, n_fg = 0;
    int                  n_g = 0, A_pad = 0;
    float                vscale = 0.f;
    uint64_t             last_use = 0;
};

// Entry of a k map (psa_ctx::kmap): output column k takes slab row (entry & ~KMAP_MIRROR); with
// KMAP_MIRROR set that row belongs to -k and is read as conj S[(T-w) mod T]  (k2_epilogue.hip)
constexpr int KMAP_MIRROR = (int)0x80000000u;

// page-locked staging buffers + copy stream of the host->device pipeline (psa_data_upload,
// psa_sed_project_upload)
struct Stager {
    void*       pin[2] = {nullptr, nullptr};
    size_t      cap = 0;
    hipEvent_t  freed[2] = {nullptr, nullptr};   // the H2D copy out of pin[i] has finished
    hipStream_t copy_stream = nullptr;
};

}  // namespace psa

enum { PSA_T_H2D = 0, PSA_T_PHASE, PSA_T_PROJECT, PSA_T_FFT, PSA_T_EPILOGUE, PSA_T_GATHER,
       PSA_T_TRANSPOSE, PSA_T_D2H, PSA_T_COUNT };

struct psa_ctx {
    int         device = 0;
    hipStream_t stream = nullptr;
    std::mutex  mu;
    int         k1_selector = PSA_K1_AUTO;
    int         compute_units = 0;

    // [PSA_NUM_SLOTS] is internal: positions minus their mean (displacement mode), materialised on
    // first use and kept while the positions and the mean stay the same
    psa::DataSlot slot[PSA_NUM_SLOTS + 1];
    std::vector<float> disp_mean;        // the mean the displacement array was built with
    uint64_t           disp_source = 0;  // generation of the positions slot it was built from
    // largest |positions - mean| per 32-atom column block (scale of displacement-mode planes)
    std::vector<unsigned> disp_block_absmax;
    std::vector<float>    disp_abs_mean;
    uint64_t              disp_abs_source = ~0ull;

    // per-call scratch
    psa::DevBuf d_kvec, d_mean_all, d_idx, d_mean_g, d_phase, d_qwork, d_fft_work, d_tables, d_absmax;
    psa::DevBuf d_upload_max;                 // running largest magnitude of an array being uploaded
    psa::DevBuf d_zeros;                      // 1 KiB of zeros (k1_planes_wide.hip: planes of the stages past a group's end)
    psa::DevBuf d_qrows, d_stage, d_bin;      // frame sharding: my rows before the FFT / all-to-all landing zone; one DFT bin

    // cached split planes (PSA_OPT_PLANES*)
    std::vector<std::unique_ptr<psa::PlaneSet>> planes;
    std::vector<uint64_t> seen_groups;        // hashes of index-list groups projected once already
    uint64_t plane_tick = 0, plane_call_mark = 0;   // sets touched since the mark belong to the call in progress
    int64_t  opt_planes = 1, opt_planes_budget = 0, opt_planes_eager = 0, opt_planes_min_k = 17;
    psa::Stager stager;
    hipStream_t d2h_stream = nullptr;         // result blocks leave while the next block is projected
    hipEvent_t  d2h_ready = nullptr;

    // frame sharding (psa_sed_fs_*): geometry of the group in flight
    int64_t fs_T_total = 0, fs_T_local = 0, fs_K_total = 0, fs_rows_k0 = 0, fs_rows_nk = 0;
    bool    fs_intensity = false;
    // results
    psa::DevBuf d_slab;      // k-major: (K_total,3,T) c64  or (K_total,T) f32
    psa::DevBuf d_out;       // reference layout: (T,K_total,3) c64 or (T,K_total) f32
    psa::DevBuf d_aux;       // (T,K_total) f32 for intensity / chiral phase of the result
    int64_t res_T = 0, res_K = 0;           // frames; ROWS of the slab (= k-vectors projected)
    bool    res_intensity = false;
    // k-vectors of the result when pairs (k, -k) / duplicates were folded: out_K columns, column k
    // from slab row kmap[k] (KMAP_MIRROR: as the partner of that row); empty = the slab's rows as they are
    int64_t              out_K = 0;
    std::vector<int32_t> kmap;
    psa::DevBuf          d_kmap, d_cols;
    int64_t              opt_fold_pairs = 1;
    psa::DevBuf          d_inten;                 // (T,out_K) f32: sum_c |d_out|^2 of a finalized complex result
    bool                 inten_valid = false;
    bool    slab_valid = false, out_valid = false;

    std::map<std::pair<int64_t, int64_t>, psa::FftPlan> plans;   // (T, batch)
    // rocFFT compiles the kernels of a length at run time on first use (tens to hundreds of ms): when a
    // trajectory of T frames becomes resident a host thread builds a small plan of that length, so the
    // first calculation finds the kernels compiled (and, through the per-user cache file, so does the
    // next process)
    std::thread  fft_primer;
    psa::FftPlan primed;
    int64_t      primed_T = 0;
    int64_t      opt_fft_prime = 1;
    int64_t      opt_k1_wide = 1;               // PSA_OPT_K1_WIDE: 256-row M blocks (k1_planes_wide.hip) where the k-list fills them (k1_planes_block_rows)
    int64_t      opt_k1_loader_waves = 1;       // PSA_OPT_K1_LOADER_WAVES: 128-row M blocks through k1_planes_lw.hip

    psa::TimingState timing;
    double oneoff_ms[4] = {0, 0, 0, 0};   // host wall clock of work done once: rocFFT plan builds, magnitude passes,
                                          // plane builds, trajectory uploads (psa_oneoff_stats)
    psa::DevBuf      d_sync;     // one float for the RCCL barrier

    ncclComm_t comm = nullptr;
    int        rank = 0, nranks = 1;
};

namespace psa {

// --- kernels_misc.hip
int launch_phase_table(psa_ctx* c, const float* d_kvec, const float* d_mean_all, const int* d_idx,
                       float* d_phase, const ProjGeom& g);
int launch_gather_mean(psa_ctx* c, const float* d_mean_all, const int* d_idx, float* d_mean_g,
                       const ProjGeom& g);
int launch_fill_synthetic(psa_ctx* c, float* d_v, int64_t T, int64_t N, uint64_t seed, int64_t t_offset, int n_modes,
                          const float* d_amp, const int* d_comp, const float* d_ct,
                          const float* d_st, const float* d_ca, const float* d_sa);
int launch_mean_over_frames(psa_ctx* c, const float* d_x, int64_t T, int64_t N, float* d_mean);
int launch_subtract_mean(psa_ctx* c, const float* d_x, const float* d_mean, float* d_out, int64_t T, int64_t N);
int launch_absmax_bits(psa_ctx* c, const float* d_x, int64_t n, unsigned* d_out, bool reset = true);
int launch_absmax_blocks(psa_ctx* c, const float* d_x, const float* d_mean, int64_t T, int64_t N, unsigned* d_out);

// --- k1_mfma.hip / k1_wave.hip
int  k1_mfma_block_rows(int K);                    // M block of the variant chosen for K
int  launch_k1_mfma(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                    const float* d_mean_g, float2* d_q, const ProjGeom& g, bool displacements);
int  launch_k1_wave(psa_ctx* c, const float* d_v, const float* d_phase, const int* d_idx,
                    const float* d_mean_g, float2* d_q, const ProjGeom& g, bool displacements);

// --- k1_split.hip ("3 x bf16": any velocity-mode group)
bool   k1_split_eligible(const int* d_idx, int64_t N_tot, int64_t n_g, bool displacements);
int    k1_split_block_rows(int K);
size_t pb_table_bytes(int M_pad, int A_pad);
int    launch_phase_table_split(psa_ctx* c, const float* d_kvec, const float* d_mean_all, const int* d_idx,
                                void* d_phase, const ProjGeom& g);
int    launch_k1_split(psa_ctx* c, const float* d_v, const void* d_phase, const int* d_idx, float2* d_q,
                       const ProjGeom& g);

// --- k1_pair.hip ("2 x f16": whole-trajectory groups, 2K > 64)
bool   k1_pair_eligible(const int* d_idx, int64_t N_tot, int64_t n_g, int64_t K, bool displacements);
int    k1_pair_atom_pad(int64_t n_g);
int    k1_pair_block_rows(int K);
float  k1_f16_vscale(unsigned absmax_bits);
size_t pf16_table_bytes(int M_pad, int A_pad);
int    launch_phase_table_f16(psa_ctx* c, const float* d_kvec, const float* d_mean_all, const int* d_idx,
                              void* d_phase, const ProjGeom& g);
int    launch_k1_pair(psa_ctx* c, const float* d_v, const void* d_phase, const int* d_idx, float2* d_q,
                      const ProjGeom& g);

// --- k1_planes.hip ("2 x f16" from cached split planes: every kind of group)
int    k1_planes_block_rows(int K, bool wide);
int    launch_split_planes(psa_ctx* c, const float* d_x, const float* d_mean, const int* d_idx, void* d_planes, int64_t T,
                           int64_t N_tot, int n_g, int A_pad, float vscale);
int    launch_k1_planes(psa_ctx* c, const void* d_planes, const void* d_phase, float2* d_q, const ProjGeom& g,
                        int64_t n_fg);
// --- k1_planes_lw.hip (the same with dedicated loader wavefronts; 128-row M blocks)
int    launch_k1_planes_lw(psa_ctx* c, const void* d_planes, const void* d_phase, float2* d_q, const ProjGeom& g,
                           int64_t n_fg);

// --- k1_planes_wide.hip (256-row M blocks: k-lists of more than 64 vectors under PSA_OPT_K1_WIDE)
int    launch_k1_planes_wide(psa_ctx* c, const void* d_planes, const void* d_phase, float2* d_q, const ProjGeom& g,
                             int64_t n_fg);

// --- k2_epilogue.hip
int launch_dft_bin(psa_ctx* c, const float2* d_q, int64_t T, int64_t bin, float2* d_out3);
// n output columns of a K_pitch-column result (T, K_pitch, 3) from rows of the k-major slab: column
// d_cols[i] (null: col_first + i) from row d_srcs[i] (null: src_first + i; KMAP_MIRROR: the partner
// -k of the row's k-vector); d_inten (may be null): sum_c |.|^2 of the same columns into (T, K_pitch)
int launch_scale_transpose_c64(psa_ctx* c, const float2* d_slab, float2* d_out, float* d_inten, int64_t T, int64_t n,
                               int64_t K_pitch, int64_t col_first, int64_t src_first, const int32_t* d_cols,
                               const int32_t* d_srcs);
int launch_intensity_accumulate(psa_ctx* c, const float2* d_q, float* d_slab_rows, int64_t T,
                                int64_t K_local, bool first_group);
int launch_transpose_f32(psa_ctx* c, const float* d_slab, float* d_out, int64_t T, int64_t K, const int32_t* d_srcs);
int launch_result_intensity(psa_ctx* c, const float2* d_out, float* d_int, int64_t n_tk);
int launch_result_chiral_c(psa_ctx* c, const float2* d_out, float* d_phase, int64_t n_tk, int c1, int c2);

}  // namespace psa
