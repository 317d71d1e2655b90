/*
 * psa_hip.h -- C ABI of libpsa_hip.so: the MI355X (gfx950) implementation of the
 * PSA spectral-energy-density hot path.
 *
 * The reference (h-walk/PSA) is pure Python and has no FFI of its own; the seam
 * this library replaces is the private method
 *
 *     SEDCalculator._calculate_sed_for_group(k_vectors_3d, group_atom_indices, mean_pos_all)
 *         -> (T, K, 3) complex64                     src/psa/core/sed_calculator.py:58-84
 *
 * plus the per-group |.|^2 accumulation of the incoherent branch of
 * SEDCalculator.calculate (sed_calculator.py:313-327).  Every entry point is
 * plain C: opaque context pointer, raw host pointers, sizes; no C++ or torch types.
 * The ctypes stub that binds it is psa_amd/_hip.py; INTEGRATION.md shows the
 * ten-line patch that routes the reference's own SEDCalculator through it.
 *
 * Conventions
 *   - every function returns 0 on success, a negative PSA_E* code on failure;
 *     psa_last_error() returns the text for the calling thread's last failure;
 *   - host arrays are C-contiguous, owned by the caller, only read/written
 *     during the call; device memory, rocFFT plans, streams and RCCL
 *     communicators are owned by the context;
 *   - entry points may be called from any host thread (each one selects the
 *     context's device itself); calls on one context are serialised by an
 *     internal mutex (the reference GUI calls from worker threads,
 *     src/psa/gui/psa_gui.py:1015, :2246).
 */
#ifndef PSA_HIP_H
#define PSA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSA_HIP_ABI_VERSION 3

/* error codes */
#define PSA_OK          0
#define PSA_EINVAL     -1   /* bad argument (shape, NULL, out-of-range index) */
#define PSA_EHIP       -2   /* HIP runtime failure            */
#define PSA_EFFT       -3   /* rocFFT failure                 */
#define PSA_ERCCL      -4   /* RCCL failure                   */
#define PSA_ESTATE     -5   /* call sequence violated (e.g. finalize before project) */
#define PSA_ENOMEM     -6

/* data slots: which resident (T, N, 3) float32 array a projection reads.
 * sed_calculator.py:69-72 projects velocities, or positions minus their mean. */
#define PSA_SLOT_VELOCITIES 0
#define PSA_SLOT_POSITIONS  1
#define PSA_NUM_SLOTS       2

/* flags for psa_sed_project */
#define PSA_F_DISPLACEMENTS  0x1  /* data = slot - mean_pos (sed_calculator.py:70)           */
#define PSA_F_INTENSITY      0x2  /* result = sum_g sum_c |S_g|^2, float32 (T,K)  (:313-327) */
                                  /* default: complex64 (T,K,3) of the (single) group (:296-311) */

/* projection-kernel selector (diagnostics; PSA_K1_AUTO is the product path) */
#define PSA_K1_AUTO   0  /* split-precision (fp32-equivalent) MFMA kernels: "2 x f16" for groups with more
                            than 16 k-vectors -- from the group's cached split planes when it has them
                            (PSA_OPT_PLANES), splitting on the fly otherwise -- "3 x bf16" below that and
                            for arrays holding NaN/Inf; displacement mode projects a materialised
                            positions - mean array the same way */
#define PSA_K1_WAVE   1  /* LDS-staged VALU kernel with wavefront shuffle sums       */
#define PSA_K1_MFMA32 2  /* always the exact-fp32 MFMA tile kernel                   */
#define PSA_K1_SPLIT_BF16 3  /* "3 x bf16" split-precision kernel for every velocity-mode group */

typedef struct psa_ctx psa_ctx;

/* ---- library / context ------------------------------------------------ */
int         psa_abi_version(void);
const char* psa_last_error(void);
int         psa_device_count(int* count);
/* page-locked host memory for result arrays: a D2H copy into it runs at the full PCIe rate and
 * touches no fresh pages (into pageable memory configuration 5's 101 MB result took 4.5 ms of copy
 * plus as much again in page faults).  Independent of any context. */
int         psa_host_alloc(size_t bytes, void** out);
int         psa_host_free(void* p);
int         psa_create(int device, psa_ctx** out);
int         psa_destroy(psa_ctx* ctx);
int         psa_synchronize(psa_ctx* ctx);
int         psa_set_k1(psa_ctx* ctx, int selector);     /* PSA_K1_* */
/* Tunables of the product path (defaults in brackets):
 *   PSA_OPT_PLANES        [1] keep, per atom group, the group's data scaled, split into its two
 *                             float16 pieces and laid out in MFMA-fragment order ("split planes",
 *                             4 bytes per value like the float32 array, the group's atoms compacted)
 *                             and project from them; 0 = always split on the fly
 *   PSA_OPT_PLANES_BUDGET [0] bytes of HBM the plane cache may hold; 0 = 45 % of the device.  Sets
 *                             are evicted least-recently-used; a group that does not fit is
 *                             projected by the on-the-fly kernels.  The float32 slot is never dropped.
 *   PSA_OPT_PLANES_EAGER  [0] 1 = build an index-list group's planes on its first projection
 *                             (default: on the second with the same list; "all atoms" always first)
 *   PSA_OPT_PLANES_MIN_K  [17] shortest k-list for which a group's planes are BUILT (a shorter list
 *                             uses them when they exist; otherwise the "3 x bf16" kernel on the
 *                             float32 array, which is HBM-bound at the same rate) */
#define PSA_OPT_PLANES         0
#define PSA_OPT_PLANES_BUDGET  1
#define PSA_OPT_PLANES_EAGER   2
#define PSA_OPT_PLANES_MIN_K   3
/*   PSA_OPT_FOLD_PAIRS    [1] when the whole k-list is given to one device, k-vectors whose exact
 *                             negation (or an identical twin) is also in the list are not projected:
 *                             S(-k)[w] = conj S(k)[(T-w) mod T] holds bit for bit in the reference's
 *                             arithmetic (float32 phase argument odd in k, real data), so the partner's
 *                             columns are written by the epilogue from the one projection.  A k-grid
 *                             symmetric about Gamma (examples/k_grid_heatmap_example.py:33-38) costs
 *                             half its projections and FFTs.  0 = project every vector. */
#define PSA_OPT_FOLD_PAIRS     4
/*   PSA_OPT_FFT_PRIME     [1] when a trajectory of T frames becomes resident (psa_data_upload /
 *                             psa_data_alloc) a host thread builds a one-k-vector rocFFT plan of length T,
 *                             so that rocFFT's run-time kernel compilation for that length happens beside
 *                             the upload instead of inside the first calculation.  Independently of this,
 *                             psa_create points ROCFFT_RTC_CACHE_PATH (unless set) at
 *                             $PSA_CACHE_DIR | $XDG_CACHE_HOME/psa_amd | $HOME/.cache/psa_amd so that later
 *                             processes load the compiled kernels instead of compiling them again. */
#define PSA_OPT_FFT_PRIME      5
/*   PSA_OPT_K1_LOADER_WAVES [1] which form of the planes kernel projects 128-row M blocks (more than 32
 *                             k-vectors): 1 = twelve wavefronts per workgroup of which four issue all
 *                             LDS-DMA and eight only multiply (k1_planes_lw.hip), 0 = eight wavefronts
 *                             that both load and multiply (k1_planes.hip; always used for the 64- and
 *                             32-row blocks of shorter lists).  Same arithmetic, same results; the loader
 *                             form measured 2-3 % faster on every shape (round 3). */
#define PSA_OPT_K1_LOADER_WAVES 6
/*   PSA_OPT_K1_WIDE       [0] 1 = lists of more than 64 k-vectors are projected in 256-row M blocks
 *                             (k1_planes_wide.hip: eight wavefronts, a ring of 1-KiB units, 28 KiB of LDS-DMA
 *                             per 128 rows and stage instead of 40).  Same arithmetic; the float32 fold
 *                             comes every 10 stages instead of every 8. */
#define PSA_OPT_K1_WIDE         7
int         psa_set_option(psa_ctx* ctx, int option, int64_t value);
/* device name / CU count / HBM bytes of the context's GPU */
int         psa_device_info(psa_ctx* ctx, char* name, int name_len,
                            int* compute_units, int64_t* hbm_bytes);

/* ---- trajectory residency ---------------------------------------------
 * Trajectory.velocities / .positions are (T, N, 3) float32 C-order
 * (src/psa/core/trajectory.py:20-23); they stay in HBM in exactly that layout. */
/* psa_data_upload streams the array through two page-locked staging buffers (filled by a few
 * host threads, drained by hipMemcpyAsync on a copy stream) -- the source may be pageable or a
 * memory-mapped .npy cache (src/psa/io/loader.py:48-79). */
int psa_data_upload(psa_ctx* ctx, int slot, const float* host, int64_t T, int64_t N);
int psa_data_alloc(psa_ctx* ctx, int slot, int64_t T, int64_t N);
int psa_data_download(psa_ctx* ctx, int slot, float* host, int64_t t0, int64_t nt);
int psa_data_release(psa_ctx* ctx, int slot);
int psa_data_shape(psa_ctx* ctx, int slot, int64_t* T, int64_t* N);

/* Fill a slot, already allocated with psa_data_alloc, with the synthetic
 * trajectory of psa_amd/synth.py (bit-identical NumPy twin there):
 *   v[t,a,c] = noise(seed,t,a,c) + sum_m [c==mode_comp[m]] amp[m]*(ct[m,t]*ca[m,a] + st[m,t]*sa[m,a])
 * tables are (n_modes, T) / (n_modes, N) float32, host.  The slot holds frames
 * [t_offset, t_offset + T) of the synthetic trajectory (noise counter and ct/st rows of those
 * frames): a frame-sharded rank generates its own slice. */
int psa_data_fill_synthetic(psa_ctx* ctx, int slot, uint64_t seed, int64_t t_offset, int n_modes,
                            const float* amp, const int32_t* mode_comp,
                            const float* ct, const float* st,
                            const float* ca, const float* sa);

/* mean over frames of a resident slot, float32 sequential-in-t accumulation then /T:
 * bit-identical to np.mean(positions, axis=0, dtype=np.float32) (sed_calculator.py:205). */
int psa_mean_positions(psa_ctx* ctx, int slot, float* mean_host /* (N,3) */);

/* The same mean for an array that stays on the host (velocity mode does not upload positions):
 * columns split over `threads` host threads (0 = up to 16), each a sequential float32 chain over the
 * frames like NumPy's -- bit-identical to np.mean(x, axis=0, dtype=np.float32), one pass at memory
 * bandwidth.  No context, no GPU. */
int psa_host_mean_frames(const float* x /* (T, cols) */, int64_t T, int64_t cols, float* mean_out, int threads);

/* ---- the hot path -------------------------------------------------------
 * psa_sed_project: for each of the G atom groups, for the K_local k-vectors given,
 *     P[k,a]   = exp(i * (k . mean_pos[idx[a]]))            float32 FMA chain + sincos
 *     q[k,c,t] = sum_a  d[t, idx[a], c] * P[k,a]            split-precision f16 MFMA (fp32-equivalent), LDS-staged tiles
 *     S[k,c,w] = FFT_t(q) / T                               batched rocFFT, in place
 * and either keeps S of the (single) group as complex64, or accumulates
 * sum_c |S|^2 over groups as float32 (PSA_F_INTENSITY).  The result stays on the
 * device, k-major, as rows [k_offset, k_offset+K_local) of a K_total-row slab so
 * that k-sharded ranks produce contiguous pieces of one array.
 *
 *   mean_pos_all : (N, 3) float32 host          (sed_calculator.py:205)
 *   k_vectors    : (K_local, 3) float32 host    (rows k_offset.. of the full list)
 *   group_idx    : concatenated atom indices of all groups (int32), or NULL with
 *                  G = 1 meaning "all N atoms in order"
 *   group_off    : (G+1) int64 offsets into group_idx (ignored when group_idx NULL)
 * Without PSA_F_INTENSITY, G must be 1.
 * The first projection of a group after an upload does work that is cached afterwards: one pass for
 * the largest magnitude of the data (the power-of-two scale of the float16 kernels; per 32-atom
 * column block for index-list groups; of slot - mean in displacement mode) with a blocking
 * read-back -- psa_sed_project_upload folds it into the upload -- and, under PSA_OPT_PLANES, the
 * build of the group's split planes.  Later calls are asynchronous on the context's stream.
 */
int psa_sed_project(psa_ctx* ctx, int slot,
                    const float* mean_pos_all,
                    const float* k_vectors, int64_t K_local,
                    int64_t K_total, int64_t k_offset,
                    const int32_t* group_idx, const int64_t* group_off, int32_t G,
                    int32_t flags);

/* The same for an array that is not resident yet: uploads `host` (T,N,3) into the slot AND
 * projects, overlapped -- the array travels in chunks of frames through the staging pipeline of
 * psa_data_upload, and each chunk's frames are projected (first atom group; the projection is
 * independent per frame, sed_calculator.py:80-81) on the compute stream while the next chunk is
 * on the PCIe link; the rocFFT plan is built meanwhile on a host thread.  Further groups, the FFT
 * and the epilogue follow on the resident array.  Result and state as after psa_data_upload +
 * psa_sed_project (all K on this device). */
int psa_sed_project_upload(psa_ctx* ctx, int slot, const float* host, int64_t T, int64_t N,
                           const float* mean_pos_all, const float* k_vectors, int64_t K,
                           const int32_t* group_idx, const int64_t* group_off, int32_t G,
                           int32_t flags);

/* Transpose the K_total-row slab to the reference's layout -- (T,K,3) complex64
 * (sed_calculator.py:277) or (T,K) float32 (:280) -- and copy it to out_host
 * (may be NULL: the result then only exists on the device, see psa_result_*).
 * out_bytes is the size of the caller's buffer and must be exactly the result's
 * (24*T*K or 4*T*K): a mismatch is PSA_EINVAL, nothing is copied.
 * out_intensity (may be NULL; complex results only): (T,K) float32 = sum_c |S|^2, i.e. SED.intensity
 * (src/psa/core/sed.py:22-24) of this very result -- produced by the same pass over the result (the
 * tile is in LDS anyway) and copied beside it; out_intensity_bytes must be 4*T*K. */
int psa_sed_finalize(psa_ctx* ctx, void* out_host, size_t out_bytes,
                     float* out_intensity, size_t out_intensity_bytes);

/* one-call convenience: project all K on this device, finalize, copy out.  A complex result of
 * >= 192 k-vectors leaves block by block (project -> FFT -> transpose -> D2H on a copy stream while the
 * next block is projected).  out_intensity as for psa_sed_finalize. */
int psa_sed_calculate(psa_ctx* ctx, int slot, const float* mean_pos_all,
                      const float* k_vectors, int64_t K,
                      const int32_t* group_idx, const int64_t* group_off, int32_t G,
                      int32_t flags, void* out_host, size_t out_bytes,
                      float* out_intensity, size_t out_intensity_bytes);

/* Pair folding (PSA_OPT_FOLD_PAIRS) as a service for callers that split a k-list themselves
 * (psa_amd/dist.py): kmap[i] = row of k-vector i among the n_unique vectors that need projecting
 * (unique_idx[r] = position of row r's vector in the input list), with bit 31 set when vector i is
 * the exact negation of that row's vector.  No context, no GPU. */
#define PSA_KMAP_MIRROR 0x80000000u
int psa_k_pairs(const float* k_vectors, int64_t K, int32_t* kmap /* K */,
                int32_t* unique_idx /* K, first n_unique valid */, int64_t* n_unique);
/* ... and the map of a result whose slab rows the caller projected from the folded list (every rank
 * that finalizes installs it after its psa_sed_project / psa_sed_gather): psa_sed_finalize then
 * returns K_out columns, column i from slab row kmap[i] & ~PSA_KMAP_MIRROR, mirrored in frequency
 * and conjugated where bit 31 is set. */
int psa_sed_set_kmap(psa_ctx* ctx, const int32_t* kmap, int64_t K_out);

/* One (k, omega) bin: S[c] = FFT_t(q)[i_w] / T for ONE k-vector and one atom group, as 3
 * complex64 -- what iSED consumes of a group's spectrum (sed_calculator.py:483, :494-499: only
 * sed[i_w, i_k, :] of the full path spectrum is used).  One pass over the trajectory and one DFT
 * dot instead of K projections and 3K FFTs.  idx NULL = all atoms. */
int psa_sed_single_bin(psa_ctx* ctx, int slot, const float* mean_pos_all, const float* k_vector,
                       const int32_t* idx, int64_t n_g, int32_t flags, int64_t i_w,
                       float* out_c64x3 /* 6 floats */);

/* Raw access to rows [row0, row0+nrows) of the k-major slab (complex64 (nrows,3,T) or float32
 * (nrows,T), whichever the last psa_sed_project produced): lets a host transport stand in for
 * psa_sed_gather when no RCCL communicator can be formed, and serves checkpointing. */
int psa_slab_read(psa_ctx* ctx, int64_t row0, int64_t nrows, void* host);
int psa_slab_write(psa_ctx* ctx, int64_t row0, int64_t nrows, const void* host);

/* SED.intensity of the finalized complex result, (T,K) float32 = sum_c |S|^2 (src/psa/core/sed.py:22-24):
 * psa_sed_finalize / psa_sed_calculate leave it on the device next to the result; this copies it out
 * (out_host NULL: only makes sure it exists). */
int psa_result_intensity(psa_ctx* ctx, float* out_host /* (T,K) */, size_t out_bytes);
/* chiral phase, option "C", of components (c1, c2) of the finalized complex
 * result: (T,K) float32           (sed_calculator.py:344-350) */
int psa_result_chiral_phase(psa_ctx* ctx, int c1, int c2, float* out_host, size_t out_bytes);

/* stage timings of the last project/finalize on this context, milliseconds:
 * [0] host->device uploads  [1] phase table  [2] projection  [3] FFT
 * [4] |.|^2 / scale epilogue  [5] gather (RCCL)  [6] transpose  [7] device->host */
int psa_last_timings(psa_ctx* ctx, double* ms /* [8] */);
/* number of projection-kernel launches and their summed duration (HIP events on
 * the context's stream) since the last call of this function */
int psa_k1_stats(psa_ctx* ctx, int64_t* launches, double* total_ms);
/* host wall clock (ms) of work that is done once and then cached, summed since the last call:
 * [0] rocFFT plan builds (run-time compiled per (T, batch))  [1] largest-magnitude passes
 * [2] split-plane builds  [3] trajectory uploads (psa_data_upload / psa_sed_project_upload) */
int psa_oneoff_stats(psa_ctx* ctx, double* ms /* [4] */);

/* diagnostics for tests: the phase table of one group as (K,N_g) complex64, and the
 * pre-FFT projection q as (K,3,T) complex64 */
int psa_debug_phase_table(psa_ctx* ctx, const float* mean_pos_all,
                          const float* k_vectors, int64_t K,
                          const int32_t* idx, int64_t n_g, int64_t N,
                          void* out_host);
int psa_debug_project_only(psa_ctx* ctx, int slot, const float* mean_pos_all,
                           const float* k_vectors, int64_t K,
                           const int32_t* idx, int64_t n_g, int32_t flags,
                           void* out_host);
/* the same for frames [t_begin, t_begin + t_count) only, written into those columns of a zeroed
 * (K,3,T) slab (what a frame-sharded or streaming projection does per piece) */
int psa_debug_project_frames(psa_ctx* ctx, int slot, const float* mean_pos_all,
                             const float* k_vectors, int64_t K,
                             const int32_t* idx, int64_t n_g, int32_t flags,
                             int64_t t_begin, int64_t t_count, void* out_host);
/* number of plane sets in the cache and their bytes */
int psa_debug_plane_cache(psa_ctx* ctx, int64_t* n_sets, int64_t* bytes);

/* ---- k-point sharding over the GPUs of a node (one process per GPU) ------
 * rank 0 calls psa_comm_unique_id and ships the 128 bytes to the other ranks by
 * any host channel; every rank then calls psa_comm_init.  psa_sed_gather moves each
 * rank's rows of the slab to `root` over RCCL (xGMI) -- or to every rank when root is
 * -1 -- as grouped point-to-point transfers; k_offsets/k_counts are the (nranks) row
 * ranges.  After it, the receiving rank(s) call psa_sed_finalize. */
#define PSA_UNIQUE_ID_BYTES 128
int psa_comm_unique_id(void* out /* PSA_UNIQUE_ID_BYTES */);
int psa_comm_init(psa_ctx* ctx, const void* unique_id, int rank, int nranks);
int psa_comm_destroy(psa_ctx* ctx);
/* every pair of ranks trades one small stamped block in the grouped point-to-point pattern the
 * data path uses; PSA_ERCCL if anything arrives damaged.  Collective: all ranks call it. */
int psa_comm_selftest(psa_ctx* ctx);
int psa_sed_gather(psa_ctx* ctx, int root, const int64_t* k_offsets, const int64_t* k_counts);
int psa_comm_barrier(psa_ctx* ctx);

/* ---- frame sharding: one exchange step before the FFT -------------------------------
 * The projection is linear in atoms and independent per frame (sed_calculator.py:80-81), the FFT
 * runs along frames (:83).  Rank r holds only frames [t_offsets[r], +t_counts[r]) of the
 * trajectory in its slot (1/n of the array), projects ALL K k-vectors on them, then an
 * all-to-all over RCCL hands every rank the frames it lacks of ITS block of k rows
 * [k_offsets[r], +k_counts[r]); FFT and epilogue run on those rows, which end up in the same
 * k-major slab rows as with k-sharding -- psa_sed_gather / psa_sed_finalize follow unchanged.
 * Per atom group (G groups need PSA_F_INTENSITY as usual):
 *     psa_sed_fs_project   q_local (K_total,3,T_local) of the group, on the device
 *     psa_sed_fs_exchange  the all-to-all (grouped ncclSend/ncclRecv, direct links)
 *                          -- or psa_sed_fs_read / psa_sed_fs_write when a host transport stands in
 *     psa_sed_fs_finish    FFT over T_total of my rows, then complex rows into the slab or
 *                          |.|^2 accumulated into it (first_group: overwrite) */
int psa_sed_fs_project(psa_ctx* ctx, int slot, const float* mean_pos_all,
                       const float* k_vectors, int64_t K_total,
                       const int32_t* idx, int64_t n_g, int32_t flags,
                       int64_t T_total, int64_t k_offset, int64_t k_count);
int psa_sed_fs_exchange(psa_ctx* ctx, const int64_t* t_offsets, const int64_t* t_counts,
                        const int64_t* k_offsets, const int64_t* k_counts);
/* rows [k0, k0+nk) of q_local as (nk,3,T_local) complex64 */
int psa_sed_fs_read(psa_ctx* ctx, int64_t k0, int64_t nk, void* host);
/* frames [t0, t0+nt) of my rows, (k_count,3,nt) complex64 */
int psa_sed_fs_write(psa_ctx* ctx, int64_t t0, int64_t nt, const void* host);
int psa_sed_fs_finish(psa_ctx* ctx, int32_t first_group);

#ifdef __cplusplus
}
#endif
#endif /* PSA_HIP_H */
