// The hot path: plane cache, projection geometry, phase table + projection launch, psa_sed_project /
// _project_upload / _finalize / _calculate (pipelined) / _single_bin, slab access, results, diagnostics.
// (part of the C ABI of libpsa_hip.so, include/psa_hip.h; shared declarations: api_internal.h)
#include "api_internal.h"

namespace psa {

uint64_t hash_idx(const int32_t* p, int64_t n) {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
    for (int64_t i = 0; i < n; ++i) h = (h ^ (uint32_t)p[i]) * 1099511628211ull;
    return h;
}

// Pairs (k, -k) and repeated k-vectors of a list.  kmap[i] = index into the list of vectors that are
// actually projected (unique_idx: their positions in the input), | KMAP_MIRROR when vector i is the
// exact negation of that one.  Exact means component-wise float equality of k_i and -k_u (so -0 == 0;
// NaN never matches): then the float32 phase argument fma(kz,rz,fma(ky,ry,kx*rx)) is the exact negative
// (sed_calculator.py:78 -- rounding is symmetric), cos is even and sin odd, the data are real, hence
// q(-k) = conj q(k) and S(-k)[w] = conj S(k)[(T-w) mod T]: the partner needs no projection and no FFT
// of its own (the reference's own heat maps are symmetric about Gamma: examples/k_grid_heatmap_example.py:33-38).
void fold_pairs(const float* k, int64_t K, std::vector<int32_t>* kmap, std::vector<int32_t>* unique_idx) {
    struct Key {
        uint32_t x, y, z;
        bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; }
    };
    struct KeyHash {
        size_t operator()(const Key& a) const {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t v : {a.x, a.y, a.z}) h = (h ^ v) * 1099511628211ull;
            return (size_t)h;
        }
    };
    auto bits = [](float v, bool negate) {
        if (negate) v = -v;
        if (v == 0.f) v = 0.f;                        // -0 and +0 are the same k
        uint32_t u;
        std::memcpy(&u, &v, sizeof(u));
        return u;
    };
    std::unordered_map<Key, int32_t, KeyHash> seen;   // k-vector -> its row among the projected ones
    seen.reserve((size_t)K * 2);
    kmap->assign((size_t)K, 0);
    unique_idx->clear();
    for (int64_t i = 0; i < K; ++i) {
        const float* v = k + 3 * i;
        const bool   has_nan = v[0] != v[0] || v[1] != v[1] || v[2] != v[2];
        const Key    same{bits(v[0], false), bits(v[1], false), bits(v[2], false)};
        const Key    neg{bits(v[0], true), bits(v[1], true), bits(v[2], true)};
        if (!has_nan) {
            auto it = seen.find(same);
            if (it != seen.end()) {
                (*kmap)[i] = it->second;
                continue;
            }
            it = seen.find(neg);
            if (it != seen.end()) {
                (*kmap)[i] = it->second | KMAP_MIRROR;
                continue;
            }
        }
        const int32_t row = (int32_t)unique_idx->size();
        unique_idx->push_back((int32_t)i);
        (*kmap)[i] = row;
        if (!has_nan) seen.emplace(same, row);
    }
}

// the list the projection runs on when folding pays: *uniq_k receives the vectors to project and *kmap
// the k map (the caller installs it with install_kmap after begin_result); false = project the list as it is
static bool fold_k_list(psa_ctx* c, const float* k, int64_t K, std::vector<float>* uniq_k, std::vector<int32_t>* kmap) {
    if (!c->opt_fold_pairs || K < 2 || K >= (1ll << 30)) return false;
    std::vector<int32_t> uidx;
    fold_pairs(k, K, kmap, &uidx);
    if ((int64_t)uidx.size() == K) return false;
    uniq_k->resize(uidx.size() * 3);
    for (size_t u = 0; u < uidx.size(); ++u) std::memcpy(uniq_k->data() + 3 * u, k + 3 * (size_t)uidx[u], 3 * sizeof(float));
    return true;
}

int install_kmap(psa_ctx* c, const std::vector<int32_t>& kmap) {
    c->kmap = kmap;
    c->out_K = (int64_t)kmap.size();
    c->out_valid = false;
    return upload(c, c->d_kmap, c->kmap.data(), c->kmap.size() * sizeof(int32_t));
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
        g->m_blk = k1_planes_block_rows((int)K_local, c->opt_k1_wide != 0);
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
int prepare_phase(psa_ctx* c, const int* d_idx, const ProjGeom& g, bool disp, int64_t k_first) {
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
static int launch_projection_once(psa_ctx* c, int slot, const int* d_idx, ProjGeom g, bool disp, const PlaneSet* ps, float2* d_q,
                                  int64_t q_stride, int64_t t_begin, int64_t t_count);

int launch_projection(psa_ctx* c, int slot, const int* d_idx, ProjGeom g, bool disp, const PlaneSet* ps, float2* d_q,
                      int64_t q_stride, int64_t t_begin, int64_t t_count) {
    // PSA_DEBUG_REPEAT_K1=n (diagnostics, tools/short_loop_timing.py): the same launch n times back to back,
    // each timed on its own (psa_k1_stats) -- the result is that of one launch
    static const int reps = [] {
        const char* e = std::getenv("PSA_DEBUG_REPEAT_K1");
        return e ? std::max(1, std::atoi(e)) : 1;
    }();
    for (int r = 0; r < reps; ++r) PSA_TRY(launch_projection_once(c, slot, d_idx, g, disp, ps, d_q, q_stride, t_begin, t_count));
    return PSA_OK;
}

static int launch_projection_once(psa_ctx* c, int slot, const int* d_idx, ProjGeom g, bool disp, const PlaneSet* ps, float2* d_q,
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
        // PSA_OPT_K1_LOADER_WAVES [1]: 128-row M blocks go to the loader-wavefront form of the kernel
        // (k1_planes_lw.hip; 2-3 % faster than the eight-wavefront form on every shape, round 3); 0 = never
        if (g.m_blk == 256) return launch_k1_planes_wide(c, pl, c->d_phase.ptr, d_q, g, ps->n_fg - fg0);
        if (g.m_blk == 128 && c->opt_k1_loader_waves) return launch_k1_planes_lw(c, pl, c->d_phase.ptr, d_q, g, ps->n_fg - fg0);
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

// Where one group's data comes from.  In order: its cached split planes -- of the velocities, or of
// positions - mean built straight from the positions (no float32 displacement array) -- else the
// float32 slot, which in displacement mode is the materialised positions - mean array (or, when HBM
// has no room for it, the positions themselves with the subtract-while-staging kernel).
// *slot_io / *disp_io come in as the caller's slot and PSA_F_DISPLACEMENTS and go out as what the
// projection has to be launched with.
int group_source(psa_ctx* c, int* slot_io, bool* disp_io, const float* mean_host, const int* d_idx,
                        const int32_t* h_idx, int64_t n_g, int64_t K, PlaneSet** ps) {
    *ps = nullptr;
    PSA_TRY(get_planes(c, *slot_io, d_idx, h_idx, n_g, K, *disp_io ? mean_host : nullptr, ps));
    if (*ps) {
        *disp_io = false;                                         // the planes already hold slot - mean
        return PSA_OK;
    }
    return materialise_displacements(c, slot_io, disp_io, mean_host);
}


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
    c->inten_valid = false;
    c->kmap.clear();
    c->out_K = K_total;
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

}  // namespace psa

using namespace psa;

extern "C" {


int psa_sed_project(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                    int64_t K_local, int64_t K_total, int64_t k_offset, const int32_t* group_idx,
                    const int64_t* group_off, int32_t G, int32_t flags) {
    PSA_TRY(enter(c));
    Guard guard(c);
    PSA_TRY(check_slot(c, slot));
    ProjectArgs   a{slot, mean_pos_all, k_vectors, K_local, K_total, k_offset, group_idx, group_off, G, flags};
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    const bool    intensity = (flags & PSA_F_INTENSITY) != 0;
    bool          disp = (flags & PSA_F_DISPLACEMENTS) != 0;
    PSA_TRY(check_project_args(c, a, N));
    // the whole list on this device: k-vectors whose negation (or twin) is in the list are not projected
    std::vector<float>   uniq_k;
    std::vector<int32_t> kmap;
    const bool folded = K_local == K_total && k_offset == 0 && fold_k_list(c, k_vectors, K_local, &uniq_k, &kmap);
    if (folded) {
        a.k_vectors = uniq_k.data();
        a.K_local = a.K_total = K_local = K_total = (int64_t)uniq_k.size() / 3;
    }
    char*  rows = nullptr;
    size_t row_bytes = 0;
    PSA_TRY(begin_result(c, T, K_total, k_offset, intensity, &rows, &row_bytes));
    if (folded) PSA_TRY(install_kmap(c, kmap));
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
    Guard                guard(c);
    std::vector<float>   uniq_k;
    std::vector<int32_t> kmap;
    const bool           folded = k_vectors != nullptr && fold_k_list(c, k_vectors, K, &uniq_k, &kmap);
    if (folded) {
        k_vectors = uniq_k.data();
        K = (int64_t)uniq_k.size() / 3;
    }
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
    if (folded) PSA_TRY(install_kmap(c, kmap));
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

static int64_t result_K(const psa_ctx* c) { return c->kmap.empty() ? c->res_K : c->out_K; }
static size_t  result_bytes(const psa_ctx* c) {
    return c->res_intensity ? (size_t)c->res_T * result_K(c) * sizeof(float) : (size_t)c->res_T * result_K(c) * 3 * sizeof(float2);
}
static size_t intensity_bytes(const psa_ctx* c) { return (size_t)c->res_T * result_K(c) * sizeof(float); }

static int check_result_buffers(const psa_ctx* c, const void* out_host, size_t out_bytes, const float* out_intensity,
                                size_t out_intensity_bytes) {
    const size_t bytes = result_bytes(c);
    PSA_REQUIRE(out_host == nullptr || out_bytes == bytes,
                "result is %zu bytes (T=%lld, K=%lld, %s), the caller's buffer %zu", bytes, (long long)c->res_T,
                (long long)result_K(c), c->res_intensity ? "float32 intensity" : "complex64 x 3", out_bytes);
    PSA_REQUIRE(out_intensity == nullptr || !c->res_intensity,
                "out_intensity goes with a complex result; an intensity result IS out_host");
    PSA_REQUIRE(out_intensity == nullptr || out_intensity_bytes == intensity_bytes(c),
                "intensity is (%lld,%lld) float32 = %zu bytes, the caller's buffer %zu", (long long)c->res_T,
                (long long)result_K(c), intensity_bytes(c), out_intensity_bytes);
    return PSA_OK;
}

int psa_sed_finalize(psa_ctx* c, void* out_host, size_t out_bytes, float* out_intensity, size_t out_intensity_bytes) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->slab_valid) {
        set_error("psa_sed_finalize before psa_sed_project");
        return PSA_ESTATE;
    }
    const int64_t T = c->res_T, K = result_K(c);
    const size_t  bytes = result_bytes(c);
    PSA_TRY(check_result_buffers(c, out_host, out_bytes, out_intensity, out_intensity_bytes));
    PSA_TRY(c->d_out.reserve(bytes));
    const int32_t* d_map = c->kmap.empty() ? nullptr : c->d_kmap.as<int32_t>();
    {
        StageTimer st(c, PSA_T_TRANSPOSE);
        if (c->res_intensity) {
            PSA_TRY(launch_transpose_f32(c, c->d_slab.as<float>(), c->d_out.as<float>(), T, K, d_map));
        } else {
            // SED.intensity (core/sed.py:22-24) comes out of the same pass over the result
            PSA_TRY(c->d_inten.reserve(intensity_bytes(c)));
            PSA_TRY(launch_scale_transpose_c64(c, c->d_slab.as<float2>(), c->d_out.as<float2>(), c->d_inten.as<float>(), T, K, K,
                                               0, 0, nullptr, d_map));
            c->inten_valid = true;
        }
    }
    c->out_valid = true;
    if (out_host || out_intensity) {
        StageTimer st(c, PSA_T_D2H);
        if (out_host) PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_out.ptr, bytes, hipMemcpyDeviceToHost, c->stream));
        if (out_intensity)
            PSA_HIP_CHECK(hipMemcpyAsync(out_intensity, c->d_inten.ptr, intensity_bytes(c), hipMemcpyDeviceToHost, c->stream));
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
// K_rows: k-vectors projected; K_out >= K_rows: columns of the result (folded pairs copy twice the
// columns per projected vector, which moves r up).
static std::vector<int64_t> pipeline_blocks(int64_t K_rows, int64_t K_out, int64_t n_g) {
    std::vector<int64_t> b;
    const int64_t        K = K_rows;
    const double         r = 1.9e4 / (double)std::max<int64_t>(n_g, 1) * (double)K_out / (double)std::max<int64_t>(K_rows, 1);
    if (const char* e = std::getenv("PSA_PIPELINE_BLOCKS")) {       // experiments: "192,64"
        int64_t left = K;
        for (const char* q = e; *q && left > 0;) {
            const int64_t n = std::min<int64_t>(std::max<int64_t>(1, std::atoll(q)), left);
            b.push_back(n);
            left -= n;
            while (*q && *q != ',') ++q;
            if (*q == ',') ++q;
        }
        if (left > 0) b.push_back(left);
        return b;
    }
    if (K < 192) {
        b.push_back(K);
    } else if (r < 1.0) {
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

// PSA_TIMELINE=1: where a pipelined calculate spends its time -- events on both streams, printed
// (ms since entry) to stderr when the call returns.  Diagnostics only (tools/e2e_timeline.py).
struct Timeline {
    bool on = std::getenv("PSA_TIMELINE") != nullptr;
    std::chrono::steady_clock::time_point t_host0 = std::chrono::steady_clock::now();
    hipEvent_t   e0 = nullptr;
    struct Mark { const char* what; int block; hipEvent_t ev; double host_ms; };
    std::vector<Mark> marks;
    double host_ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count(); }
    void start(hipStream_t s) {
        if (!on) return;
        (void)hipEventCreate(&e0);
        (void)hipEventRecord(e0, s);
    }
    void mark(const char* what, int block, hipStream_t s) {
        if (!on) return;
        hipEvent_t ev = nullptr;
        (void)hipEventCreate(&ev);
        (void)hipEventRecord(ev, s);
        marks.push_back({what, block, ev, host_ms()});
    }
    void report() {
        if (!on) return;
        const double t_end = host_ms();
        std::fprintf(stderr, "[psa timeline] %-22s %5s %10s %10s\n", "event", "block", "device_ms", "issued_ms");
        for (auto& m : marks) {
            float ms = -1.f;
            (void)hipEventElapsedTime(&ms, e0, m.ev);
            std::fprintf(stderr, "[psa timeline] %-22s %5d %10.3f %10.3f\n", m.what, m.block, ms, m.host_ms);
            (void)hipEventDestroy(m.ev);
        }
        std::fprintf(stderr, "[psa timeline] %-22s %5s %10s %10.3f\n", "return", "", "", t_end);
        if (e0) (void)hipEventDestroy(e0);
    }
};

// Complex result of one group, all K on this device, straight to the host: per block of k-vectors
// project -> FFT -> scale/transpose (+ intensity) into its columns of (T, K, 3) -> 2-D D2H on a copy
// stream (full PCIe rate at >= 1.5-KB rows: tools/probes/d2h_2d.hip), overlapped with the next block.
// Folded lists: a block's columns are its own k-vectors' and their partners' (for a grid symmetric
// about Gamma: two runs of columns per block).
static int calculate_pipelined(psa_ctx* c, const ProjectArgs& a_in, void* out_host, float* out_intensity) {
    int           slot = a_in.slot;
    ProjectArgs   a = a_in;
    const int64_t K_out = a.K_total;
    PSA_TRY(check_slot(c, slot));
    const int64_t T = c->slot[slot].T, N = c->slot[slot].N;
    bool          disp = (a.flags & PSA_F_DISPLACEMENTS) != 0;
    PSA_TRY(check_project_args(c, a, N));
    std::vector<float>   uniq_k;
    std::vector<int32_t> kmap;
    const bool           folded = fold_k_list(c, a.k_vectors, K_out, &uniq_k, &kmap);
    if (folded) {
        a.k_vectors = uniq_k.data();
        a.K_local = a.K_total = (int64_t)uniq_k.size() / 3;
    }
    const int64_t K = a.K_total;                                   // rows of the slab
    char*  rows = nullptr;
    size_t row_bytes = 0;
    PSA_TRY(begin_result(c, T, K, 0, false, &rows, &row_bytes));
    if (folded) PSA_TRY(install_kmap(c, kmap));
    PSA_TRY(upload_project_inputs(c, a, N));
    PSA_TRY(c->d_out.reserve(result_bytes(c)));
    PSA_TRY(c->d_inten.reserve(intensity_bytes(c)));
    if (!c->d2h_stream) PSA_HIP_CHECK(hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
    if (!c->d2h_ready) PSA_HIP_CHECK(hipEventCreateWithFlags(&c->d2h_ready, hipEventDisableTiming));
    const int64_t  n_g = a.group_idx ? (a.group_off[1] - a.group_off[0]) : N;
    const int*     d_idx = a.group_idx ? c->d_idx.as<int>() : nullptr;
    const int32_t* h_idx = a.group_idx ? a.group_idx : nullptr;
    if (n_g == 0) {
        std::memset(out_host, 0, result_bytes(c));
        if (out_intensity) std::memset(out_intensity, 0, intensity_bytes(c));
        PSA_HIP_CHECK(hipMemsetAsync(c->d_out.ptr, 0, result_bytes(c), c->stream));
        PSA_HIP_CHECK(hipMemsetAsync(c->d_inten.ptr, 0, intensity_bytes(c), c->stream));
        PSA_HIP_CHECK(hipMemsetAsync(rows, 0, row_bytes * (size_t)K, c->stream));
        c->out_valid = c->inten_valid = true;
        return PSA_OK;
    }
    const std::vector<int64_t> blocks = pipeline_blocks(K, K_out, n_g);
    // columns of every block, ascending within a block: cols / srcs, block b at [first[b], first[b+1])
    std::vector<int32_t> cols, srcs;
    std::vector<size_t>  first(blocks.size() + 1, 0);
    bool                 copy_by_block = true;
    if (folded) {
        std::vector<int> block_of((size_t)K);
        {
            int64_t r0 = 0;
            for (size_t b = 0; b < blocks.size(); r0 += blocks[b], ++b)
                for (int64_t r = r0; r < r0 + blocks[b]; ++r) block_of[(size_t)r] = (int)b;
        }
        for (int64_t k = 0; k < K_out; ++k) ++first[(size_t)block_of[(size_t)(kmap[k] & ~KMAP_MIRROR)] + 1];
        for (size_t b = 0; b < blocks.size(); ++b) first[b + 1] += first[b];
        cols.resize((size_t)K_out);
        srcs.resize((size_t)K_out);
        std::vector<size_t> at(first.begin(), first.end() - 1);
        for (int64_t k = 0; k < K_out; ++k) {
            const size_t i = at[(size_t)block_of[(size_t)(kmap[k] & ~KMAP_MIRROR)]]++;
            cols[i] = (int32_t)k;
            srcs[i] = kmap[k];
        }
        for (size_t b = 0; b < blocks.size() && copy_by_block; ++b) {   // more than a few runs of columns: one copy at the end
            int runs = 0;
            for (size_t i = first[b]; i < first[b + 1]; ++i) runs += i == first[b] || cols[i] != cols[i - 1] + 1;
            copy_by_block = runs <= 8;
        }
        PSA_TRY(c->d_cols.reserve((size_t)2 * K_out * sizeof(int32_t)));
        PSA_HIP_CHECK(hipMemcpyAsync(c->d_cols.ptr, cols.data(), (size_t)K_out * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        PSA_HIP_CHECK(hipMemcpyAsync(c->d_cols.as<int32_t>() + K_out, srcs.data(), (size_t)K_out * sizeof(int32_t),
                                     hipMemcpyHostToDevice, c->stream));
    }
    PlaneSet* ps = nullptr;
    PSA_TRY(group_source(c, &slot, &disp, a.mean_pos_all, d_idx, h_idx, n_g, K, &ps));
    Timeline tl;
    tl.start(c->stream);
    const size_t pitch = (size_t)K_out * 3 * sizeof(float2);
    auto copy_columns = [&](int64_t col0, int64_t n) -> int {
        const size_t off = (size_t)col0 * 3 * sizeof(float2), width = (size_t)n * 3 * sizeof(float2);
        PSA_HIP_CHECK(hipMemcpy2DAsync((char*)out_host + off, pitch, (const char*)c->d_out.ptr + off, pitch, width, (size_t)T,
                                       hipMemcpyDeviceToHost, c->d2h_stream));
        return PSA_OK;
    };
    int64_t k0 = 0;
    for (size_t b = 0; b < blocks.size(); ++b) {
        const int64_t nk = blocks[b];
        float2*       d_q = (float2*)(rows + row_bytes * (size_t)k0);
        ProjGeom      g;
        PSA_TRY(make_geom(c, slot, nk, n_g, d_idx, h_idx, disp, ps, 0, &g));
        PSA_TRY(prepare_phase(c, d_idx, g, disp, k0));
        PSA_TRY(launch_projection(c, slot, d_idx, g, disp, ps, d_q, T, 0, T));
        tl.mark("projected", (int)b, c->stream);
        {
            StageTimer st(c, PSA_T_FFT);
            PSA_TRY(run_fft(c, d_q, T, 3 * nk));
        }
        tl.mark("fft done", (int)b, c->stream);
        {
            StageTimer st(c, PSA_T_TRANSPOSE);
            if (folded)
                PSA_TRY(launch_scale_transpose_c64(c, (const float2*)rows, c->d_out.as<float2>(), c->d_inten.as<float>(), T,
                                                   (int64_t)(first[b + 1] - first[b]), K_out, 0, 0,
                                                   c->d_cols.as<int32_t>() + first[b], c->d_cols.as<int32_t>() + K_out + first[b]));
            else
                PSA_TRY(launch_scale_transpose_c64(c, (const float2*)rows, c->d_out.as<float2>(), c->d_inten.as<float>(), T, nk,
                                                   K_out, k0, k0, nullptr, nullptr));
        }
        tl.mark("transposed", (int)b, c->stream);
        if (copy_by_block) {
            PSA_HIP_CHECK(hipEventRecord(c->d2h_ready, c->stream));
            PSA_HIP_CHECK(hipStreamWaitEvent(c->d2h_stream, c->d2h_ready, 0));
            tl.mark("copy can start", (int)b, c->d2h_stream);
            if (!folded) {
                PSA_TRY(copy_columns(k0, nk));
            } else {
                for (size_t i = first[b]; i < first[b + 1];) {
                    size_t j = i + 1;
                    while (j < first[b + 1] && cols[j] == cols[j - 1] + 1) ++j;
                    PSA_TRY(copy_columns(cols[i], (int64_t)(j - i)));
                    i = j;
                }
            }
            tl.mark("copied", (int)b, c->d2h_stream);
        }
        k0 += nk;
    }
    c->out_valid = c->inten_valid = true;
    if (!copy_by_block || out_intensity) {
        PSA_HIP_CHECK(hipEventRecord(c->d2h_ready, c->stream));
        PSA_HIP_CHECK(hipStreamWaitEvent(c->d2h_stream, c->d2h_ready, 0));
        if (!copy_by_block)
            PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_out.ptr, result_bytes(c), hipMemcpyDeviceToHost, c->d2h_stream));
        if (out_intensity)
            PSA_HIP_CHECK(hipMemcpyAsync(out_intensity, c->d_inten.ptr, intensity_bytes(c), hipMemcpyDeviceToHost, c->d2h_stream));
        tl.mark("intensity copied", -1, c->d2h_stream);
    }
    PSA_HIP_CHECK(hipStreamSynchronize(c->d2h_stream));
    PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    tl.report();
    return PSA_OK;
}

int psa_sed_calculate(psa_ctx* c, int slot, const float* mean_pos_all, const float* k_vectors,
                      int64_t K, const int32_t* group_idx, const int64_t* group_off, int32_t G,
                      int32_t flags, void* out_host, size_t out_bytes, float* out_intensity, size_t out_intensity_bytes) {
    PSA_REQUIRE(K >= 1, "need at least one k-vector");
    PSA_REQUIRE(out_intensity == nullptr || !(flags & PSA_F_INTENSITY),
                "out_intensity goes with a complex result; an intensity result IS out_host");
    if (out_host && !(flags & PSA_F_INTENSITY) && G == 1 && K >= 192 && c && c->k1_selector == PSA_K1_AUTO) {
        PSA_TRY(enter(c));
        Guard guard(c);
        PSA_TRY(check_slot(c, slot));
        const size_t bytes = (size_t)c->slot[slot].T * K * 3 * sizeof(float2);
        PSA_REQUIRE(k_vectors != nullptr, "null k_vectors");
        PSA_REQUIRE(out_bytes == bytes, "result is %zu bytes (T=%lld, K=%lld, complex64 x 3), the caller's buffer %zu", bytes,
                    (long long)c->slot[slot].T, (long long)K, out_bytes);
        PSA_REQUIRE(out_intensity == nullptr || out_intensity_bytes == bytes / 6,
                    "intensity is (%lld,%lld) float32 = %zu bytes, the caller's buffer %zu", (long long)c->slot[slot].T,
                    (long long)K, bytes / 6, out_intensity_bytes);
        const ProjectArgs a{slot, mean_pos_all, k_vectors, K, K, 0, group_idx, group_off, G, flags};
        return calculate_pipelined(c, a, out_host, out_intensity);
    }
    PSA_TRY(psa_sed_project(c, slot, mean_pos_all, k_vectors, K, K, 0, group_idx, group_off, G, flags));
    return psa_sed_finalize(c, out_host, out_bytes, out_intensity, out_intensity_bytes);
}

// The k map of a result whose rows were projected from a folded list by the caller (a sharded run:
// psa_amd/dist.py folds, shards the unique vectors, and installs the map on the ranks that finalize).
int psa_sed_set_kmap(psa_ctx* c, const int32_t* kmap, int64_t K_out) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->slab_valid) {
        set_error("psa_sed_set_kmap before psa_sed_project");
        return PSA_ESTATE;
    }
    PSA_REQUIRE(K_out >= 0 && K_out < (1ll << 30) && (kmap != nullptr || K_out == 0), "bad k map");
    for (int64_t k = 0; k < K_out; ++k)
        PSA_REQUIRE((int64_t)(kmap[k] & ~KMAP_MIRROR) < c->res_K, "k map entry %lld points past the slab's %lld rows",
                    (long long)k, (long long)c->res_K);
    return install_kmap(c, std::vector<int32_t>(kmap, kmap + K_out));
}

int psa_k_pairs(const float* k_vectors, int64_t K, int32_t* kmap, int32_t* unique_idx, int64_t* n_unique) {
    PSA_REQUIRE(K >= 0 && K < (1ll << 30) && (K == 0 || (k_vectors && kmap && unique_idx)) && n_unique, "bad argument");
    std::vector<int32_t> m, u;
    fold_pairs(k_vectors, K, &m, &u);
    if (K) std::memcpy(kmap, m.data(), (size_t)K * sizeof(int32_t));
    if (!u.empty()) std::memcpy(unique_idx, u.data(), u.size() * sizeof(int32_t));
    *n_unique = (int64_t)u.size();
    return PSA_OK;
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
    c->out_valid = c->inten_valid = false;
    return PSA_OK;
}

int psa_result_intensity(psa_ctx* c, float* out_host, size_t out_bytes) {
    PSA_TRY(enter(c));
    Guard guard(c);
    if (!c->out_valid || c->res_intensity) {
        set_error("psa_result_intensity needs a finalized complex result");
        return PSA_ESTATE;
    }
    const int64_t n = c->res_T * result_K(c);
    PSA_REQUIRE(out_host == nullptr || out_bytes == (size_t)n * sizeof(float),
                "result is (%lld,%lld) float32 = %zu bytes, the caller's buffer %zu", (long long)c->res_T,
                (long long)result_K(c), (size_t)n * sizeof(float), out_bytes);
    if (!c->inten_valid) {                      // (finalize and calculate leave it behind; kept for results placed otherwise)
        PSA_TRY(c->d_inten.reserve((size_t)n * sizeof(float)));
        StageTimer st(c, PSA_T_EPILOGUE);
        PSA_TRY(launch_result_intensity(c, c->d_out.as<float2>(), c->d_inten.as<float>(), n));
        c->inten_valid = true;
    }
    if (out_host) {
        PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_inten.ptr, (size_t)n * sizeof(float),
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
    const int64_t n = c->res_T * result_K(c);
    PSA_REQUIRE(out_host == nullptr || out_bytes == (size_t)n * sizeof(float),
                "result is (%lld,%lld) float32 = %zu bytes, the caller's buffer %zu", (long long)c->res_T,
                (long long)result_K(c), (size_t)n * sizeof(float), out_bytes);
    PSA_TRY(c->d_aux.reserve((size_t)n * sizeof(float)));
    PSA_TRY(launch_result_chiral_c(c, c->d_out.as<float2>(), c->d_aux.as<float>(), n, c1, c2));
    if (out_host) {
        PSA_HIP_CHECK(hipMemcpyAsync(out_host, c->d_aux.ptr, (size_t)n * sizeof(float),
                                     hipMemcpyDeviceToHost, c->stream));
        PSA_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
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

}  // extern "C"
