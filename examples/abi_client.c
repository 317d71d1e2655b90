/* A plain C99 client of the C ABI (include/psa_hip.h): what a host program in any language with a C
 * FFI does to run the SED hot path -- no Python, no C++, no HIP headers on this side.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/abi_client.c -Lpsa_amd/csrc -lpsa_hip \
 *       -Wl,-rpath,$PWD/psa_amd/csrc -o /tmp/abi_client
 *   /tmp/abi_client in.bin out.bin
 *
 * in.bin : int64 T, N, K, n_idx (little endian), then float32 positions (T,N,3), velocities (T,N,3),
 *          k_vectors (K,3), int32 idx (n_idx; 0 = all atoms)
 * out.bin: complex64 sed (T,K,3), float32 intensity (T,K)
 *
 * The calls follow SEDCalculator.calculate (reference src/psa/core/sed_calculator.py:182-336) for one
 * coherent group: mean positions (:205), then phase table, projection, FFT (:58-84) in
 * psa_sed_calculate, which also returns SED.intensity (core/sed.py:22-24) of the result.
 * tests/test_gpu_abi_client.py builds it, runs it on the GPU and compares with the oracle. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "psa_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != PSA_OK) {                                                               \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, psa_last_error());               \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static int read_exact(FILE* f, void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 2;
    }
    if (psa_abi_version() != PSA_HIP_ABI_VERSION) {
        fprintf(stderr, "library ABI %d, header ABI %d\n", psa_abi_version(), PSA_HIP_ABI_VERSION);
        return 1;
    }
    FILE* in = fopen(argv[1], "rb");
    if (!in) return perror(argv[1]), 1;
    int64_t dims[4];
    if (read_exact(in, dims, sizeof dims)) return fprintf(stderr, "short header\n"), 1;
    const int64_t T = dims[0], N = dims[1], K = dims[2], n_idx = dims[3];
    const size_t  traj_floats = (size_t)T * N * 3;
    float*   pos = malloc(traj_floats * sizeof(float));
    float*   vel = malloc(traj_floats * sizeof(float));
    float*   kv = malloc((size_t)K * 3 * sizeof(float));
    int32_t* idx = n_idx ? malloc((size_t)n_idx * sizeof(int32_t)) : NULL;
    float*   mean = malloc((size_t)N * 3 * sizeof(float));
    if (!pos || !vel || !kv || !mean || (n_idx && !idx)) return fprintf(stderr, "out of memory\n"), 1;
    if (read_exact(in, pos, traj_floats * sizeof(float)) || read_exact(in, vel, traj_floats * sizeof(float)) ||
        read_exact(in, kv, (size_t)K * 3 * sizeof(float)) || (n_idx && read_exact(in, idx, (size_t)n_idx * sizeof(int32_t))))
        return fprintf(stderr, "short input\n"), 1;
    fclose(in);

    /* mean_pos_all = np.mean(positions, axis=0, dtype=float32), bit for bit (sed_calculator.py:205) */
    CHECK(psa_host_mean_frames(pos, T, N * 3, mean, 4));

    int n_dev = 0;
    CHECK(psa_device_count(&n_dev));
    if (n_dev < 1) return fprintf(stderr, "no GPU\n"), 1;
    psa_ctx* ctx = NULL;
    CHECK(psa_create(0, &ctx));
    CHECK(psa_data_upload(ctx, 0, vel, T, N));             /* the trajectory stays in HBM for later calls */

    /* results into page-locked memory: the copy out runs at the full link rate */
    const size_t sed_bytes = (size_t)T * K * 3 * 2 * sizeof(float), inten_bytes = (size_t)T * K * sizeof(float);
    void *sed = NULL, *inten = NULL;
    CHECK(psa_host_alloc(sed_bytes, &sed));
    CHECK(psa_host_alloc(inten_bytes, &inten));
    const int64_t group_off[2] = {0, n_idx};
    CHECK(psa_sed_calculate(ctx, 0, mean, kv, K, idx, idx ? group_off : NULL, 1, 0, sed, sed_bytes, (float*)inten,
                            inten_bytes));

    /* a buffer of the wrong size is refused, nothing is copied */
    if (psa_sed_finalize(ctx, sed, sed_bytes - 8, NULL, 0) != PSA_EINVAL) return fprintf(stderr, "size check missing\n"), 1;

    FILE* out = fopen(argv[2], "wb");
    if (!out) return perror(argv[2]), 1;
    if (fwrite(sed, 1, sed_bytes, out) != sed_bytes || fwrite(inten, 1, inten_bytes, out) != inten_bytes)
        return fprintf(stderr, "short write\n"), 1;
    fclose(out);
    double ms[8];
    CHECK(psa_last_timings(ctx, ms));
    printf("T %lld N %lld K %lld: project %.3f ms, fft %.3f ms\n", (long long)T, (long long)N, (long long)K, ms[2], ms[3]);
    CHECK(psa_host_free(sed));
    CHECK(psa_host_free(inten));
    CHECK(psa_destroy(ctx));
    free(pos), free(vel), free(kv), free(idx), free(mean);
    return 0;
}
