/* A plain C99 client of include/dptm.h: what a non-Python host would write.  Reads a flat weight blob and
 * the five input arrays from one binary file, scores them through dptm_score and dptm_score_device-free
 * host entry points, and writes the probabilities.  Built with gcc (no HIP headers needed) and linked
 * against libdptm.so by tests/test_gpu_c_client.py.
 *
 * file layout (little endian): int64 n_weights, int64 n_sites, float weights[n_weights],
 *   uint8 tok21[n][21], uint8 tok33[n][33], uint8 tok51[n][51], float strct[n][8], float esm[n][1280] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "dptm.h"

static void* read_block(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s input.bin output.bin\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t hdr[2];
    if (fread(hdr, sizeof(int64_t), 2, f) != 2) return 2;
    const int64_t nw = hdr[0], n = hdr[1];
    float* w = (float*)read_block(f, (size_t)nw * 4);
    uint8_t* t21 = (uint8_t*)read_block(f, (size_t)n * 21);
    uint8_t* t33 = (uint8_t*)read_block(f, (size_t)n * 33);
    uint8_t* t51 = (uint8_t*)read_block(f, (size_t)n * 51);
    float* st = (float*)read_block(f, (size_t)n * 8 * 4);
    float* esm = (float*)read_block(f, (size_t)n * 1280 * 4);
    fclose(f);

    dptm_spec spec = {3, {21, 33, 51}, 21, 64, 4, 256, 1280, 8, 128, 512, 256, 512, 256, 2};
    if (dptm_weight_count(&spec) != nw) {
        fprintf(stderr, "weight count mismatch: %lld\n", (long long)dptm_weight_count(&spec));
        return 3;
    }
    dptm_handle* h = NULL;
    if (dptm_create(w, (size_t)nw * 4, &spec, 0, &h) != DPTM_OK) {
        fprintf(stderr, "dptm_create: %s\n", dptm_last_error());
        return 4;
    }
    float* out = (float*)malloc((size_t)n * 2 * 4 + 4);
    if (dptm_score(h, t21, t33, t51, st, esm, n, out) != DPTM_OK) {
        fprintf(stderr, "dptm_score: %s\n", dptm_last_error());
        return 5;
    }
    /* error behaviour: a NULL buffer is rejected with a message, the handle stays usable */
    if (dptm_score(h, NULL, t33, t51, st, esm, n, out) == DPTM_OK || dptm_last_error()[0] == 0) return 6;
    if (dptm_set_precision(h, 99) == DPTM_OK) return 7;
    if (dptm_set_precision(h, DPTM_PREC_F32) != DPTM_OK) return 8;
    dptm_destroy(h);
    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 4, (size_t)n * 2, f) != (size_t)n * 2) return 9;
    fclose(f);
    printf("scored %lld sites\n", (long long)n);
    return 0;
}
