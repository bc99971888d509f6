/* A plain-C consumer of include/makisu_mi.h -- what the cgo shim compiles down to.
 * Usage: driver <file>...   Scans the files in one batch and prints, per file,
 *   F <index> <size> <n_chunks> sha256:<chunk_root> sha256:<file_sha256> <crc32 %x>
 * and per chunk
 *   C <file_index> <offset> <length> <dup_of> sha256:<digest>
 * Built and run by tests/test_gpu_parity.py::test_plain_c_consumer. */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <sys/stat.h>

#include "makisu_mi.h"

static void hex(const uint8_t* d, char* out) {
    static const char* x = "0123456789abcdef";
    for (int i = 0; i < 32; i++) { out[2 * i] = x[d[i] >> 4]; out[2 * i + 1] = x[d[i] & 15]; }
    out[64] = 0;
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != MI_OK) {                                                          \
            fprintf(stderr, "gpu scan: %s: %s\n", #call, mi_last_error(ctx));        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    mi_ctx* ctx = NULL;
    mi_config cfg;
    mi_config_default(&cfg);
    cfg.flags = MI_FLAG_FILE_SHA256 | MI_FLAG_FILE_CRC32;
    if (mi_ctx_create(&cfg, &ctx) != MI_OK) {
        fprintf(stderr, "gpu scan: %s\n", mi_last_error(NULL));
        return 2;
    }
    mi_batch* b = NULL;
    CHECK(mi_batch_begin(ctx, (uint64_t)(argc - 1), 0, &b));
    for (int i = 1; i < argc; i++) {
        struct stat st;
        if (stat(argv[i], &st) != 0) { perror(argv[i]); return 1; }
        CHECK(mi_batch_add_path(b, argv[i], (uint64_t)st.st_size, (uint64_t)i));
    }
    CHECK(mi_batch_run(b));
    uint64_t nf = 0, nc = 0, nb = 0;
    CHECK(mi_batch_counts(b, &nf, &nc, &nb));
    mi_file_result* files = calloc(nf ? nf : 1, sizeof *files);
    mi_chunk_result* chunks = calloc(nc ? nc : 1, sizeof *chunks);
    CHECK(mi_batch_files(b, files, nf));
    CHECK(mi_batch_chunks(b, chunks, nc));
    char h1[65], h2[65];
    for (uint64_t f = 0; f < nf; f++) {
        hex(files[f].chunk_root, h1);
        hex(files[f].file_sha256, h2);
        printf("F %llu %llu %u sha256:%s sha256:%s %x\n", (unsigned long long)f,
               (unsigned long long)files[f].size, files[f].n_chunks, h1, h2, files[f].crc32);
    }
    for (uint64_t c = 0; c < nc; c++) {
        hex(chunks[c].sha256, h1);
        printf("C %llu %llu %u %lld sha256:%s\n", (unsigned long long)chunks[c].file_index,
               (unsigned long long)chunks[c].offset, chunks[c].length, (long long)chunks[c].dup_of, h1);
    }
    /* the chunk index across batches: first pass everything is new, second pass all known */
    mi_index* idx = NULL;
    CHECK(mi_index_create(ctx, 0, &idx));
    for (int pass = 0; pass < 2; pass++) {
        uint64_t n_new = 0, n_known = 0, held = 0;
        CHECK(mi_index_add_batch(idx, b, NULL, 0, &n_new, &n_known));
        CHECK(mi_index_count(idx, &held));
        printf("I %d %llu %llu %llu\n", pass, (unsigned long long)n_new, (unsigned long long)n_known,
               (unsigned long long)held);
    }
    mi_index_free(idx);
    /* tario.IsSimilarHeader on two entries that differ only in content */
    if (nf >= 2) {
        mi_tree_entry ea, eb;
        memset(&ea, 0, sizeof ea);
        ea.relpath = "etc/passwd";
        ea.size = 5; ea.mtime_sec = 100; ea.mode = 0100644; ea.kind = 1;
        eb = ea;
        int same_meta = 0, same_content = 0;
        CHECK(mi_entry_similar(&ea, &eb, 0, NULL, NULL, &same_meta));
        CHECK(mi_entry_similar(&ea, &eb, 0, files[0].chunk_root, files[1].chunk_root, &same_content));
        printf("S %d %d\n", same_meta, same_content);
    }
    mi_stats st;
    CHECK(mi_get_stats(ctx, &st));
    fprintf(stderr, "scanned %llu bytes in %llu files -> %llu chunks (%llu unique), %.3f ms on the GPU\n",
            (unsigned long long)st.bytes_in, (unsigned long long)st.n_files,
            (unsigned long long)st.n_chunks, (unsigned long long)st.n_unique, st.ms_total);
    free(files);
    free(chunks);
    CHECK(mi_batch_free(b));
    mi_ctx_destroy(ctx);
    return 0;
}
