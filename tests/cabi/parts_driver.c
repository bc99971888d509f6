/* A plain-C consumer of the PARTS interface of include/makisu_mi.h -- the loop INTEGRATION.md
 * shows for a Go host ("One huge file across the GPUs"), here with one ctx and one batch per part
 * standing in for the GPUs.
 * Usage: parts_driver <file> <n_parts>
 * Prints one line per chunk, parts in order:  C <offset> <length> sha256:<digest>
 * and a last line:  R <rounds>.  Built and run by tests/test_gpu_parts.py::test_plain_c_parts. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "makisu_mi.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != MI_OK) {                                                          \
            fprintf(stderr, "parts: %s: %s\n", #call, mi_last_error(ctx));           \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    const char* path = argv[1];
    int n = atoi(argv[2]);
    struct stat sb;
    if (stat(path, &sb) != 0 || n < 1 || n > 64) { perror(path); return 2; }
    const uint64_t size = (uint64_t)sb.st_size, groups = (size + MI_PART_ALIGN - 1) / MI_PART_ALIGN;
    if ((uint64_t)n > groups) n = (int)groups;
    mi_ctx* ctx = NULL;
    mi_config cfg;
    mi_config_default(&cfg);
    if (mi_ctx_create(&cfg, &ctx) != MI_OK) { fprintf(stderr, "parts: %s\n", mi_last_error(NULL)); return 2; }
    mi_batch* b[64];
    for (int k = 0; k < n; k++) {
        const uint64_t begin = groups * (uint64_t)k / (uint64_t)n * MI_PART_ALIGN;
        const uint64_t end = k + 1 == n ? size : groups * (uint64_t)(k + 1) / (uint64_t)n * MI_PART_ALIGN;
        CHECK(mi_batch_begin(ctx, 1, 0, &b[k]));
        CHECK(mi_batch_add_path_part(b[k], path, size, begin, end, (uint64_t)k));
        CHECK(mi_batch_scan_cuts(b[k]));
    }
    int rounds = 0, changed = 1;
    while (changed) {
        changed = 0;
        rounds++;
        for (int k = 1; k < n; k++) {
            mi_part_state prev, mine;
            CHECK(mi_batch_parts(b[k - 1], &prev, 1, NULL));
            CHECK(mi_batch_parts(b[k], &mine, 1, NULL));
            if (mine.entry != prev.exit) changed = 1;
            if (mine.entry != prev.exit || !mine.entry_confirmed)
                CHECK(mi_batch_set_part_entry(b[k], mine.file_index, prev.exit));
            CHECK(mi_batch_fix_cuts(b[k]));      /* before part k + 1 reads this part's exit */
        }
    }
    for (int k = 0; k < n; k++) {
        uint64_t nf = 0, nc = 0, nb = 0;
        CHECK(mi_batch_run(b[k]));
        CHECK(mi_batch_counts(b[k], &nf, &nc, &nb));
        mi_chunk_result* ch = calloc(nc ? nc : 1, sizeof *ch);
        CHECK(mi_batch_chunks(b[k], ch, nc));
        for (uint64_t c = 0; c < nc; c++) {
            printf("C %llu %u sha256:", (unsigned long long)ch[c].offset, ch[c].length);
            for (int i = 0; i < 32; i++) printf("%02x", ch[c].sha256[i]);
            printf("\n");
        }
        free(ch);
        CHECK(mi_batch_free(b[k]));
    }
    printf("R %d\n", rounds);
    return mi_ctx_destroy(ctx) == MI_OK ? 0 : 1;
}
