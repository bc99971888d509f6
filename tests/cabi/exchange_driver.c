/* A plain-C consumer of the MULTI-GPU interface of include/makisu_mi.h in its single-process form -- what
 * INTEGRATION.md shows for a Go host ("Multi-GPU from Go"): one ctx per device, one host thread per device for the
 * scan, mi_comm_init_all once, mi_dedup_allgather_all per round of batches (MI_EXCHANGE_FORM=alltoall in the environment:
 * mi_dedup_alltoall_all, the hash-partitioned form of the same exchange).
 * Usage: exchange_driver <n_ranks> <files_per_rank> [device ...]      (no devices given: rank r on device r)
 * Rank r scans files_per_rank synthetic 64 KiB files whose content ids are  r * files_per_rank / 2 + i  -- every
 * rank's first half repeats the previous rank's second half, so half of the job's chunks are cross-rank duplicates.
 * Prints per rank:  K <rank> <n_chunks> <first_global_of_its_rows> <rows with dup_of >= 0> <gather ms> <marking ms>
 * per chunk:        C <rank> <dup_of> sha256:<digest>
 * and a last line:  T <n_total> <n_unique> <rccl_ranks>
 * Built and run (on the RCCL test double: n ranks on one GPU) by tests/test_gpu_native_exchange.py::test_plain_c_exchange. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "makisu_mi.h"

#define MAXR 16
static mi_ctx* ctx[MAXR];
static mi_batch* batch[MAXR];
static int n_ranks, files_per_rank, failed[MAXR];

static void* scan(void* arg) {                                   /* one host thread per device */
    const int r = (int)(size_t)arg;
    uint64_t* sizes = malloc(sizeof(uint64_t) * (size_t)files_per_rank);
    uint64_t* cids = malloc(sizeof(uint64_t) * (size_t)files_per_rank);
    for (int i = 0; i < files_per_rank; i++) {
        sizes[i] = 65536;
        cids[i] = (uint64_t)r * (uint64_t)(files_per_rank / 2) + (uint64_t)i;
    }
    if (mi_batch_begin(ctx[r], (uint64_t)files_per_rank, 0, &batch[r]) != MI_OK ||
        mi_batch_add_synthetic(batch[r], (uint64_t)files_per_rank, sizes, cids, 0x4D414B49ull) != MI_OK ||
        mi_batch_run(batch[r]) != MI_OK) {
        fprintf(stderr, "exchange: rank %d: %s\n", r, mi_last_error(ctx[r]));
        failed[r] = 1;
    }
    free(sizes);
    free(cids);
    return NULL;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    n_ranks = atoi(argv[1]);
    files_per_rank = atoi(argv[2]);
    if (n_ranks < 1 || n_ranks > MAXR || files_per_rank < 2) return 2;
    for (int r = 0; r < n_ranks; r++) {
        mi_config cfg;
        mi_config_default(&cfg);
        cfg.device = argc > 3 ? atoi(argv[3 + (r < argc - 3 ? r : argc - 4)]) : r;
        cfg.flags |= MI_FLAG_NO_DEDUP;                           /* the job-wide marking supersedes the in-batch one */
        if (mi_ctx_create(&cfg, &ctx[r]) != MI_OK) { fprintf(stderr, "exchange: %s\n", mi_last_error(NULL)); return 2; }
    }
    if (mi_comm_init_all(ctx, n_ranks) != MI_OK) { fprintf(stderr, "exchange: %s\n", mi_last_error(ctx[0])); return 1; }
    pthread_t th[MAXR];
    for (int r = 0; r < n_ranks; r++) pthread_create(&th[r], NULL, scan, (void*)(size_t)r);
    for (int r = 0; r < n_ranks; r++) pthread_join(th[r], NULL);
    for (int r = 0; r < n_ranks; r++) if (failed[r]) return 1;
    uint64_t n_total = 0, n_unique = 0;
    const char* form = getenv("MI_EXCHANGE_FORM");
    int (*exchange)(mi_batch**, int, uint64_t*, uint64_t*) =
        form && !strcmp(form, "alltoall") ? mi_dedup_alltoall_all : mi_dedup_allgather_all;
    for (int round = 0; round < 2; round++)                      /* twice: the exchange buffers are reused */
        if (exchange(batch, n_ranks, &n_total, &n_unique) != MI_OK) {
            for (int r = 0; r < n_ranks; r++) fprintf(stderr, "exchange: rank %d: %s\n", r, mi_last_error(ctx[r]));
            return 1;
        }
    uint64_t first = 0;
    int ranks_seen = 0;
    for (int r = 0; r < n_ranks; r++) {
        uint64_t nf = 0, nc = 0, nb = 0, dups = 0;
        double gather = 0, marking = 0;
        if (mi_batch_counts(batch[r], &nf, &nc, &nb) != MI_OK) return 1;
        mi_chunk_result* ch = calloc(nc ? nc : 1, sizeof *ch);
        if (mi_batch_chunks(batch[r], ch, nc) != MI_OK) { fprintf(stderr, "exchange: %s\n", mi_last_error(ctx[r])); return 1; }
        for (uint64_t c = 0; c < nc; c++) dups += ch[c].dup_of >= 0;
        mi_comm_exchange_ms(ctx[r], &gather, &marking);
        mi_comm_ranks(ctx[r], &ranks_seen);
        printf("K %d %llu %llu %llu %.3f %.3f\n", r, (unsigned long long)nc, (unsigned long long)first,
               (unsigned long long)dups, gather, marking);
        for (uint64_t c = 0; c < nc; c++) {
            printf("C %d %lld sha256:", r, (long long)ch[c].dup_of);
            for (int i = 0; i < 32; i++) printf("%02x", ch[c].sha256[i]);
            printf("\n");
        }
        first += nc;
        free(ch);
    }
    printf("T %llu %llu %d\n", (unsigned long long)n_total, (unsigned long long)n_unique, ranks_seen);
    for (int r = 0; r < n_ranks; r++) {
        mi_batch_free(batch[r]);
        mi_comm_destroy(ctx[r]);
        if (mi_ctx_destroy(ctx[r]) != MI_OK) return 1;
    }
    return 0;
}
