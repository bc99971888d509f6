/* The content-aware commit from plain C (C99, no C++ runtime on this side): what a cgo shim does, as a C program.
 *   commit_driver <root> <victim file below root>
 * Two MemFS handles on <root> -- one committing with a ctx (the GPU scan inside), one without (the reference's commit) --
 * and ONE library call per commit: mi_memfs_commit_layer.  Prints
 *   C0 <TarDigest hex> entries=N            first commit with the ctx: everything is new
 *   P0 <TarDigest hex> entries=N            the same without a ctx (the same digest: the same tar)
 *   C1 <TarDigest hex> entries=0            nothing changed
 *   -- the victim is rewritten: same size, same mtime second, other bytes --
 *   C2 <TarDigest hex> entries=N content_changed=K scanned=F opened=F' read=B
 *   P2 <TarDigest hex> entries=0            tario.IsSimilarHeader does not see it (lib/tario/compare.go:101-103)
 *   N0 <TarDigest hex> entries=N ctxs=3 verified=F     a third handle, three ctxs on the device: mi_memfs_commit_layer_n over the tree as
 *                                                      it is now -- the tar of a header-only commit of that tree (N0p)
 * Test infrastructure (tests/test_gpu_commit.py). */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "makisu_mi.h"

static void hex32(const uint8_t* d, char* out) {
    static const char* x = "0123456789abcdef";
    for (int i = 0; i < 32; ++i) { out[2 * i] = x[d[i] >> 4]; out[2 * i + 1] = x[d[i] & 15]; }
    out[64] = 0;
}

static int commit(mi_memfs* fs, mi_ctx* ctx, const char* tag, int with_stats) {
    mi_layer_config cfg;
    mi_layer_config_default(&cfg);
    cfg.gzip_level = MI_GZIP_OFF;
    mi_layer_result res;
    int done = 0;
    int rc = mi_memfs_commit_layer(fs, ctx, 1, NULL, 0, &cfg, &res, NULL, &done);
    if (rc || !done) {
        fprintf(stderr, "%s: mi_memfs_commit_layer = %d: %s\n", tag, rc, mi_memfs_error(fs));
        return 1;
    }
    char hex[65];
    hex32(res.tar_sha256, hex);
    if (!with_stats) {
        printf("%s %s entries=%llu\n", tag, hex, (unsigned long long)res.n_entries);
        return 0;
    }
    mi_commit_stats st;
    if (mi_memfs_commit_stats(fs, &st)) return 1;
    printf("%s %s entries=%llu content_changed=%llu scanned=%llu opened=%llu read=%llu\n", tag, hex,
           (unsigned long long)res.n_entries, (unsigned long long)st.n_content_changed,
           (unsigned long long)st.n_scanned_files, (unsigned long long)st.files_opened,
           (unsigned long long)st.file_bytes_read);
    return 0;
}

static int rewrite_same_size_same_second(const char* path) {
    struct stat st;
    if (stat(path, &st)) return 1;
    int fd = open(path, O_RDWR);
    if (fd < 0) return 1;
    uint8_t* buf = (uint8_t*)malloc((size_t)st.st_size + 1);
    if (pread(fd, buf, (size_t)st.st_size, 0) != st.st_size) return 1;
    for (off_t i = 0; i < st.st_size; ++i) buf[i] ^= 0x5a;
    if (pwrite(fd, buf, (size_t)st.st_size, 0) != st.st_size) return 1;
    free(buf);
    struct timespec ts[2] = {st.st_atim, st.st_mtim};
    if (futimens(fd, ts)) return 1;
    close(fd);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    mi_config cfg;
    mi_config_default(&cfg);
    mi_ctx* ctx = NULL;
    if (mi_ctx_create(&cfg, &ctx)) { fprintf(stderr, "mi_ctx_create: %s\n", mi_last_error(NULL)); return 1; }
    mi_memfs *gpu = NULL, *plain = NULL;
    if (mi_memfs_create(argv[1], NULL, 0, 0, &gpu) || mi_memfs_create(argv[1], NULL, 0, 0, &plain)) return 1;
    int bad = commit(gpu, ctx, "C0", 0) || commit(plain, NULL, "P0", 0) || commit(gpu, ctx, "C1", 0);
    if (!bad) bad = rewrite_same_size_same_second(argv[2]);
    if (!bad) bad = commit(gpu, ctx, "C2", 1) || commit(plain, NULL, "P2", 0);
    /* the same call over several ctxs (one per GPU on a real node; three on the one device here) */
    mi_ctx* more[3] = {ctx, NULL, NULL};
    mi_memfs *many = NULL, *fresh = NULL;
    if (!bad) bad = mi_ctx_create(&cfg, &more[1]) || mi_ctx_create(&cfg, &more[2]) || mi_memfs_create(argv[1], NULL, 0, 0, &many) ||
                    mi_memfs_create(argv[1], NULL, 0, 0, &fresh);
    if (!bad) {
        mi_layer_config lc;
        mi_layer_config_default(&lc);
        lc.gzip_level = MI_GZIP_OFF;
        mi_layer_result res;
        int done = 0;
        mi_commit_stats st;
        char hex[65];
        if (mi_memfs_commit_layer_n(many, more, 3, 1, NULL, 0, &lc, &res, NULL, &done) || !done || mi_memfs_commit_stats(many, &st)) {
            fprintf(stderr, "N0: mi_memfs_commit_layer_n: %s\n", mi_memfs_error(many));
            bad = 1;
        } else {
            hex32(res.tar_sha256, hex);
            printf("N0 %s entries=%llu ctxs=%llu verified=%llu\n", hex, (unsigned long long)res.n_entries, (unsigned long long)st.n_ctxs,
                   (unsigned long long)st.n_verified_files);
            bad = commit(fresh, NULL, "N0p", 0);
        }
    }
    if (many) mi_memfs_free(many);
    if (fresh) mi_memfs_free(fresh);
    for (int i = 1; i < 3; ++i) if (more[i] && mi_ctx_destroy(more[i])) { fprintf(stderr, "mi_ctx_destroy: %s\n", mi_last_error(more[i])); bad = 1; }
    uint8_t root[32];
    int has = 0;
    const char* rel = argv[2] + strlen(argv[1]);
    if (!bad && (mi_memfs_root_of(gpu, rel, root, &has) || !has)) { fprintf(stderr, "no root for %s\n", rel); bad = 1; }
    mi_memfs_free(gpu);                      /* before the ctx: the handle keeps a batch of it */
    mi_memfs_free(plain);
    if (mi_ctx_destroy(ctx)) { fprintf(stderr, "mi_ctx_destroy: %s\n", mi_last_error(ctx)); return 1; }
    return bad;
}
