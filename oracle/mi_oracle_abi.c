/*
 * mi_oracle_abi.c -- the oracle behind the SAME signatures as the product's C ABI
 * (include/makisu_mi.h), prefix mi_ref_ instead of mi_ ("CPU-oracle twins with
 * identical signatures so the parity harness calls both").
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/.  The twins cover the batch path
 * (config -> ctx -> batch -> add bytes / path -> run -> file and chunk rows), duplicate marking over
 * a digest array and the standalone digests; they share the product's structs (mi_config,
 * mi_file_result, mi_chunk_result) by including its public header, and compute with the oracle's
 * plain-C routines (mi_oracle.c).  "Device" pointers of the product ABI are host pointers here.
 */
#define _GNU_SOURCE
#include "../include/makisu_mi.h"
#include "mi_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

struct mi_ref_ctx { mi_config cfg; char err[256]; };
struct mi_ref_batch {
    struct mi_ref_ctx* ctx;
    uint8_t* data; uint64_t used, cap;
    uint64_t* off; uint64_t* size; uint64_t* tag; uint64_t n, ncap;
    mi_ref_file* files; mi_ref_chunk* chunks; uint64_t n_chunks; int ran;
};
typedef struct mi_ref_ctx mi_ref_ctx;
typedef struct mi_ref_batch mi_ref_batch;

int mi_ref_abi_version(void) { return MI_ABI_VERSION; }

int mi_ref_config_default(mi_config* cfg) {
    if (!cfg) return MI_ERR_INVALID;
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->gear_seed = 0x4D414B49ull;
    cfg->mask_bits = 13;
    cfg->min_size = 2048;
    cfg->max_size = 65536;
    return MI_OK;
}

int mi_ref_ctx_create(const mi_config* cfg, mi_ref_ctx** out) {
    if (!cfg || !out || cfg->struct_size != sizeof(mi_config)) return MI_ERR_INVALID;
    if (cfg->mask_bits > 32 || cfg->min_size < 64 || cfg->max_size < cfg->min_size || cfg->max_size > (1u << 30))
        return MI_ERR_INVALID;
    mi_ref_ctx* c = (mi_ref_ctx*)calloc(1, sizeof *c);
    if (!c) return MI_ERR_NOMEM;
    c->cfg = *cfg;
    *out = c;
    return MI_OK;
}
int mi_ref_ctx_destroy(mi_ref_ctx* c) { free(c); return MI_OK; }
const char* mi_ref_last_error(mi_ref_ctx* c) { return c ? c->err : ""; }

int mi_ref_batch_begin(mi_ref_ctx* c, uint64_t n_files_hint, uint64_t bytes_hint, mi_ref_batch** out) {
    (void)n_files_hint; (void)bytes_hint;
    if (!c || !out) return MI_ERR_INVALID;
    mi_ref_batch* b = (mi_ref_batch*)calloc(1, sizeof *b);
    if (!b) return MI_ERR_NOMEM;
    b->ctx = c;
    *out = b;
    return MI_OK;
}

static int reserve(mi_ref_batch* b, uint64_t len) {
    if (b->n == b->ncap) {
        uint64_t nc = b->ncap ? 2 * b->ncap : 64;
        b->off = (uint64_t*)realloc(b->off, nc * 8);
        b->size = (uint64_t*)realloc(b->size, nc * 8);
        b->tag = (uint64_t*)realloc(b->tag, nc * 8);
        if (!b->off || !b->size || !b->tag) return MI_ERR_NOMEM;
        b->ncap = nc;
    }
    if (b->used + len > b->cap) {
        uint64_t nc = (b->used + len) * 2 + 4096;
        uint8_t* p = (uint8_t*)realloc(b->data, nc);
        if (!p) return MI_ERR_NOMEM;
        b->data = p;
        b->cap = nc;
    }
    return MI_OK;
}

int mi_ref_batch_add_bytes(mi_ref_batch* b, const void* data, uint64_t len, uint64_t user_tag) {
    if (!b || (!data && len)) return MI_ERR_INVALID;
    if (b->ran) { snprintf(b->ctx->err, sizeof b->ctx->err, "batch already ran; begin a new batch"); return MI_ERR_STATE; }
    int rc = reserve(b, len);
    if (rc) return rc;
    if (len) memcpy(b->data + b->used, data, len);
    b->off[b->n] = b->used; b->size[b->n] = len; b->tag[b->n] = user_tag;
    b->used += len;
    b->n++;
    return MI_OK;
}

int mi_ref_batch_add_path(mi_ref_batch* b, const char* path, uint64_t size, uint64_t user_tag) {
    if (!b || !path) return MI_ERR_INVALID;
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { snprintf(b->ctx->err, sizeof b->ctx->err, "open %s: %s", path, strerror(errno)); return MI_ERR_IO; }
    int rc = reserve(b, size);
    uint64_t got = 0;
    while (!rc && got < size) {                      /* io.CopyN: exactly `size` bytes */
        ssize_t r = pread(fd, b->data + b->used + got, size - got, (off_t)got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) { snprintf(b->ctx->err, sizeof b->ctx->err, "read %s: file shorter than the size given", path); rc = MI_ERR_IO; }
        else got += (uint64_t)r;
    }
    close(fd);
    if (rc) return rc;
    b->off[b->n] = b->used; b->size[b->n] = size; b->tag[b->n] = user_tag;
    b->used += size;
    b->n++;
    return MI_OK;
}

int mi_ref_batch_run(mi_ref_batch* b) {
    if (!b) return MI_ERR_INVALID;
    if (b->ran) return MI_ERR_STATE;
    const mi_config* cfg = &b->ctx->cfg;
    mi_ref_cdc_params p = {cfg->gear_seed, cfg->mask_bits, cfg->min_size, cfg->max_size};
    uint64_t cap = 1;
    for (uint64_t f = 0; f < b->n; f++) cap += b->size[f] / cfg->min_size + 2;
    b->files = (mi_ref_file*)calloc(b->n ? b->n : 1, sizeof(mi_ref_file));
    b->chunks = (mi_ref_chunk*)calloc(cap, sizeof(mi_ref_chunk));
    if (!b->files || !b->chunks) return MI_ERR_NOMEM;
    int flags = 0;
    if (cfg->flags & MI_FLAG_FILE_SHA256) flags |= MI_REF_FILE_SHA256;
    if (cfg->flags & MI_FLAG_FILE_CRC32) flags |= MI_REF_FILE_CRC32;
    if (cfg->flags & MI_FLAG_NO_DEDUP) flags |= MI_REF_NO_DEDUP;
    uint8_t dummy = 0;
    b->n_chunks = mi_ref_scan_batch(b->data ? b->data : &dummy, b->off, b->size, b->n, &p, 1, 1, flags, b->files,
                                    b->chunks, cap);
    if (b->n_chunks == (uint64_t)-1) return MI_ERR_INVALID;
    b->ran = 1;
    return MI_OK;
}

int mi_ref_batch_counts(mi_ref_batch* b, uint64_t* n_files, uint64_t* n_chunks, uint64_t* n_bytes) {
    if (!b) return MI_ERR_INVALID;
    if (n_files) *n_files = b->n;
    if (n_chunks) *n_chunks = b->ran ? b->n_chunks : 0;
    if (n_bytes) *n_bytes = b->used;
    return MI_OK;
}

int mi_ref_batch_files(mi_ref_batch* b, mi_file_result* out, uint64_t cap) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    if (!b->ran) { snprintf(b->ctx->err, sizeof b->ctx->err, "results requested before mi_batch_run"); return MI_ERR_STATE; }
    if (cap < b->n) return MI_ERR_CAPACITY;
    for (uint64_t f = 0; f < b->n; f++) {
        memset(&out[f], 0, sizeof out[f]);
        out[f].user_tag = b->tag[f];
        out[f].size = b->size[f];
        out[f].first_chunk = b->files[f].first_chunk;
        out[f].n_chunks = (uint32_t)b->files[f].n_chunks;
        out[f].crc32 = b->files[f].crc32;
        memcpy(out[f].chunk_root, b->files[f].chunk_root, 32);
        if (b->ctx->cfg.flags & MI_FLAG_FILE_SHA256) memcpy(out[f].file_sha256, b->files[f].file_sha256, 32);
    }
    return MI_OK;
}

int mi_ref_batch_chunks(mi_ref_batch* b, mi_chunk_result* out, uint64_t cap) {
    if (!b || (!out && cap)) return MI_ERR_INVALID;
    if (!b->ran) return MI_ERR_STATE;
    if (cap < b->n_chunks) return MI_ERR_CAPACITY;
    for (uint64_t i = 0; i < b->n_chunks; i++) {
        memset(&out[i], 0, sizeof out[i]);
        out[i].file_index = b->chunks[i].file_index;
        out[i].offset = b->chunks[i].offset;
        out[i].length = b->chunks[i].length;
        out[i].dup_of = b->chunks[i].dup_of;
        memcpy(out[i].sha256, b->chunks[i].sha256, 32);
    }
    return MI_OK;
}

int mi_ref_batch_free(mi_ref_batch* b) {
    if (!b) return MI_ERR_INVALID;
    free(b->data); free(b->off); free(b->size); free(b->tag); free(b->files); free(b->chunks);
    free(b);
    return MI_OK;
}

/* mi_dedup_mark's twin: the digest and dup_of arrays are HOST memory here */
int mi_ref_dedup_mark(mi_ref_ctx* c, const void* digests, uint64_t n, void* dup_of, uint64_t* n_unique) {
    if (!c || (n && (!digests || !dup_of))) return MI_ERR_INVALID;
    uint64_t u = mi_ref_dedup((const uint8_t*)digests, n, (int64_t*)dup_of);
    if (n_unique) *n_unique = u;
    return MI_OK;
}

int mi_ref_sha256_many(mi_ref_ctx* c, const void* data, const uint64_t* offsets, const uint64_t* lens, uint64_t n,
                       uint8_t* out) {
    if (!c || (n && (!offsets || !lens || !out))) return MI_ERR_INVALID;
    for (uint64_t i = 0; i < n; i++) mi_ref_sha256((const uint8_t*)data + offsets[i], (size_t)lens[i], out + 32 * i, 1);
    return MI_OK;
}
