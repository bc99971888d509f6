/* layer_driver.c -- commitLayer from plain C: what the cgo shim's AddLayerByScan would do with the host entry points.
 *   layer_driver <before-dir> <after-dir> <out.tar.gz>
 * walks both trees (scan rules), diffs them (mi_snapshot_diff), writes the changed entries, their carried ancestors and
 * one whiteout per deleted subtree in commit order through the layer writer (tar framing + both stream digests + gzip,
 * step.tarAndGzipDiffs / commitLayer, lib/builder/step/common.go:35-111) and prints the DigestPair:
 *   T <tar sha256 hex> <tar bytes>\n G <gzip sha256 hex> <gzip bytes>\n N <entries>\n                                */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "makisu_mi.h"

static int walk(const char* dir, mi_tree** t, mi_tree_entry** ents, uint64_t* n) {
    int rc = mi_tree_walk(dir, NULL, NULL, 0, MI_TREE_SCAN, t, n);
    if (rc != MI_OK) return rc;
    *ents = calloc(*n ? *n : 1, sizeof **ents);
    return mi_tree_entries(*t, *ents, *n);
}

static void hex(const uint8_t* d, char* out) {
    for (int i = 0; i < 32; i++) sprintf(out + 2 * i, "%02x", d[i]);
}

typedef struct { const char* key; const mi_tree_entry* e; int whiteout; } item;
static int by_key(const void* a, const void* b) { return strcmp(((const item*)a)->key, ((const item*)b)->key); }

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    mi_tree *tb = NULL, *ta = NULL;
    mi_tree_entry *eb = NULL, *ea = NULL;
    uint64_t nb = 0, na = 0;
    if (walk(argv[1], &tb, &eb, &nb) != MI_OK || walk(argv[2], &ta, &ea, &na) != MI_OK) return 1;
    mi_snapshot_side before = {eb, nb, NULL, 0}, after = {ea, na, NULL, 0};
    uint8_t* flags = calloc(na ? na : 1, 1);
    uint8_t* wh = calloc(nb ? nb : 1, 1);
    if (mi_snapshot_diff(&before, &after, 0, flags, wh) != MI_OK) return 1;
    /* the layer's files map, sorted by key = the absolute path (rangeFiles, mem_layer.go:232-244) */
    item* items = calloc(na + nb + 1, sizeof *items);
    size_t n = 0;
    for (uint64_t i = 0; i < na; i++)
        if (flags[i] != MI_DIFF_SAME && ea[i].relpath[0] && strcmp(ea[i].relpath, ".") != 0)
            items[n++] = (item){ea[i].relpath, &ea[i], 0};
    for (uint64_t j = 0; j < nb; j++)
        if (wh[j]) items[n++] = (item){eb[j].relpath, &eb[j], 1};
    qsort(items, n, sizeof *items, by_key);
    int fd = open(argv[3], O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return 1;
    mi_layer_config cfg;
    mi_layer_config_default(&cfg);
    cfg.out_fd = fd;
    cfg.gzip_level = MI_GZIP_DEFAULT;
    mi_layer* layer = NULL;
    if (mi_layer_begin(&cfg, &layer) != MI_OK) return 1;
    char path[8192];
    for (size_t k = 0; k < n; k++) {
        int rc;
        if (items[k].whiteout) {
            snprintf(path, sizeof path, "/%s", items[k].e->relpath);
            rc = mi_layer_add_whiteout(layer, path);
        } else {
            snprintf(path, sizeof path, "%s/%s", argv[2], items[k].e->relpath);
            rc = mi_layer_add(layer, items[k].e, items[k].e->kind == 1 ? path : NULL);
        }
        if (rc != MI_OK) { fprintf(stderr, "layer: %s\n", mi_layer_error(layer)); return 1; }
    }
    mi_layer_result res;
    if (mi_layer_finish(layer, &res) != MI_OK) { fprintf(stderr, "finish: %s\n", mi_layer_error(layer)); return 1; }
    close(fd);
    char t[65], g[65];
    hex(res.tar_sha256, t);
    hex(res.gzip_sha256, g);
    printf("T %s %llu\nG %s %llu\nN %llu\n", t, (unsigned long long)res.tar_bytes, g, (unsigned long long)res.gzip_bytes,
           (unsigned long long)res.n_entries);
    mi_layer_free(layer);
    free(items); free(flags); free(wh); free(eb); free(ea);
    mi_tree_free(tb);
    mi_tree_free(ta);
    return 0;
}
