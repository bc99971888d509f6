/* memfs_driver.c -- a build stage's snapshot side from plain C, on the MemFS handle:
 *   memfs_driver <root> <shell command> <out1.tar> <out2.tar>
 * NewMemFS(root); AddLayerByScan -> out1.tar; the command runs (a RUN step changing the root); AddLayerByScan again ->
 * out2.tar.  Each layer goes through the layer writer as mi_copy_layer_entries hands it over (commit order, source
 * paths, whiteouts by name).  Prints per layer:  L <k> <entries> <tar sha256 hex>\n  and the entry names  E <k> <name>\n */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "makisu_mi.h"

static int scan_and_commit(mi_memfs* fs, const char* root, const char* out_path, int k) {
    mi_tree* t = NULL;
    uint64_t n = 0;
    if (mi_tree_walk(root, root, NULL, 0, MI_TREE_SCAN, &t, &n) != MI_OK) return 1;
    mi_tree_entry* walked = calloc(n ? n : 1, sizeof *walked);
    if (mi_tree_entries(t, walked, n) != MI_OK) return 1;
    mi_copy_layer* cl = NULL;
    uint64_t ne = 0;
    if (mi_memfs_add_layer_by_scan(fs, walked, n, NULL, 0, &cl, &ne) != MI_OK) {
        fprintf(stderr, "scan: %s\n", mi_memfs_error(fs));
        return 1;
    }
    mi_tree_entry* ents = calloc(ne ? ne : 1, sizeof *ents);
    const char** srcs = calloc(ne ? ne : 1, sizeof *srcs);
    if (mi_copy_layer_entries(cl, ents, srcs, ne) != MI_OK) return 1;
    FILE* f = fopen(out_path, "wb");
    if (!f) return 1;
    mi_layer_config cfg;
    mi_layer_config_default(&cfg);
    cfg.out_fd = fileno(f);
    cfg.gzip_level = MI_GZIP_OFF;
    mi_layer* layer = NULL;
    if (mi_layer_begin(&cfg, &layer) != MI_OK) return 1;
    for (uint64_t i = 0; i < ne; i++) {
        if (mi_layer_add(layer, &ents[i], ents[i].kind == 1 && srcs[i][0] ? srcs[i] : NULL) != MI_OK) {
            fprintf(stderr, "layer: %s\n", mi_layer_error(layer));
            return 1;
        }
        printf("E %d %s\n", k, ents[i].relpath);
    }
    mi_layer_result res;
    if (mi_layer_finish(layer, &res) != MI_OK) return 1;
    char hex[65];
    for (int i = 0; i < 32; i++) sprintf(hex + 2 * i, "%02x", res.tar_sha256[i]);
    printf("L %d %llu %s\n", k, (unsigned long long)ne, hex);
    mi_layer_free(layer);
    fclose(f);
    mi_copy_layer_free(cl);
    free(ents); free(srcs); free(walked);
    mi_tree_free(t);
    return 0;
}

int main(int argc, char** argv) {
    if (argc != 5) return 2;
    mi_memfs* fs = NULL;
    if (mi_memfs_create(argv[1], NULL, 0, 0, &fs) != MI_OK) return 1;
    if (scan_and_commit(fs, argv[1], argv[3], 1)) return 1;
    if (system(argv[2]) != 0) return 1;
    if (scan_and_commit(fs, argv[1], argv[4], 2)) return 1;
    uint64_t n = 0;
    if (mi_memfs_entries(fs, NULL, NULL, 0, &n) != MI_ERR_CAPACITY && n != 0) return 1;
    printf("T %llu\n", (unsigned long long)n);
    mi_memfs_free(fs);
    return 0;
}
