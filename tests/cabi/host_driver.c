/* Plain-C consumer of the HOST-side entry points of include/makisu_mi.h (no GPU needed):
 *   host_driver <before-dir> <after-dir>
 * walks both directories the way the snapshot scan does, prints the new walk in commit order and
 * the layer diff between the two.  Built and run by tests/test_host_walk.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "makisu_mi.h"

static int walk(const char* dir, mi_tree** t, mi_tree_entry** ents, uint64_t* n) {
    int rc = mi_tree_walk(dir, NULL, NULL, 0, MI_TREE_SCAN, t, n);
    if (rc != MI_OK) return rc;
    *ents = calloc(*n ? *n : 1, sizeof **ents);
    return mi_tree_entries(*t, *ents, *n);
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    mi_tree *tb = NULL, *ta = NULL;
    mi_tree_entry *eb = NULL, *ea = NULL;
    uint64_t nb = 0, na = 0;
    if (walk(argv[1], &tb, &eb, &nb) != MI_OK || walk(argv[2], &ta, &ea, &na) != MI_OK) {
        fprintf(stderr, "walk failed\n");
        return 1;
    }
    uint64_t* order = calloc(na ? na : 1, sizeof *order);
    if (mi_entries_commit_order(ea, na, order) != MI_OK) return 1;
    for (uint64_t k = 0; k < na; k++) printf("O %s\n", ea[order[k]].relpath);
    mi_snapshot_side before = {eb, nb, NULL, 0}, after = {ea, na, NULL, 0};
    uint8_t* flags = calloc(na ? na : 1, 1);
    uint8_t* wh = calloc(nb ? nb : 1, 1);
    if (mi_snapshot_diff(&before, &after, 1 /* ignore_time */, flags, wh) != MI_OK) return 1;
    for (uint64_t i = 0; i < na; i++)
        if (flags[i] != MI_DIFF_SAME) printf("%c %s\n", flags[i] == MI_DIFF_CHANGED ? 'C' : 'A', ea[i].relpath);
    for (uint64_t j = 0; j < nb; j++)
        if (wh[j]) printf("W %s\n", eb[j].relpath);
    int similar = -1;
    if (na && nb && mi_entry_similar(&eb[0], &ea[0], 1, NULL, NULL, &similar) != MI_OK) return 1;
    printf("S %d\n", similar);
    free(order); free(flags); free(wh); free(eb); free(ea);
    mi_tree_free(tb);
    mi_tree_free(ta);
    return 0;
}
