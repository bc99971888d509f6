// mi_local.h -- the library's own cross-file helpers: C linkage (the translation units name them without sharing C++
// headers), HIDDEN visibility -- they are not part of the ABI and `nm -D libmakisu_mi.so` does not list them
// (tests/test_abi.py: the library exports what the two public headers declare and nothing else).
#pragma once
#include "../../include/makisu_mi.h"

#define MI_LOCAL __attribute__((visibility("hidden")))

extern "C" {
// mi_api.hip
MI_LOCAL void        mi_set_error(mi_batch* b, const char* msg);     // b NULL: the message mi_last_error(NULL) returns
MI_LOCAL void**      mi_batch_tree_slot(mi_batch* b);                // the batch's walk record (mi_tree.hip owns its type)
MI_LOCAL int         mi_batch_file_size(mi_batch* b, uint64_t file_index, uint64_t* size);
MI_LOCAL const char* mi_last_error_of_batch(mi_batch* b);
// what the batch's arena holds now (bytes allocated): the room a window of an oversize tree can count on
MI_LOCAL int         mi_batch_arena_room(mi_batch* b, uint64_t* bytes);
// device memory behind the arena now, in how many pieces (1: one allocation), and how often its base address has changed
MI_LOCAL int         mi_batch_arena_info(mi_batch* b, uint64_t* bytes, uint64_t* pieces, uint64_t* moves);
// mi_batch_read_file for the pipelined commit: the batch is still being staged / scanned on another thread; the call waits
// until the bytes it is asked for have landed in HBM
MI_LOCAL int         mi_batch_read_file_landed(mi_batch* b, uint64_t file_index, uint64_t offset, void* dst, uint64_t len);
// MI_LAYER_TIMING: what the window behind mi_batch_read_file* did so far (seconds waiting for bytes to land, seconds in its copies)
MI_LOCAL void        mi_batch_read_stats(mi_batch* b, double* wait_s, double* fetch_s, uint64_t* fetches, uint64_t* bytes);
// mi_layer.hip: from now on mi_layer_add_batch_file reads through mi_batch_read_file_landed
MI_LOCAL void        mi_layer_set_pipelined(mi_layer* layer, int on);
// what the layer held against source sums so far: files, their bytes, chunks that were right only at the second fetch
MI_LOCAL void        mi_layer_verify_counts(mi_layer* layer, uint64_t* files, uint64_t* bytes, uint64_t* refetched);
// the walk's small files read in place: a block of host memory as one piece of the arena, and the table rows of files
// that lie in it; "host-fed bytes are on their way" (the reader threads set up behind the walk's first directories)
MI_LOCAL int  mi_batch_add_block(mi_batch* b, const void* src, uint64_t len, void (*release)(void*), void* release_arg,
                                 uint64_t* at_out);
MI_LOCAL int  mi_batch_add_placed(mi_batch* b, uint64_t n, const uint64_t* arena_off, const uint64_t* sizes,
                                  const uint64_t* tags, const uint64_t* sums);
// the end-to-end byte sums (mi_filesum.h): does the batch keep them (MI_FLAG_FILE_SUMS; a MemFS handle's batch always does --
// mi_batch_keep_sums, before its first file); the sums of chunk k (1 MiB of the file) of a row (*has = 0: none kept); forget what the
// read-back windows hold (the next read fetches again); which hop lost a chunk (a line for the error message)
MI_LOCAL int  mi_batch_keeps_sums(mi_batch* b);
MI_LOCAL void mi_batch_keep_sums(mi_batch* b, int on);
MI_LOCAL int  mi_batch_chunk_sum(mi_batch* b, uint64_t file_index, uint64_t chunk, uint64_t* sum_a, uint64_t* sum_b, int* has);
MI_LOCAL void mi_batch_drop_windows(mi_batch* b);
MI_LOCAL int  mi_batch_prepare_read(mi_batch* b);      // the read-back windows now, not at the tar writer's first read
MI_LOCAL int  mi_batch_explain_chunk(mi_batch* b, uint64_t file_index, uint64_t chunk, char* msg, uint64_t cap);
MI_LOCAL void mi_batch_expect_host_bytes(mi_batch* b);
// mi_batch_reserve for a walk whose enumeration runs ahead of what it hands over (and for mi_memfs_reserve_device): the arena
// is the piecewise kind (mi_arena.hip) -- what is coming is known roughly and keeps growing
MI_LOCAL int  mi_batch_reserve_ahead(mi_batch* b, uint64_t more_files, uint64_t more_bytes);
// one handle over n batches, one per ctx: what a walk hands over is spread by bytes, the commit sees one batch (mi_internal.h:
// members); the members and their loads (n = 0: not a group)
MI_LOCAL int  mi_batch_group_begin(mi_ctx* const* ctxs, uint32_t n, mi_batch** out);
MI_LOCAL uint64_t mi_batch_group_splits(mi_batch* b);      // files of the group that are split over its members as parts
MI_LOCAL int  mi_batch_group_members(mi_batch* b, mi_batch* const** members, const uint64_t** bytes, uint64_t* n);
// job-wide marking of a rank's own rows, enqueued on the ctx stream; the first-occurrence count stays in
// ctx->dd_nuniq (device).  For mi_comm.hip
MI_LOCAL int  mi_dedup_mark_range_enqueue(mi_ctx* c, const void* d_digests, uint64_t n_total, uint64_t own_first,
                                          uint64_t own_n, void* d_dup_of_own);
// mi_index.hip: mi_index_add_batch for digests in host memory (a batch of another GPU than the index's)
MI_LOCAL int  mi_index_add_digests(mi_index* index, const void* digests, uint64_t n, uint8_t* known_out, uint64_t* n_new, uint64_t* n_known);
MI_LOCAL int  mi_index_same_ctx(mi_index* index, mi_batch* b);
// mi_tree.hip
MI_LOCAL void mi_batch_tree_free(void* tree);
}
