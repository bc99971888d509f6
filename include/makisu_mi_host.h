/* makisu_mi_host.h -- OPTIONAL host helpers of libmakisu_mi.so.  NOT part of the drop-in contract.
 *
 * include/makisu_mi.h is the boundary SURVEY.md 8(b) lists plus what rows a1-a14 need.  The functions below restate
 * pieces of the reference that SURVEY.md section 2 marks OUT OF SCOPE for this engine (lib/fileio: "disk -> disk copy,
 * not hashed"; lib/utils; the Go standard library's filepath.Match / Glob; MemFS's untar and checkpoint halves).  They
 * were written in round 3 so that the snapshot side could be driven end to end from C in tests; a Go host has all of
 * them already (they ARE its own code) and would not bind them.  Frozen: no new entry point is added here.
 * Same conventions as makisu_mi.h (int return codes, err / err_cap buffers, host logic, no device).               */
#ifndef MAKISU_MI_HOST_H
#define MAKISU_MI_HOST_H

#include "makisu_mi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The rest of the caller's side of a COPY/ADD step.
 * mi_resolve_chown: NewCopyOperation's --chown handling (lib/snapshot/copy_op.go:51-60 -> utils.ResolveChown,
 *   lib/utils/utils.go:186-228): "" = 0:0; "<user>[:<group>]", each a decimal number or a name from the user / group
 *   database; no group = the uid; chown together with preserve_owner (--archive) is an error.  MI_ERR_INVALID + the
 *   reference's message in err.
 * mi_path_match: path/filepath.Match of the Go toolchain the reference builds with ('*' and '?' never match '/',
 *   "[^a-c]" classes on runes, '\\' escapes; MI_ERR_INVALID = ErrBadPattern).
 * mi_context_sources: addCopyStep.resolveFromPaths (lib/builder/step/add_copy_step.go:171-185): every source is joined
 *   to the context root and expanded with filepath.Glob (matches of one pattern in sorted order); a pattern that
 *   matches nothing, or is malformed, stands for itself.  out = the resolved paths, NUL-terminated, back to back (*n_out
 *   paths, *bytes_out bytes; MI_ERR_CAPACITY if cap is smaller: call with cap 0 to size).  Each goes to
 *   mi_batch_add_tree(..., rel_base = context dir, MI_TREE_CONTEXT) in this order for the cache ID, and -- trimmed of
 *   the root -- into mi_copy_op.srcs.  Host logic.                                                                */
MI_BLOCK int  mi_resolve_chown(const char* chown, int preserve_owner, int64_t* uid, int64_t* gid, char* err, uint64_t err_cap);
MI_BLOCK int  mi_path_match(const char* pattern, const char* name, int* matched);
MI_BLOCK int  mi_context_sources(const char* context_root, const char* const* from_paths, uint64_t n_paths, char* out,
                        uint64_t cap, uint64_t* n_out, uint64_t* bytes_out);
/* CopyOperation.Execute (lib/snapshot/copy_op.go:83-147) over fileio.Copier (lib/fileio/copy.go): the on-disk copy of the
 * step, for builds that modify the file system.  op as for mi_memfs_add_layer_by_copy_ops (dst resolved; "dir/" = copy INTO it).
 * flags: MI_COPY_CHOWN --chown was given (owner = op->uid/gid for everything copied and for a destination directory
 * that has to be created); MI_COPY_INTERNAL the sources are a previous stage's (--from: no blacklist, owners kept);
 * MI_COPY_PRESERVE_OWNER --archive (a created destination directory gets the source's owner).  Without flags: from the
 * context, everything owned by 0:0.  Missing ancestors of the destination are created 0755 root:root; permission bits
 * are kept, mtimes are not; a symlink is copied as a link; special files are skipped; a source directory that contains
 * the destination does not recurse into it.  MI_ERR_IO + the reference's message.  Host logic.                        */
#define MI_COPY_CHOWN          0x1u
#define MI_COPY_INTERNAL       0x2u
#define MI_COPY_PRESERVE_OWNER 0x4u
MI_BLOCK int  mi_copy_op_execute(const mi_copy_op* op, uint32_t flags, const char* const* blacklist, uint64_t n_blacklist,
                        char* err, uint64_t err_cap);

/* UpdateFromTarReader with untar = true (the FROM step / a cached layer applied with --modifyfs): the entries of a PLAIN
 * tar -- mi_tar_entries(tar) with their data offsets; a gzip blob goes through mi_tar_inflate first -- are written below
 * the root as MemFS.untarOneItem does (lib/snapshot/mem_fs.go:571-718): a ".wh.<x>" entry removes <x>; what is already
 * on disk with a similar header stays; a directory on a directory is updated in place (tario.ApplyHeader: chown, chmod,
 * mtime); anything else is removed and created again; an absolute symlink target is re-rooted; hard links come last; the
 * mtimes of the parent directories are put back at the end -- and every header is merged into the tree as above.  After
 * it a scan of the root finds nothing to add.  Needs the privileges the reference needs (chown).  MI_ERR_IO + the
 * reference's message ("untar one item <path>: ...") on failure.                                                  */
MI_BLOCK int  mi_memfs_untar(mi_memfs* fs, const char* tar_path, const mi_tree_entry* layer, const uint64_t* data_offsets,
                    uint64_t n_layer, uint64_t* n_merged);

/* MemFS.Checkpoint (mem_fs.go:132-185): what a later stage will COPY --from is copied aside, to new_root + the path it has
 * below the root (patterns expanded like COPY sources; relative sources are below the root; the blacklist of the handle
 * applies; a created target directory gets the source's owner, everything copied keeps its own).                    */
MI_BLOCK int  mi_memfs_checkpoint(mi_memfs* fs, const char* new_root, const char* const* sources, uint64_t n_sources);


#ifdef __cplusplus
}
#endif

#endif /* MAKISU_MI_HOST_H */
