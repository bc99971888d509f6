// mi_tar.hip -- host side, no device code: reading a layer tar WITHOUT extracting it.
//
// The reference fills its in-memory tree from base / cached layers by walking their tar headers
// (MemFS.UpdateFromTarReader, lib/snapshot/mem_fs.go:165-255: tar.Reader.Next per entry, hard
// links collected for a second pass, whiteouts applied by untarOneItem :571-650).  For the
// content scan the interesting part of a layer tar is that every regular file's bytes sit in ONE
// contiguous range of it: mi_tar_open lists the entries (as mi_tree_entry rows, the "before"
// side of mi_snapshot_diff) together with those ranges, so the files of a pulled layer can be
// handed to a batch straight out of the archive.
//
// Stored layers are gzip blobs (tario.NewGzipReader, lib/tario/gzip.go:50-53; pulled by
// lib/builder/build_node.go:133-148): mi_tar_open recognises the gzip magic and lists the entries
// through a streaming inflate (zlib); the data offsets it reports are offsets in the UNCOMPRESSED
// tar, which mi_tar_inflate writes out (returning its SHA-256 -- the layer's tar digest / diffID)
// so that mi_batch_add_path_range can read the members from it.
//
// Format: POSIX ustar / pax (typeflag 'x': path, linkpath, size, uid, gid, mtime) and the GNU
// extensions Go's archive/tar and docker write (typeflags 'L' / 'K' long name / link, base-256
// numeric fields, the old-GNU magic).  Checked against Python's tarfile on ustar, pax and GNU
// archives and on the reference's own layer fixture (tests/test_host_tar.py).
//
// WHAT IS AN ARCHIVE, what is its end and what is an error follow the reader the reference
// calls -- archive/tar of the Go 1.14 toolchain (Makefile:34; not under /root/reference: restated
// from the published source, reader.go next / readHeader / parsePAX, strconv.go parseNumeric /
// parsePAXRecord / parsePAXTime / mergePAX), not POSIX and not tarfile, where they differ:
//   * the archive ends at two zero blocks, at ONE zero block followed by the end of the data, or
//     at the end of the data on a block boundary (also inside a member's padding); a zero block
//     followed by a non-zero block and a last block of 1..511 bytes are errors ("read header: ...",
//     mem_fs.go:185);
//   * a numeric field is trimmed of spaces and NULs on both sides, cut at its first NUL and must
//     then be octal digits only; base-256 values beyond 63 bits are errors;
//   * a pax record is "<len> <key>=<value>\n" with len >= 5, the newline where len says, a key
//     that is not empty; a record with an EMPTY value keeps the header's own field; size, uid, gid
//     must be decimal int64 and the three times "[-]digits[.digits]" -- anything else fails the
//     archive; of several 'x' headers before one member the last one counts;
//   * a GLOBAL pax header ('g') is an entry of its own (kind 4, the header's name) and its records
//     touch NO later member (archive/tar hands it to the caller and merges nothing);
//   * a GNU long name / link wins over a pax path / linkpath (merged first), and an empty one
//     changes nothing;
//   * the size field of a link, symlink, device, directory or fifo header describes no data.
// NOT restated: sparse members (GNU 'S', pax GNU.sparse.*), which archive/tar expands to their logical size -- they are
// listed with the bytes the archive holds for them (docker and makisu write none); a global header's node in the tree
// (the reference keeps one of an "unsupported type" that fails its next scan; mi_memfs leaves kind 4 out).
#include "../../include/makisu_mi.h"

#include <errno.h>
#include <math.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "host_sha256.h"

#include <map>
#include <string>
#include <vector>

namespace mi_tarfile {

struct Item {
    std::string name, link;
    bool has_link = false;
    uint64_t size = 0, data_off = 0;
    int64_t mtime = 0;
    uint32_t mode = 0, uid = 0, gid = 0;
    uint8_t kind = 0;      // 0 dir, 1 regular, 2 symlink, 3 hard link, 4 other (device, fifo)
    int64_t file_index = -1;
};

struct Tar {
    std::vector<Item> items;
    std::string error;
};

static std::string field(const unsigned char* p, size_t n) {        // NUL-terminated or full width
    size_t k = 0;
    while (k < n && p[k]) ++k;
    return std::string((const char*)p, k);
}

// numeric field as archive/tar's parseNumeric reads it: GNU base-256 (top bit set: big-endian two's
// complement, at most 63 bits of magnitude), else octal text -- spaces and NULs trimmed on both sides, cut at
// the first NUL, then octal digits only; an empty field reads as 0
static bool number(const unsigned char* p, size_t n, int64_t* out) {
    if (n && (p[0] & 0x80)) {
        const unsigned char inv = (p[0] & 0x40) ? 0xff : 0x00;
        uint64_t x = 0;
        for (size_t i = 0; i < n; ++i) {
            unsigned char c = (unsigned char)(p[i] ^ inv);
            if (i == 0) c &= 0x7f;                           // the flag bit is not data
            if (x >> 56) return false;                       // beyond 64 bits
            x = (x << 8) | c;
        }
        if (x >> 63) return false;
        *out = inv ? (int64_t)~x : (int64_t)x;
        return true;
    }
    size_t a = 0, b = n;
    while (a < b && (p[a] == ' ' || p[a] == 0)) ++a;
    while (b > a && (p[b - 1] == ' ' || p[b - 1] == 0)) --b;
    for (size_t i = a; i < b; ++i) if (p[i] == 0) { b = i; break; }
    *out = 0;
    if (a == b) return true;                                 // nothing but padding
    uint64_t v = 0;
    for (size_t i = a; i < b; ++i) {
        if (p[i] < '0' || p[i] > '7' || (v >> 61)) return false;
        v = (v << 3) | (uint64_t)(p[i] - '0');
    }
    *out = (int64_t)v;
    return true;
}

static bool all_zero(const unsigned char* b) {
    for (int i = 0; i < 512; ++i) if (b[i]) return false;
    return true;
}

static bool checksum_ok(const unsigned char* b) {
    int64_t want = 0;
    if (!number(b + 148, 8, &want)) return false;
    int64_t u = 0, s = 0;
    for (int i = 0; i < 512; ++i) {
        const unsigned char c = (i >= 148 && i < 156) ? (unsigned char)' ' : b[i];
        u += c;
        s += (signed char)c;
    }
    return want == u || want == s;
}

// strconv.ParseInt(s, 10, 64): an optional sign, decimal digits and nothing else, within int64
static bool parse_int64(const std::string& s, int64_t* out) {
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; ++i; }
    if (i == s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (~0ull - d) / 10) return false;
        v = v * 10 + d;
    }
    if (v > (neg ? (1ull << 63) : (1ull << 63) - 1)) return false;
    *out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return true;
}

// parsePAXTime, down to what the reference keeps of it -- whole seconds (tario.IsSimilarHeader and WriteHeader
// Truncate to the second, lib/tario/compare.go:70-72, write.go:56: rounding DOWN): "[-]digits[.digits]", the
// fraction cut to nanoseconds; a fraction of a negative time moves it further from zero
static bool pax_seconds(const std::string& s, int64_t* out) {
    const size_t dot = s.find('.');
    const std::string ss = s.substr(0, dot), sn = dot == std::string::npos ? std::string() : s.substr(dot + 1);
    int64_t secs = 0;
    if (!parse_int64(ss, &secs)) return false;
    bool frac = false;
    for (size_t i = 0; i < sn.size(); ++i) {
        if (sn[i] < '0' || sn[i] > '9') return false;
        if (i < 9 && sn[i] != '0') frac = true;
    }
    if (frac && ss[0] == '-' && secs > INT64_MIN) --secs;
    *out = secs;
    return true;
}

// "len key=value\n" records (parsePAXRecord + validPAXRecord); a later record of a key replaces an earlier one
static bool parse_pax(const std::string& body, std::map<std::string, std::string>* kv) {
    size_t at = 0;
    while (at < body.size()) {
        const size_t sp = body.find(' ', at);
        if (sp == std::string::npos) return false;
        int64_t len = 0;
        if (!parse_int64(body.substr(at, sp - at), &len) || len < 5 || (uint64_t)len > body.size() - at) return false;
        const size_t end = at + (size_t)len;                  // one past the record's newline
        if (end < sp + 2 || body[end - 1] != '\n') return false;
        const std::string rec = body.substr(sp + 1, end - 1 - (sp + 1));
        const size_t eq = rec.find('=');
        if (eq == std::string::npos || eq == 0) return false;
        const std::string key = rec.substr(0, eq), val = rec.substr(eq + 1);
        const bool text = key == "path" || key == "linkpath" || key == "uname" || key == "gname";
        if ((text ? val : key).find('\0') != std::string::npos) return false;
        (*kv)[key] = val;
        at = end;
    }
    return true;
}

// Where the tar bytes come from: a plain file (random access) or a gzip stream (forward only --
// the parser only ever moves forward).
struct Source {
    int fd = -1;
    bool gz = false;
    uint64_t size = 0;                 // plain: file size; gzip: unknown until the stream ends
    // gzip state
    z_stream z;
    bool z_ready = false, z_end = false;
    std::vector<unsigned char> in, sink;   // compressed input window; where skipped bytes are inflated to
    uint64_t in_off = 0, pos = 0;      // compressed bytes consumed from fd, uncompressed position
    std::string error;

    ~Source() { if (z_ready) inflateEnd(&z); }

    bool open_fd(int f) {
        fd = f;
        struct stat st;
        if (fstat(fd, &st) != 0) { error = std::string("stat: ") + strerror(errno); return false; }
        size = (uint64_t)st.st_size;
        unsigned char magic[2] = {0, 0};
        if (size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            gz = true;
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, 15 + 16) != Z_OK) { error = "inflateInit2 failed"; return false; }
            z_ready = true;
            in.resize(1 << 20);
            sink.resize(1 << 16);      // per source: two threads may list two archives at once
        }
        return true;
    }
    // inflates up to n bytes into dst (dst may be NULL: discard); returns bytes produced
    size_t inflate_some(unsigned char* dst, size_t n) {
        size_t made = 0;
        while (made < n && !z_end) {
            if (z.avail_in == 0) {
                const ssize_t r = pread(fd, in.data(), in.size(), (off_t)in_off);
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) { error = "gzip stream ends early"; return made; }
                in_off += (uint64_t)r;
                z.next_in = in.data();
                z.avail_in = (uInt)r;
            }
            const size_t want = n - made;
            unsigned char* out = dst ? dst + made : sink.data();
            const size_t room = dst ? want : (want < sink.size() ? want : sink.size());
            z.next_out = out;
            z.avail_out = (uInt)(room > 0x40000000u ? 0x40000000u : room);
            const uInt before = z.avail_out;
            const int rc = inflate(&z, Z_NO_FLUSH);
            made += before - z.avail_out;
            if (rc == Z_STREAM_END) {
                // concatenated members (pgzip writes one, but RFC 1952 allows several)
                if (z.avail_in > 0 || in_off < size) { if (inflateReset(&z) != Z_OK) { error = "inflateReset failed"; return made; } }
                if (z.avail_in == 0 && in_off >= size) z_end = true;
            } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                error = "corrupt gzip stream";
                return made;
            }
        }
        pos += made;
        return made;
    }
    // exactly n bytes at uncompressed offset off; false at end of data (error empty) or on error
    bool read_at(uint64_t off, void* dst, size_t n) {
        if (!gz) {
            if (off + n > size) return false;
            size_t got = 0;
            while (got < n) {
                const ssize_t r = pread(fd, (char*)dst + got, n - got, (off_t)(off + got));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) { error = "short read in the archive"; return false; }
                got += (size_t)r;
            }
            return true;
        }
        if (off < pos) { error = "gzip source cannot seek backwards"; return false; }
        while (pos < off) {
            const uint64_t skip = off - pos;
            if (inflate_some(nullptr, (size_t)(skip > (1u << 30) ? (1u << 30) : skip)) == 0) return false;
        }
        return inflate_some((unsigned char*)dst, n) == n;
    }
    // up to n bytes at uncompressed offset off: fewer (0 included) at the end of the data; error set on failure
    size_t read_upto(uint64_t off, void* dst, size_t n) {
        if (!gz) {
            if (off >= size) return 0;
            const size_t m = (size_t)(size - off < n ? size - off : n);
            return read_at(off, dst, m) ? m : 0;
        }
        if (off < pos) { error = "gzip source cannot seek backwards"; return 0; }
        while (pos < off) {
            const uint64_t skip = off - pos;
            if (inflate_some(nullptr, (size_t)(skip > (1u << 30) ? (1u << 30) : skip)) == 0) return 0;   // ends before `off`
        }
        return inflate_some((unsigned char*)dst, n);
    }
    // can [off, off+n) exist?  (plain files: bounds check; gzip: only known by reading on)
    bool has_range(uint64_t off, uint64_t n) const { return gz || off + n <= size; }
};

static int parse(Source& src, Tar* t) {
    uint64_t off = 0;
    std::map<std::string, std::string> next_pax;               // the records of the last 'x' header before this member
    std::string gnu_name, gnu_link;                            // 'L' / 'K' before this member ("" = none)
    int64_t n_regular = 0;
    unsigned char blk[512];
    std::string open_member;                                   // gzip source: the entry whose data must reach `off`
    uint64_t open_member_end = 0;                              // ... its last byte + 1 (padding not counted)
    for (;;) {
        const size_t got = src.read_upto(off, blk, 512);
        if (got < 512) {
            if (!src.error.empty()) { t->error = src.error; return MI_ERR_IO; }
            // a gzip stream is only as long as it inflates to: if it ended inside the previous
            // entry's data, say what the plain-tar path says for the same archive
            if (src.gz && !open_member.empty() && src.pos < open_member_end) {
                t->error = "entry " + open_member + " runs past the end of the archive";
                return MI_ERR_INVALID;
            }
            if (got) { t->error = "unexpected end of the archive inside the header at offset " + std::to_string(off); return MI_ERR_INVALID; }
            break;                                             // end of data without the zero blocks (io.EOF from readHeader)
        }
        open_member.clear();
        if (all_zero(blk)) {                                   // end-of-archive marker: a second zero block, or nothing at all, after it
            const size_t more = src.read_upto(off + 512, blk, 512);
            if (!src.error.empty()) { t->error = src.error; return MI_ERR_IO; }
            if (more == 0 || (more == 512 && all_zero(blk))) break;
            t->error = more < 512 ? "unexpected end of the archive after a zero block at offset " + std::to_string(off)
                                  : "a zero block is followed by a header at offset " + std::to_string(off + 512);
            return MI_ERR_INVALID;
        }
        if (!checksum_ok(blk)) { t->error = "bad tar header checksum at offset " + std::to_string(off); return MI_ERR_INVALID; }
        const bool ustar = memcmp(blk + 257, "ustar\0", 6) == 0;                   // POSIX ustar / pax, or star with its trailer
        const bool star = ustar && memcmp(blk + 508, "tar\0", 4) == 0;
        const bool gnu = memcmp(blk + 257, "ustar  \0", 8) == 0;                   // old GNU: magic and version in one
        int64_t size = 0, mode = 0, uid = 0, gid = 0, mtime = 0, dev = 0;
        if (!number(blk + 124, 12, &size) || !number(blk + 100, 8, &mode) || !number(blk + 108, 8, &uid) ||
            !number(blk + 116, 8, &gid) || !number(blk + 136, 12, &mtime) ||
            ((ustar || gnu) && (!number(blk + 329, 8, &dev) || !number(blk + 337, 8, &dev))) ||
            (star && (!number(blk + 476, 12, &dev) || !number(blk + 488, 12, &dev)))) {
            t->error = "bad numeric field in the tar header at offset " + std::to_string(off);
            return MI_ERR_INVALID;
        }
        const unsigned char type = blk[156];
        const bool header_only = type == '1' || type == '2' || type == '3' || type == '4' || type == '5' || type == '6';
        if (size < 0 && !header_only) {
            t->error = "negative size in the tar header at offset " + std::to_string(off);
            return MI_ERR_INVALID;
        }
        const uint64_t data = off + 512;
        if (type == 'x' || type == 'g' || type == 'L' || type == 'K') {           // metadata: for the next member, or ('g') for nobody
            const uint64_t padded = ((uint64_t)size + 511) & ~511ull;
            if ((uint64_t)size > (64u << 20) || !src.has_range(data, (uint64_t)size)) {
                t->error = "truncated extended header at offset " + std::to_string(off);
                return MI_ERR_INVALID;
            }
            std::string body((size_t)size, '\0');
            if (size && !src.read_at(data, &body[0], (size_t)size)) {
                t->error = src.error.empty() ? "truncated extended header at offset " + std::to_string(off) : src.error;
                return MI_ERR_INVALID;
            }
            off = data + padded;
            if (type == 'L') { gnu_name = body.substr(0, body.find('\0')); continue; }
            if (type == 'K') { gnu_link = body.substr(0, body.find('\0')); continue; }
            std::map<std::string, std::string> recs;
            if (!parse_pax(body, &recs)) {
                t->error = "malformed pax record at offset " + std::to_string(data - 512);
                return MI_ERR_INVALID;
            }
            if (type == 'x') { next_pax.swap(recs); continue; }
            // 'g': handed to the caller as a member of its own -- only its name (a non-empty path record replaces it)
            // and its typeflag survive; no later member sees its records
            // (and what an 'x', 'L' or 'K' before it said is dropped with it: they live for one call of Next)
            Item g;
            g.name = field(blk, 100);
            if (ustar) { const std::string prefix = field(blk + 345, star ? 131 : 155); if (!prefix.empty()) g.name = prefix + "/" + g.name; }
            auto gp = recs.find("path");
            if (gp != recs.end() && !gp->second.empty()) g.name = gp->second;
            g.kind = 4;
            t->items.push_back(g);
            next_pax.clear();
            gnu_name.clear();
            gnu_link.clear();
            continue;
        }
        Item it;
        it.name = field(blk, 100);
        if (ustar) {                                            // (not old GNU): the prefix field, 131 bytes wide in star's layout
            const std::string prefix = field(blk + 345, star ? 131 : 155);
            if (!prefix.empty()) it.name = prefix + "/" + it.name;
        }
        it.link = field(blk + 157, 100);
        // mergePAX: a record with an empty value keeps the header's own field; a number or a time that does not
        // parse fails the archive
        bool pax_ok = true;
        for (const auto& kv : next_pax) {
            const std::string& k = kv.first;
            const std::string& v = kv.second;
            int64_t ignored = 0;
            if (v.empty()) continue;
            if (k == "path") it.name = v;
            else if (k == "linkpath") it.link = v;
            else if (k == "size") pax_ok &= parse_int64(v, &size);
            else if (k == "uid") pax_ok &= parse_int64(v, &uid);
            else if (k == "gid") pax_ok &= parse_int64(v, &gid);
            else if (k == "mtime") pax_ok &= pax_seconds(v, &mtime);
            else if (k == "atime" || k == "ctime") pax_ok &= pax_seconds(v, &ignored);
        }
        next_pax.clear();
        if (!gnu_name.empty()) it.name = gnu_name;              // after the pax records: a GNU long name wins
        if (!gnu_link.empty()) it.link = gnu_link;
        gnu_name.clear();
        gnu_link.clear();
        // "Legacy archives use trailing slash for directories": decided on the final name, and from here on the
        // member is header-only like any directory -- its size field describes no data
        const bool legacy_dir = type == 0 && !it.name.empty() && it.name.back() == '/';
        if (!pax_ok || (size < 0 && !header_only && !legacy_dir)) {
            t->error = "bad pax value for the member at offset " + std::to_string(off);
            return MI_ERR_INVALID;
        }
        uint32_t type_bits = 0;
        switch (type) {
            case '0': case 0: case '7': it.kind = 1; type_bits = S_IFREG; break;
            case '5': it.kind = 0; type_bits = S_IFDIR; break;
            case '2': it.kind = 2; type_bits = S_IFLNK; break;
            case '1': it.kind = 3; type_bits = S_IFREG; break;
            case '3': it.kind = 4; type_bits = S_IFCHR; break;
            case '4': it.kind = 4; type_bits = S_IFBLK; break;
            case '6': it.kind = 4; type_bits = S_IFIFO; break;
            default:  it.kind = 4; break;
        }
        if (legacy_dir) { it.kind = 0; type_bits = S_IFDIR; }
        it.has_link = it.kind == 2 || it.kind == 3;
        it.mode = ((uint32_t)mode & 07777u) | type_bits;
        it.uid = (uint32_t)uid;
        it.gid = (uint32_t)gid;
        it.mtime = mtime;
        it.size = it.kind == 1 ? (uint64_t)size : 0;
        it.data_off = it.kind == 1 ? data : 0;
        const uint64_t skip = header_only || legacy_dir ? 0 : (((uint64_t)size + 511) & ~511ull);
        if (it.kind == 1) {
            if (!src.has_range(data, (uint64_t)size)) { t->error = "entry " + it.name + " runs past the end of the archive"; return MI_ERR_INVALID; }
            it.file_index = n_regular++;
            if (src.gz && size > 0) { open_member = it.name; open_member_end = data + (uint64_t)size; }
        }
        else if (src.gz && skip > 0) { open_member = it.name; open_member_end = data + (uint64_t)size; }   // any other member with a data area
        else if (!src.gz && skip > 0 && !src.has_range(data, (uint64_t)size)) {
            t->error = "entry " + it.name + " runs past the end of the archive";
            return MI_ERR_INVALID;
        }
        t->items.push_back(std::move(it));
        off = data + skip;
    }
    return MI_OK;
}

// header name -> the relpath form the walks use: no leading "/" or "./", no trailing "/", "." for
// the root (pathutils.RelPath + AbsPath semantics of untarOneItem's filepath.Join(root, hdr.Name))
static std::string rel_name(const std::string& n) {
    {   // what archives hold almost always: clean elements, at most slashes at the ends -- no element list for those
        size_t a = 0, b = n.size();
        while (a < b && n[a] == '/') ++a;
        while (b > a && n[b - 1] == '/') --b;
        bool clean = b > a;
        for (size_t i = a; clean && i < b; ++i) {
            if (n[i] == '/' && n[i + 1] == '/') clean = false;                   // (i + 1 < b: b - 1 is not a slash)
            if (n[i] == '.' && (i == a || n[i - 1] == '/')) {
                const size_t k = (i + 1 < b && n[i + 1] == '.') ? i + 2 : i + 1;
                if (k >= b || n[k] == '/') clean = false;                        // a "." or ".." element
            }
        }
        if (clean) return n.substr(a, b - a);
    }
    std::vector<std::string> parts;
    size_t i = 0;
    while (i < n.size()) {
        while (i < n.size() && n[i] == '/') ++i;
        size_t j = i;
        while (j < n.size() && n[j] != '/') ++j;
        if (j > i) {
            const std::string el = n.substr(i, j - i);
            if (el == "..") { if (!parts.empty()) parts.pop_back(); }
            else if (el != ".") parts.push_back(el);
        }
        i = j;
    }
    std::string out;
    for (size_t k = 0; k < parts.size(); ++k) out += (k ? "/" : "") + parts[k];
    return out.empty() ? "." : out;
}

}  // namespace mi_tarfile

using mi_tarfile::Tar;

struct mi_tar {
    Tar t;
    std::vector<std::string> rel;
};

extern "C" {

static void put_err(char* err, uint64_t cap, const std::string& msg) {
    if (err && cap) snprintf(err, (size_t)cap, "%s", msg.c_str());
}

int mi_tar_open_ex(const char* path, mi_tar** out, uint64_t* n_entries, int* is_gzip, char* err, uint64_t err_cap) {
    if (!path || !out) return MI_ERR_INVALID;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { put_err(err, err_cap, std::string("open ") + path + ": " + strerror(errno)); return MI_ERR_IO; }
    mi_tar* h = new mi_tar();
    int rc;
    {
        mi_tarfile::Source src;
        if (!src.open_fd(fd)) { h->t.error = src.error; rc = MI_ERR_IO; }
        else {
            rc = mi_tarfile::parse(src, &h->t);
            if (is_gzip) *is_gzip = src.gz ? 1 : 0;
        }
    }
    close(fd);
    if (rc) {
        put_err(err, err_cap, std::string(path) + ": " + h->t.error);
        delete h;
        return rc;
    }
    for (const mi_tarfile::Item& it : h->t.items) h->rel.push_back(mi_tarfile::rel_name(it.name));
    *out = h;
    if (n_entries) *n_entries = h->t.items.size();
    return MI_OK;
}

int mi_tar_open(const char* path, mi_tar** out, uint64_t* n_entries) {
    return mi_tar_open_ex(path, out, n_entries, nullptr, nullptr, 0);
}

// gzip blob -> uncompressed tar file + its SHA-256 (the layer's tar digest).  A plain tar is copied.
int mi_tar_inflate(const char* blob_path, const char* tar_path_out, uint64_t* tar_bytes, uint8_t* tar_sha256,
                   uint8_t* blob_sha256, char* err, uint64_t err_cap) {
    if (!blob_path) return MI_ERR_INVALID;
    const int fd = open(blob_path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { put_err(err, err_cap, std::string("open ") + blob_path + ": " + strerror(errno)); return MI_ERR_IO; }
    int ofd = -1;
    if (tar_path_out) {
        ofd = open(tar_path_out, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
        if (ofd < 0) {
            put_err(err, err_cap, std::string("create ") + tar_path_out + ": " + strerror(errno));
            close(fd);
            return MI_ERR_IO;
        }
    }
    int rc = MI_OK;
    uint64_t total = 0;
    mi_host::Sha256 sha_tar, sha_blob;
    {
        mi_tarfile::Source src;
        std::vector<unsigned char> buf(1 << 20);
        if (!src.open_fd(fd)) { put_err(err, err_cap, src.error); rc = MI_ERR_IO; }
        while (!rc) {
            size_t got;
            if (src.gz) {
                got = src.inflate_some(buf.data(), buf.size());
                if (!src.error.empty()) { put_err(err, err_cap, std::string(blob_path) + ": " + src.error); rc = MI_ERR_INVALID; break; }
            } else {
                const uint64_t left = src.size - total;
                got = (size_t)(left < buf.size() ? left : buf.size());
                if (got && !src.read_at(total, buf.data(), got)) { put_err(err, err_cap, src.error); rc = MI_ERR_IO; break; }
            }
            if (got == 0) break;
            sha_tar.update(buf.data(), got);
            total += got;
            size_t w = 0;
            while (ofd >= 0 && w < got) {
                const ssize_t r = write(ofd, buf.data() + w, got - w);
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) { put_err(err, err_cap, std::string("write ") + tar_path_out + ": " + strerror(errno)); rc = MI_ERR_IO; break; }
                w += (size_t)r;
            }
        }
    }
    if (!rc && blob_sha256) {                                   // the blob's own digest (GzipDescriptor.Digest)
        std::vector<unsigned char> buf(1 << 20);
        uint64_t off = 0;
        for (;;) {
            const ssize_t r = pread(fd, buf.data(), buf.size(), (off_t)off);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) { put_err(err, err_cap, std::string("read ") + blob_path + ": " + strerror(errno)); rc = MI_ERR_IO; break; }
            if (r == 0) break;
            sha_blob.update(buf.data(), (size_t)r);
            off += (uint64_t)r;
        }
        if (!rc) sha_blob.final(blob_sha256);
    }
    close(fd);
    if (ofd >= 0) close(ofd);
    if (rc) return rc;
    if (tar_bytes) *tar_bytes = total;
    if (tar_sha256) sha_tar.final(tar_sha256);
    return MI_OK;
}

int mi_tar_entries(const mi_tar* tar, mi_tree_entry* out, uint64_t* data_offsets, uint64_t cap) {
    if (!tar || (cap && !out)) return MI_ERR_INVALID;
    const uint64_t n = tar->t.items.size();
    if (cap < n) return MI_ERR_CAPACITY;
    for (uint64_t i = 0; i < n; ++i) {
        const mi_tarfile::Item& it = tar->t.items[(size_t)i];
        out[i].relpath = tar->rel[(size_t)i].c_str();
        out[i].link_target = it.has_link ? it.link.c_str() : nullptr;
        out[i].file_index = it.file_index;
        out[i].size = it.size;
        out[i].mtime_sec = it.mtime;
        out[i].mode = it.mode;
        out[i].kind = it.kind;
        out[i].uid = it.uid;
        out[i].gid = it.gid;
        if (data_offsets) data_offsets[i] = it.data_off;
    }
    return MI_OK;
}

void mi_tar_free(mi_tar* tar) { delete tar; }

}  // extern "C"
