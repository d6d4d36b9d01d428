// fpt_bgzf.hpp -- the BGZF container (SAM/BAM specification 4.1) for the two host-side readers
// (fpt_bam.cpp: alignments; fpt_track.cpp: bgzip-compressed statistics tracks): a chain of
// independent gzip members of at most 64 KiB that carry their own compressed size in the "BC"
// extra subfield, which is what makes them inflatable in parallel and addressable by tabix's
// virtual offsets (member offset << 16 | offset inside the inflated member).
#pragma once

#include <zlib.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace fptz {

constexpr size_t kMaxBlock = 1 << 16;  // a BGZF member is at most 64 KiB, and so is what it holds

struct bgzf_block {
    size_t cpos;    // deflate payload (offset inside the buffer the header was parsed in)
    uint32_t clen;  // its length
    uint32_t isize, crc;
    size_t opos;    // where the inflated bytes go
};

// header of the gzip member at p (n bytes available): total member size through the BC subfield.
// Returns 0 when more bytes are needed, -1 when it is not a BGZF member.
inline long bgzf_member_size(const unsigned char *p, size_t n, size_t *payload_off) {
    if (n < 18) return 0;
    if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return -1;
    const size_t xlen = p[10] | (size_t)p[11] << 8;
    if (n < 12 + xlen) return xlen > 4096 ? -1 : 0;
    size_t q = 12;
    long bsize = -1;
    while (q + 4 <= 12 + xlen) {
        const size_t slen = p[q + 2] | (size_t)p[q + 3] << 8;
        if (q + 4 + slen > 12 + xlen) return -1;
        if (p[q] == 66 && p[q + 1] == 67 && slen == 2) bsize = (long)(p[q + 4] | (size_t)p[q + 5] << 8);
        q += 4 + slen;
    }
    if (bsize < 0) return -1;
    const long total = bsize + 1;
    if ((size_t)total < 12 + xlen + 8) return -1;  // no room for CRC32 + ISIZE
    *payload_off = 12 + xlen;
    return total;
}

// one member whole in memory at p (n bytes available): fills b (cpos relative to p) and returns its
// size, 0 / -1 as above; -1 also when ISIZE is beyond 64 KiB
inline long bgzf_parse_member(const unsigned char *p, size_t n, bgzf_block *b) {
    size_t pay = 0;
    const long sz = bgzf_member_size(p, n, &pay);
    if (sz <= 0) return sz;
    if ((size_t)sz > n) return 0;
    b->cpos = pay;
    b->clen = (uint32_t)(sz - (long)pay - 8);
    std::memcpy(&b->crc, p + sz - 8, 4);
    std::memcpy(&b->isize, p + sz - 4, 4);
    if (b->isize > kMaxBlock) return -1;
    return sz;
}

// raw deflate payload -> dst (isize bytes), CRC32 and both lengths checked
inline bool inflate_block(const unsigned char *src, uint32_t clen, unsigned char *dst, uint32_t isize, uint32_t crc) {
    if (isize == 0) return crc == 0;  // the end-of-file marker (and any other empty member) holds nothing
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<unsigned char *>(src);
    zs.avail_in = clen;
    zs.next_out = dst;
    zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = (rc == Z_STREAM_END) && zs.avail_out == 0 && zs.avail_in == 0;
    inflateEnd(&zs);
    return ok && (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, isize) == crc;
}

// the first `want` bytes (or all, if fewer) of what a raw deflate payload inflates to: a look at how
// a member begins without inflating it (no CRC check: a look, not a read).  Returns the bytes made.
inline uint32_t inflate_prefix(const unsigned char *src, uint32_t clen, unsigned char *dst, uint32_t want) {
    if (want == 0) return 0;
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return 0;
    zs.next_in = const_cast<unsigned char *>(src);
    zs.avail_in = clen;
    zs.next_out = dst;
    zs.avail_out = want;
    const int rc = inflate(&zs, Z_SYNC_FLUSH);
    const uint32_t got = want - zs.avail_out;
    inflateEnd(&zs);
    return (rc == Z_OK || rc == Z_STREAM_END || rc == Z_BUF_ERROR) ? got : 0;
}

}  // namespace fptz
