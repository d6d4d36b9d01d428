// fpt_bam.cpp -- host-side alignment reader for the cut-count ingestion (no htslib in this image:
// BGZF is a chain of gzip members, which zlib inflates; the BAM record layout is that of the
// SAM/BAM specification, section 4.2).  Sequential: one pass over the file hands every alignment's
// (reference id, start, end, flag, MAPQ) to the caller in batches; the cut position and the read
// filters of the reference (cutcounts.py:119-145, 196-205, 231-248) are applied on the device by
// k_cut_counts.  reference_end = start + reference-consuming CIGAR operations (M, D, N, =, X), as
// pysam computes it.
//
// PARITY UNPINNED for this reader: pysam / htslib are not in the image and the reference ships no
// alignment fixtures, so it is tested on BAM files written by the tests themselves.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fpt.h"

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp

struct fpt_bam {
    FILE *f = nullptr;
    z_stream zs;
    bool zs_live = false, eof = false;
    std::vector<unsigned char> in, out;  // compressed chunk, decompressed bytes not consumed yet
    size_t in_pos = 0, in_len = 0, out_pos = 0;
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    std::string error;

    // make at least `n` decompressed bytes available at out[out_pos..]; false at end of file
    bool need(size_t n) {
        while (out.size() - out_pos < n) {
            if (out_pos > (1u << 20)) {  // drop what has been consumed
                out.erase(out.begin(), out.begin() + (long)out_pos);
                out_pos = 0;
            }
            if (in_pos == in_len) {
                if (eof) return false;
                in_len = fread(in.data(), 1, in.size(), f);
                in_pos = 0;
                if (in_len == 0) {
                    eof = true;
                    if (zs_live) error = "truncated BGZF block at the end of the file";
                    return false;
                }
            }
            if (!zs_live) {
                std::memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, 15 + 32) != Z_OK) {  // gzip / zlib header detected by zlib
                    error = "inflateInit2 failed";
                    return false;
                }
                zs_live = true;
            }
            unsigned char buf[1 << 16];
            zs.next_in = in.data() + in_pos;
            zs.avail_in = (uInt)(in_len - in_pos);
            zs.next_out = buf;
            zs.avail_out = sizeof buf;
            int rc = inflate(&zs, Z_NO_FLUSH);
            in_pos = in_len - zs.avail_in;
            out.insert(out.end(), buf, buf + (sizeof buf - zs.avail_out));
            if (rc == Z_STREAM_END) {  // end of one BGZF block: the next member starts a new stream
                inflateEnd(&zs);
                zs_live = false;
            } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                error = "corrupt BGZF block";
                return false;
            }
        }
        return true;
    }
    template <typename T>
    T get() {
        T v;
        std::memcpy(&v, out.data() + out_pos, sizeof(T));
        out_pos += sizeof(T);
        return v;
    }
};

extern "C" {
#pragma GCC visibility push(default)

int fpt_bam_open(const char *path, fpt_bam **out) {
    if (!path || !out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fpt_internal_fail(FPT_ERR_INVALID, "Cannot open BAM file: %s", path);  // cutcounts.py:103
    fpt_bam *b = new fpt_bam();
    b->f = f;
    b->in.resize(1 << 18);
    auto bad = [&](const char *what) {
        std::string msg = b->error.empty() ? what : b->error;
        fclose(f);
        delete b;
        return fpt_internal_fail(FPT_ERR_INVALID, "%s: %s", path, msg.c_str());
    };
    if (!b->need(12) || std::memcmp(b->out.data() + b->out_pos, "BAM\1", 4) != 0) return bad("not a BAM file");
    b->out_pos += 4;
    const int32_t l_text = b->get<int32_t>();
    if (l_text < 0 || !b->need((size_t)l_text + 4)) return bad("truncated header");
    b->out_pos += (size_t)l_text;
    const int32_t n_ref = b->get<int32_t>();
    if (n_ref < 0) return bad("bad reference count");
    for (int i = 0; i < n_ref; ++i) {
        if (!b->need(4)) return bad("truncated reference list");
        const int32_t l_name = b->get<int32_t>();
        if (l_name <= 0 || !b->need((size_t)l_name + 4)) return bad("truncated reference list");
        b->ref_names.emplace_back((const char *)b->out.data() + b->out_pos, (size_t)l_name - 1);
        b->out_pos += (size_t)l_name;
        b->ref_lens.push_back(b->get<int32_t>());
    }
    *out = b;
    return FPT_OK;
}

int fpt_bam_close(fpt_bam *b) {
    if (!b) return FPT_OK;
    if (b->zs_live) inflateEnd(&b->zs);
    if (b->f) fclose(b->f);
    delete b;
    return FPT_OK;
}

int fpt_bam_n_refs(fpt_bam *b, int32_t *n_out) {
    if (!b || !n_out) return fpt_internal_fail(FPT_ERR_INVALID, "null argument");
    *n_out = (int32_t)b->ref_names.size();
    return FPT_OK;
}

int fpt_bam_ref(fpt_bam *b, int32_t i, char *name_out, int32_t cap, int64_t *len_out) {
    if (!b || i < 0 || i >= (int32_t)b->ref_names.size()) return fpt_internal_fail(FPT_ERR_INVALID, "bad reference index");
    if (name_out && cap > 0) {
        std::strncpy(name_out, b->ref_names[i].c_str(), (size_t)cap - 1);
        name_out[cap - 1] = 0;
    }
    if (len_out) *len_out = b->ref_lens[i];
    return FPT_OK;
}

int fpt_bam_read(fpt_bam *b, int64_t max_reads, int32_t *ref_id, int32_t *ref_start, int32_t *ref_end,
                 uint16_t *flag, uint8_t *mapq, int64_t *n_out) {
    if (!b || !n_out || max_reads < 0 || (max_reads > 0 && (!ref_id || !ref_start || !ref_end || !flag || !mapq)))
        return fpt_internal_fail(FPT_ERR_INVALID, "bad arguments");
    int64_t n = 0;
    while (n < max_reads) {
        if (!b->need(4)) break;  // end of file (or an error, reported below)
        const int32_t block = b->get<int32_t>();
        if (block < 32 || !b->need((size_t)block)) {
            b->error = "truncated alignment record";
            break;
        }
        const size_t rec = b->out_pos;
        const int32_t rid = b->get<int32_t>(), pos = b->get<int32_t>();
        const uint8_t l_name = b->get<uint8_t>(), mq = b->get<uint8_t>();
        (void)b->get<uint16_t>();  // bin
        const uint16_t n_cig = b->get<uint16_t>(), fl = b->get<uint16_t>();
        b->out_pos = rec + 32;  // l_seq, next_refID, next_pos, tlen are not needed
        int64_t span = 0;
        if ((size_t)32 + l_name + 4u * n_cig <= (size_t)block) {
            const unsigned char *cig = b->out.data() + rec + 32 + l_name;
            for (int k = 0; k < n_cig; ++k) {
                uint32_t v;
                std::memcpy(&v, cig + 4 * k, 4);
                const uint32_t op = v & 0xf;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += v >> 4;  // M D N = X
            }
        }
        b->out_pos = rec + (size_t)block;
        ref_id[n] = rid;
        ref_start[n] = pos;
        ref_end[n] = (int32_t)(pos + span);
        flag[n] = fl;
        mapq[n] = mq;
        ++n;
    }
    *n_out = n;
    if (!b->error.empty()) return fpt_internal_fail(FPT_ERR_INVALID, "%s", b->error.c_str());
    return FPT_OK;
}

#pragma GCC visibility pop
}
