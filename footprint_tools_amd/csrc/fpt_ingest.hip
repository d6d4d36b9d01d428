// fpt_ingest.hip -- cut-count ingestion on the device: alignments -> per-base
// 5' cut counts of a batch of intervals, written straight into the padded CSR arrays the fused
// scan reads (what `prediction.compute` fetches per interval, modeling/predict.pyx:132-140).
//
// Reference rule (cutcounts.py): a read that passes `validate_read` (:119-145: QC-fail and
// duplicate flags as configured, MAPQ >= min_qual) and, when paired, is a proper pair and neither
// secondary nor supplementary (:196-205), adds one cut at
//     reference_start + offset[0]      on '+'  (forward reads)
//     reference_end   + offset[1]      on '-'  (reverse reads)          (:231-248)
// with offset = (0, -1) by default; `lookup` (:274-313) returns the counts of [start, end).  Every
// read is counted once whether or not its mate lies in the fetched window (the generator flushes
// unpaired mates, :226-229), so the count at a position does not depend on the interval asked for:
// counts[x] = number of valid reads whose cut position is x.  That is what this kernel adds up,
// one lane per read, scattering into every interval of the batch whose padded range holds x.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fpt.h"

int fpt_internal_fail(int code, const char *fmt, ...);  // fpt_capi.cpp
hipStream_t fpt_internal_stream(fpt_ctx *c);
int fpt_internal_check_ctx(fpt_ctx *c);

namespace {

constexpr uint16_t kPaired = 0x1, kProper = 0x2, kUnmapped = 0x4, kReverse = 0x10, kSecondary = 0x100,
                   kQcFail = 0x200, kDup = 0x400, kSupplementary = 0x800;

// intervals sorted by key = (reference id << 32 | padded start); `maxend[i]` = running maximum of
// the padded end keys of intervals 0..i, so that a walk back from the last interval starting at or
// before x can stop as soon as no earlier interval reaches x
__global__ void __launch_bounds__(256) k_cut_counts(fpt_cutcount_desc d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n_reads) return;
    const uint16_t fl = d.flag[i];
    if (fl & kUnmapped) return;  // no reference_end: the reference's int(None) would raise
    if (d.remove_qcfail && (fl & kQcFail)) return;
    if (d.remove_dups && (fl & kDup)) return;
    if ((int)d.mapq[i] < d.min_qual) return;
    if ((fl & kPaired) && (!(fl & kProper) || (fl & (kSecondary | kSupplementary)))) return;
    const bool rev = (fl & kReverse) != 0;
    // a reverse read without a reference-consuming CIGAR operation has no reference_end (the reader
    // hands it over as -1; pysam gives None and the reference's lookup fails on it): left out
    if (rev && d.ref_end[i] < 0) return;
    const int64_t pos = rev ? (int64_t)d.ref_end[i] + d.offset_minus : (int64_t)d.ref_start[i] + d.offset_plus;
    if (pos < 0) return;
    const int64_t x = ((int64_t)d.ref_id[i] << 32) | pos;
    // last interval whose padded start key is <= x
    int64_t lo = 0, hi = d.n_intervals;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (d.start_key[mid] <= x) lo = mid + 1; else hi = mid;
    }
    for (int64_t j = lo - 1; j >= 0 && d.maxend_key[j] > x; --j) {
        int64_t rel = x - d.start_key[j];
        const int64_t plen = d.padded_len[j];
        if (rel < plen) {
            // an interval on strand '-' gets its arrays mirrored and swapped (cutcounts.py:307-311)
            const bool flip = d.flip && d.flip[j];
            if (flip) rel = plen - 1 - rel;
            double *dst = (rev != flip) ? d.counts_minus : d.counts_plus;
            atomicAdd(dst + d.counts_off[j] + rel, 1.0);
        }
    }
}

// Sequence gather: the ASCII bytes of every interval's [start, start + len) of its chromosome, back
// to back, from the bytes of a FASTA file resident on the device -- what `fasta_func.fetch` returns
// per interval (modeling/predict.pyx:136-140), for a whole batch.  A chromosome is described by its
// .fai line (length, byte offset of the first base, bases per line, bytes per line); positions
// outside it read as 'N' (pysam truncates; the scan then uses the default propensity).  One
// workgroup per interval.
__global__ void __launch_bounds__(256) k_seq_gather(const uint8_t *__restrict__ fasta, int64_t fasta_bytes,
                                                    const int64_t *__restrict__ iv, int64_t n_iv,
                                                    uint8_t *__restrict__ out) {
    const int64_t i = blockIdx.x;
    if (i >= n_iv) return;
    const int64_t *d = iv + i * 7;  // start, len, out offset, chromosome length, first-base offset, line bases, line bytes
    const int64_t start = d[0], n = d[1], o = d[2], clen = d[3], coff = d[4], lb = d[5], lw = d[6];
    for (int64_t k = threadIdx.x; k < n; k += blockDim.x) {
        const int64_t g = start + k;
        uint8_t ch = 'N';
        if (g >= 0 && g < clen && lb > 0) {
            const int64_t pos = coff + (g / lb) * lw + g % lb;
            if (pos >= 0 && pos < fasta_bytes) ch = fasta[pos];
        }
        out[o + k] = ch;
    }
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int fpt_seq_gather_dev(fpt_ctx *c, const uint8_t *fasta_dev, int64_t fasta_bytes, const int64_t *intervals_dev,
                       int64_t n_intervals, uint8_t *seq_out_dev) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    if (n_intervals < 0 || fasta_bytes < 0) return fpt_internal_fail(FPT_ERR_INVALID, "negative size");
    if (n_intervals == 0) return FPT_OK;
    if (!fasta_dev || !intervals_dev || !seq_out_dev) return fpt_internal_fail(FPT_ERR_INVALID, "null buffer");
    if (n_intervals > 0x7fffffff) return fpt_internal_fail(FPT_ERR_INVALID, "too many intervals in one call");
    hipLaunchKernelGGL(k_seq_gather, dim3((unsigned)n_intervals), dim3(256), 0, fpt_internal_stream(c), fasta_dev,
                       fasta_bytes, intervals_dev, n_intervals, seq_out_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fpt_internal_fail(FPT_ERR_HIP, "k_seq_gather launch failed: %s", hipGetErrorString(e));
    return FPT_OK;
}

int fpt_cut_counts_dev(fpt_ctx *c, const fpt_cutcount_desc *d) {
    if (int rc = fpt_internal_check_ctx(c)) return rc;
    if (!d) return fpt_internal_fail(FPT_ERR_INVALID, "null descriptor");
    if (d->n_reads < 0 || d->n_intervals < 0) return fpt_internal_fail(FPT_ERR_INVALID, "negative size");
    if (d->n_reads == 0 || d->n_intervals == 0) return FPT_OK;
    if (!d->ref_id || !d->ref_start || !d->ref_end || !d->flag || !d->mapq)
        return fpt_internal_fail(FPT_ERR_INVALID, "null read arrays");
    if (!d->start_key || !d->maxend_key || !d->padded_len || !d->counts_off || !d->counts_plus || !d->counts_minus)
        return fpt_internal_fail(FPT_ERR_INVALID, "null interval / count arrays");
    const int64_t blocks = (d->n_reads + 255) / 256;
    if (blocks > 0x7fffffff) return fpt_internal_fail(FPT_ERR_INVALID, "too many reads in one call");
    hipLaunchKernelGGL(k_cut_counts, dim3((unsigned)blocks), dim3(256), 0, fpt_internal_stream(c), *d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fpt_internal_fail(FPT_ERR_HIP, "k_cut_counts launch failed: %s", hipGetErrorString(e));
    return FPT_OK;
}

#pragma GCC visibility pop
}
