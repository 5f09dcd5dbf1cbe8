/* The ISA-L inflate API surface the reference sources name (fastqreader.h:34, fastqreader.cpp:88-149 and
 * :169-199, bgzf.h:177-185), implemented over zlib.
 *
 * TEST INFRASTRUCTURE ONLY.  ISA-L v2.31.1 is an un-vendored dependency of the reference and is not
 * installed in this image; this shim of our own lets the reference binaries built here
 * (oracle/_ref/fastp_ref*, oracle/build_ref.sh) read ".gz" inputs - plain gzip through
 * FastqReader::readToBufIgzip and bgzip-written files through BgzfMtReader - so that the tests can
 * compare compressed-input runs of the binding with the reference itself.  Only the behaviour the
 * reference relies on is provided:
 *   isal_read_gzip_header   parses one gzip member header at next_in (RFC 1952) and steps over it
 *   isal_inflate            crc_flag ISAL_GZIP_NO_HDR_VER: raw deflate data, then the 8-byte trailer
 *                           (CRC-32, ISIZE) is checked; block_state = ISAL_BLOCK_FINISH and bfinal = 1
 *                           once the trailer has been consumed, possibly over several calls
 *   isal_inflate_stateless  crc_flag ISAL_GZIP: one whole member in, its text out (total_out)
 * Nothing on the per-read hot path touches this API.
 */
#ifndef ORACLE_SHIM_IGZIP_LIB_H
#define ORACLE_SHIM_IGZIP_LIB_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define ISAL_DECOMP_OK 0
#define ISAL_END_INPUT 1
#define ISAL_INVALID_BLOCK (-1)
#define ISAL_UNSUPPORTED_METHOD (-5)
#define ISAL_INCORRECT_CHECKSUM (-6)
#define ISAL_GZIP_NO_HDR_VER 3
#define ISAL_GZIP 2
enum isal_block_state { ISAL_BLOCK_NEW_HDR = 0, ISAL_BLOCK_FINISH = 11 };

struct isal_gzip_header { uint32_t dummy; };

struct inflate_state {
    uint8_t* next_out;
    uint32_t avail_out;
    uint32_t total_out;
    uint8_t* next_in;
    uint64_t read_in;
    uint32_t avail_in;
    int32_t  read_in_length;
    uint32_t crc_flag;
    uint32_t crc;
    uint32_t hist_bits;
    enum isal_block_state block_state;
    uint32_t bfinal;
    /* the shim's own state */
    z_stream* shim_z;          /* raw inflater of the member being read */
    uint32_t shim_crc;         /* CRC-32 of the member's text so far */
    uint32_t shim_size;        /* its length mod 2^32 */
    uint32_t shim_body_done;   /* the deflate data has ended, the trailer is being collected */
    uint32_t shim_trailer_got;
    uint8_t shim_trailer[8];
};

static inline void isal_gzip_header_init(struct isal_gzip_header* h) { memset(h, 0, sizeof(*h)); }
static inline void isal_inflate_init(struct inflate_state* s) { memset(s, 0, sizeof(*s)); }
static inline void isal_inflate_reset(struct inflate_state* s) {
    z_stream* z = s->shim_z;
    memset(s, 0, sizeof(*s));
    if (z) {
        inflateReset(z);
        s->shim_z = z;
    }
}

/* length of the gzip header at p (n bytes available), 0 if it is not complete, -1 if it is not a gzip header */
static inline long oracle_shim_gzip_header_len(const uint8_t* p, size_t n) {
    if (n < 10) return 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return -1;
    const int flg = p[3];
    size_t at = 10;
    if (flg & 4) {
        if (at + 2 > n) return 0;
        at += 2 + ((size_t)p[at] | ((size_t)p[at + 1] << 8));
        if (at > n) return 0;
    }
    for (int pass = 0; pass < 2; pass++)
        if (flg & (pass == 0 ? 8 : 16)) {
            while (at < n && p[at]) at++;
            if (at >= n) return 0;
            at++;
        }
    if (flg & 2) at += 2;
    return at <= n ? (long)at : 0;
}

static inline int isal_read_gzip_header(struct inflate_state* s, struct isal_gzip_header* h) {
    (void)h;
    const long len = oracle_shim_gzip_header_len(s->next_in, s->avail_in);
    if (len < 0) return ISAL_UNSUPPORTED_METHOD;
    if (len == 0) return ISAL_END_INPUT;
    s->next_in += len;
    s->avail_in -= (uint32_t)len;
    return ISAL_DECOMP_OK;
}

static inline int isal_inflate(struct inflate_state* s) {
    if (s->block_state == ISAL_BLOCK_FINISH) return ISAL_DECOMP_OK;
    if (!s->shim_z) {
        s->shim_z = (z_stream*)calloc(1, sizeof(z_stream));
        if (!s->shim_z || inflateInit2(s->shim_z, -15) != Z_OK) return ISAL_INVALID_BLOCK;
    }
    if (!s->shim_body_done) {
        z_stream* z = s->shim_z;
        z->next_in = s->next_in;
        z->avail_in = s->avail_in;
        z->next_out = s->next_out;
        z->avail_out = s->avail_out;
        const int rc = inflate(z, Z_NO_FLUSH);
        const uint32_t made = s->avail_out - z->avail_out;
        s->shim_crc = (uint32_t)crc32(s->shim_crc, s->next_out, made);
        s->shim_size += made;
        s->total_out += made;
        s->next_in = z->next_in;
        s->avail_in = z->avail_in;
        s->next_out = z->next_out;
        s->avail_out = z->avail_out;
        if (rc == Z_STREAM_END) s->shim_body_done = 1;
        else if (rc != Z_OK && rc != Z_BUF_ERROR) return ISAL_INVALID_BLOCK;
    }
    if (s->shim_body_done) {
        while (s->shim_trailer_got < 8 && s->avail_in) {
            s->shim_trailer[s->shim_trailer_got++] = *s->next_in++;
            s->avail_in--;
        }
        if (s->shim_trailer_got == 8) {
            const uint8_t* t = s->shim_trailer;
            const uint32_t c = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t n = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            if (c != s->shim_crc || n != s->shim_size) return ISAL_INCORRECT_CHECKSUM;
            s->crc = c;
            s->block_state = ISAL_BLOCK_FINISH;
            s->bfinal = 1;
        }
    }
    return ISAL_DECOMP_OK;
}

static inline int isal_inflate_stateless(struct inflate_state* s) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, s->crc_flag == ISAL_GZIP ? 15 + 16 : -15) != Z_OK) return ISAL_INVALID_BLOCK;
    z.next_in = s->next_in;
    z.avail_in = s->avail_in;
    z.next_out = s->next_out;
    z.avail_out = s->avail_out;
    const int rc = inflate(&z, Z_FINISH);
    s->total_out = (uint32_t)z.total_out;
    s->next_in = z.next_in;
    s->avail_in = z.avail_in;
    s->next_out = z.next_out;
    s->avail_out = z.avail_out;
    inflateEnd(&z);
    if (rc != Z_STREAM_END) return ISAL_INVALID_BLOCK;
    s->block_state = ISAL_BLOCK_FINISH;
    s->bfinal = 1;
    return ISAL_DECOMP_OK;
}
#endif
