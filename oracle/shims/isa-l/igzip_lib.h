/* Stub of the ISA-L inflate API surface that the reference sources name
 * (fastqreader.h:34, fastqreader.cpp:101-193, bgzf.h:177-185).
 *
 * TEST INFRASTRUCTURE ONLY.  ISA-L v2.31.1 is an un-vendored dependency of the
 * reference and is not installed in this image; the reference oracle binary
 * (oracle/_ref/fastp_ref) is only ever fed plain-text FASTQ, so every entry
 * point here simply reports failure.  Nothing on the per-read hot path touches
 * this API.
 */
#ifndef ORACLE_SHIM_IGZIP_LIB_H
#define ORACLE_SHIM_IGZIP_LIB_H
#include <stdint.h>
#include <string.h>

#define ISAL_DECOMP_OK 0
#define ISAL_END_INPUT 1
#define ISAL_UNSUPPORTED_METHOD (-5)
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
};

static inline void isal_gzip_header_init(struct isal_gzip_header* h) { memset(h, 0, sizeof(*h)); }
static inline void isal_inflate_init(struct inflate_state* s) { memset(s, 0, sizeof(*s)); }
static inline void isal_inflate_reset(struct inflate_state* s) { memset(s, 0, sizeof(*s)); }
static inline int isal_read_gzip_header(struct inflate_state* s, struct isal_gzip_header* h) {
    (void)s; (void)h; return ISAL_UNSUPPORTED_METHOD;
}
static inline int isal_inflate(struct inflate_state* s) { (void)s; return ISAL_UNSUPPORTED_METHOD; }
static inline int isal_inflate_stateless(struct inflate_state* s) { (void)s; return ISAL_UNSUPPORTED_METHOD; }
#endif
