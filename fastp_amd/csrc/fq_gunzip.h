// fq_gunzip.h - a streaming gzip inflater for the HOST side of the file loop (fq_stream.cpp): the bytes of a ".gz" input that
// is not bgzip-written, member after member, to text.
//
// Reference code this stands in for: FastqReader::readToBufIgzip (src/fastqreader.cpp:88-149) = ISA-L's igzip on the
// reference's reader thread (isal_read_gzip_header / isal_inflate, CRC-32 and ISIZE of every member checked).  A general
// gzip stream has no member boundaries that can be found without decoding it, so it cannot be cut up for the device the way
// bgzip's members are (fastp_gpu_inflate_bgzf); it is inflated on the host - by this class on one thread (pipes; the tables and
// the CRC also serve fq_pgunzip.h, which puts several threads on one stream of a regular file) - zlib's inflate does ~0.35 GB/s of FASTQ text on the build host, this one ~2 x that (tools/gunzip_bench.cpp),
// by the usual means: a 64-bit bit buffer refilled with one unaligned load, an 11-bit direct table whose entries carry the
// literal / length base / extra-bit count, matches copied eight bytes at a time, CRC-32 by carry-less multiplication.
//
// Contract: RFC 1951 / RFC 1952 exactly.  Stored, fixed and dynamic blocks; any number of members one behind the other;
// over-subscribed or incomplete code sets, distances beyond the window start, codes 286/287 and 30/31, LEN/NLEN
// mismatches, a wrong CRC-32 or ISIZE, bytes that are no gzip header behind a member and a file that ends inside a member
// are errors (the caller ends the run with them).  tests/test_gunzip.py compares it with zlib on every block type, level
// and strategy, on random data, on damaged streams and across arbitrary read / refill boundaries.
#pragma once
#include <errno.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fqgz {

// ---- CRC-32 (IEEE 802.3, reflected; what gzip trailers hold) ---------------------------------------------------------------
struct Crc32 {
    uint32_t t[8][256];
    bool clmul;
    Crc32() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFFu];
#if defined(__x86_64__)
        clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
#else
        clmul = false;
#endif
    }
    uint32_t bytes(uint32_t c, const uint8_t* p, size_t n) const {   // c: running value in its inverted form
        while (n >= 8) {
            uint32_t lo, hi;
            memcpy(&lo, p, 4);
            memcpy(&hi, p + 4, 4);
            lo ^= c;
            c = t[7][lo & 0xFFu] ^ t[6][(lo >> 8) & 0xFFu] ^ t[5][(lo >> 16) & 0xFFu] ^ t[4][lo >> 24] ^ t[3][hi & 0xFFu] ^ t[2][(hi >> 8) & 0xFFu] ^
                t[1][(hi >> 16) & 0xFFu] ^ t[0][hi >> 24];
            p += 8;
            n -= 8;
        }
        while (n--) c = (c >> 8) ^ t[0][(c ^ *p++) & 0xFFu];
        return c;
    }
#if defined(__x86_64__)
    // folding by carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"):
    // four 128-bit lanes folded 64 bytes at a time, then one lane, then Barrett reduction.  Constants for the reflected
    // polynomial 0x1DB710641: x^(4*128+32), x^(4*128-32), x^(128+32), x^(128-32), x^64 mod P, and P' / mu.
    __attribute__((target("pclmul,sse4.1"))) static __m128i fold1(__m128i x, __m128i next, __m128i k) {
        const __m128i a = _mm_clmulepi64_si128(x, k, 0x00);
        x = _mm_clmulepi64_si128(x, k, 0x11);
        return _mm_xor_si128(_mm_xor_si128(x, a), next);
    }
    __attribute__((target("pclmul,sse4.1"))) uint32_t clmul_blocks(uint32_t c, const uint8_t* p, size_t n) const {   // n: a multiple of 16, >= 64
        const __m128i k1k2 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);
        const __m128i k3k4 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
        const __m128i k5 = _mm_set_epi64x(0, 0x0000000163cd6124ll);
        const __m128i poly = _mm_set_epi64x(0x00000001F7011641ll, 0x00000001DB710641ll);
        const __m128i mask32 = _mm_set_epi32(0, 0, 0, -1);
        __m128i x0 = _mm_loadu_si128((const __m128i*)p), x1 = _mm_loadu_si128((const __m128i*)(p + 16)), x2 = _mm_loadu_si128((const __m128i*)(p + 32)),
                x3 = _mm_loadu_si128((const __m128i*)(p + 48));
        x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)c));
        p += 64;
        n -= 64;
        while (n >= 64) {
            __m128i a0 = _mm_clmulepi64_si128(x0, k1k2, 0x00), a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00),
                    a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00);
            x0 = _mm_clmulepi64_si128(x0, k1k2, 0x11);
            x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
            x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
            x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
            x0 = _mm_xor_si128(_mm_xor_si128(x0, a0), _mm_loadu_si128((const __m128i*)p));
            x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i*)(p + 16)));
            x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i*)(p + 32)));
            x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i*)(p + 48)));
            p += 64;
            n -= 64;
        }
        // four lanes -> one
        __m128i x = fold1(x0, x1, k3k4);
        x = fold1(x, x2, k3k4);
        x = fold1(x, x3, k3k4);
        while (n >= 16) {
            x = fold1(x, _mm_loadu_si128((const __m128i*)p), k3k4);
            p += 16;
            n -= 16;
        }
        // 128 -> 64 bits
        __m128i a = _mm_clmulepi64_si128(x, k3k4, 0x10);
        x = _mm_xor_si128(_mm_srli_si128(x, 8), a);
        a = _mm_and_si128(x, mask32);
        x = _mm_srli_si128(x, 4);
        a = _mm_clmulepi64_si128(a, k5, 0x00);
        x = _mm_xor_si128(x, a);
        // Barrett reduction 64 -> 32 bits
        a = _mm_and_si128(x, mask32);
        a = _mm_clmulepi64_si128(a, poly, 0x10);
        a = _mm_and_si128(a, mask32);
        a = _mm_clmulepi64_si128(a, poly, 0x00);
        x = _mm_xor_si128(x, a);
        return (uint32_t)_mm_extract_epi32(x, 1);
    }
#endif
    uint32_t update(uint32_t crc, const uint8_t* p, size_t n) const {   // zlib's crc32() convention
        uint32_t c = ~crc;
#if defined(__x86_64__)
        if (clmul && n >= 128) {
            const size_t blk = n & ~(size_t)15;
            c = clmul_blocks(c, p, blk);
            p += blk;
            n -= blk;
        }
#endif
        return ~bytes(c, p, n);
    }
};

// ---- decoding tables ----------------------------------------------------------------------------------------------------------
// entry (u32):  bits 0..7   bits this entry consumes (code length [+ extra bits of a length / distance symbol];
//                           for a subtable pointer: the root bits)
//               bits 8..11  code length alone (length / distance symbols: what to shift the saved bits by to reach the extra bits);
//                           for a subtable pointer: the subtable's index bits
//               bits 12..15 kind: E_LIT, E_LEN (also distances), E_EOB, E_SUB, 0 = no such code
//               bits 16..31 literal | length base | distance base | subtable start
enum { E_LIT = 1 << 12, E_LEN = 2 << 12, E_EOB = 4 << 12, E_SUB = 8 << 12 };
enum { LROOT = 11, DROOT = 8, LTAB = (1 << LROOT) + 288 * 16, DTAB = (1 << DROOT) + 32 * 128 };

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// n code lengths -> table.  kind: 0 literal/length, 1 distance, 2 the code-length alphabet (entries = E_LIT | symbol).
// false: over-subscribed, or incomplete where RFC 1951 readers (zlib: inftrees.c) do not accept it - an incomplete set is
// legal only as a single code of one bit (a distance alphabet with one symbol, or none at all).
inline bool build_table(const uint8_t* lens, int n, int kind, uint32_t* tab, int root) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    int maxlen = 15;
    while (maxlen > 0 && !count[maxlen]) maxlen--;
    const int rsize = 1 << root;
    if (maxlen == 0) {   // no codes at all: legal for distances (a block of literals only); every lookup is "no such code"
        for (int i = 0; i < rsize; i++) tab[i] = 0;
        return kind == 1;
    }
    int left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    if (left > 0 && (kind == 2 || maxlen != 1)) return false;
    uint32_t next[16];
    uint32_t code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; l++) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    for (int i = 0; i < rsize; i++) tab[i] = 0;
    // subtables: one per root prefix that has longer codes, sized by the longest code under it
    uint8_t subbits[1 << LROOT];
    if (maxlen > root) {
        memset(subbits, 0, (size_t)rsize);
        uint32_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t c = nx[l]++;
            if (l <= root) continue;
            uint32_t rev = 0;
            for (int b = 0; b < l; b++) rev |= ((c >> b) & 1u) << (l - 1 - b);
            uint8_t& sb = subbits[rev & (uint32_t)(rsize - 1)];
            sb = (uint8_t)std::max<int>(sb, l - root);
        }
        uint32_t at = (uint32_t)rsize;
        for (int i = 0; i < rsize; i++)
            if (subbits[i]) {
                tab[i] = E_SUB | (at << 16) | ((uint32_t)subbits[i] << 8) | (uint32_t)root;
                for (uint32_t k = 0; k < (1u << subbits[i]); k++) tab[at + k] = 0;
                at += 1u << subbits[i];
            }
    }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = next[l]++;
        uint32_t rev = 0;
        for (int b = 0; b < l; b++) rev |= ((c >> b) & 1u) << (l - 1 - b);
        uint32_t e;
        if (kind == 2) {
            e = E_LIT | ((uint32_t)s << 16);
        } else if (kind == 1) {
            if (s >= 30) e = 0;   // distance codes 30 / 31 never occur in valid data: decoding one is an error
            else e = E_LEN | ((uint32_t)DIST_BASE[s] << 16) | (uint32_t)DIST_EXTRA[s];
        } else if (s < 256) {
            e = E_LIT | ((uint32_t)s << 16);
        } else if (s == 256) {
            e = E_EOB;
        } else if (s < 286) {
            e = E_LEN | ((uint32_t)LEN_BASE[s - 257] << 16) | (uint32_t)LEN_EXTRA[s - 257];
        } else {
            e = 0;
        }
        if (l <= root) {
            if (e) e = (e & ~0xFFu) | (((e & 0xFFu) + (uint32_t)l) & 0xFFu) | ((uint32_t)l << 8);   // bits consumed = code + extra; code length apart
            for (uint32_t k = rev; k < (uint32_t)rsize; k += 1u << l) tab[k] = e;
        } else {
            const uint32_t p = tab[rev & (uint32_t)(rsize - 1)];
            const uint32_t start = p >> 16, sb = (p >> 8) & 15u;
            const int sl = l - root;   // bits of the code inside the subtable
            if (e) e = (e & ~0xFFu) | (((e & 0xFFu) + (uint32_t)sl) & 0xFFu) | ((uint32_t)sl << 8);
            for (uint32_t k = rev >> root; k < (1u << sb); k += 1u << sl) tab[start + k] = e;
        }
    }
    return true;
}

// ---- the inflater -------------------------------------------------------------------------------------------------------------
// error codes of Gunzip::read (negative return): 1 = reading the file failed, 4 = damaged stream / file ends inside a member
class Gunzip {
  public:
    int fd = -1;
    int64_t fpos = 0, fsize = 0;   // file bytes read so far / the file's size
    bool seekable = true;          // false: a pipe (--stdin) - read() in sequence until it returns 0, fsize is not used
    bool at_eof = false;           // every member has been delivered and the file has ended

    // FASTP_GPU_STREAM_GUNZIP_INCAP_KB: the input buffer's size (tests: a small one, so that small files cross many refills)
    Gunzip() : in_cap_(in_cap_from_env()), in_(in_cap_ + 64), win_((size_t)WIN + OUT_CAP + 512) {}

    // up to `want` bytes of text to dst; fewer only at the end of the file; < 0: error (*err as above)
    int64_t read(uint8_t* dst, int64_t want, int* err) {
        int64_t made = 0;
        while (made < want) {
            if (deliver_ < op_) {   // text decoded earlier
                const int64_t n = std::min<int64_t>(want - made, (int64_t)(op_ - deliver_));
                memcpy(dst + made, win_.data() + deliver_, (size_t)n);
                deliver_ += (size_t)n;
                made += n;
                continue;
            }
            if (at_eof) break;
            const int rc = step();
            if (rc) { *err = rc; return -1; }
        }
        return made;
    }

  private:
    enum { IN_CAP = 4 << 20, WIN = 32768, OUT_CAP = 1 << 20 };
    enum State { S_HEADER, S_BLOCK, S_STORED, S_CODES, S_TRAILER };
    static size_t in_cap_from_env() {
        const char* e = getenv("FASTP_GPU_STREAM_GUNZIP_INCAP_KB");
        const long kb = e ? atol(e) : 0;
        return kb > 0 ? (size_t)std::max(192L, kb) << 10 : (size_t)IN_CAP;   // (a header asks for 64 KiB at once)
    }
    size_t in_cap_;
    std::vector<uint8_t> in_, win_;
    size_t ip_ = 0, in_len_ = 0;       // unread input = in_[ip_, in_len_)
    bool file_done_ = false;           // the file has been read to its end (the input then carries 64 zero bytes of padding)
    uint64_t bb_ = 0;                  // bit buffer; bits beyond bc_ may hold garbage from the refill and are never trusted
    int bc_ = 0;
    size_t op_ = WIN, deliver_ = WIN;  // decoded text = win_[.., op_); [deliver_, op_) not yet handed out
    size_t mstart_ = WIN;              // where the current member's text starts in win_ (0: further back than the buffer reaches);
                                       // a distance must not reach below it - a member is a stream of its own
    State st_ = S_HEADER;
    bool last_block_ = false;
    bool member_seen_ = false;
    uint32_t stored_left_ = 0;
    uint32_t crc_ = 0, isize_ = 0;
    size_t crc_from_ = WIN;            // text of the current member in win_[crc_from_, op_) is not in crc_ yet
    uint32_t lt_[LTAB], dt_[DTAB];
    static const Crc32& crc_tab() {
        static const Crc32 c;
        return c;
    }

    // make at least `need` unread bytes available (or everything up to the end of the file); false: read error
    bool fill(size_t need) {
        while (in_len_ - ip_ < need && !file_done_) {
            // the bytes whose bits still sit in the bit buffer stay in front of ip_: unread_bits() steps back over them
            const size_t keep = std::min(ip_, (size_t)((bc_ + 7) >> 3));
            if (ip_ > keep) {
                const size_t drop = ip_ - keep;
                memmove(in_.data(), in_.data() + drop, in_len_ - drop);
                in_len_ -= drop;
                ip_ = keep;
            }
            const int64_t room = (int64_t)in_cap_ - (int64_t)in_len_;
            const int64_t ask = seekable ? std::min<int64_t>(room, fsize - fpos) : room;
            if (ask <= 0) {
                file_done_ = true;
                break;
            }
            int64_t got = 0;
            while (got < ask) {
                const ssize_t r = seekable ? pread(fd, in_.data() + in_len_ + got, (size_t)(ask - got), (off_t)(fpos + got))
                                           : ::read(fd, in_.data() + in_len_ + got, (size_t)(ask - got));
                if (r < 0 && errno == EINTR) continue;
                if (r < 0 || (r == 0 && seekable)) return false;
                if (r == 0) { file_done_ = true; break; }
                got += r;
                if (!seekable) break;   // a pipe hands over what it has: decode that first
            }
            fpos += got;
            in_len_ += (size_t)got;
            if (seekable && fpos >= fsize) file_done_ = true;
        }
        if (file_done_) memset(in_.data() + in_len_, 0, 64);   // loads past the end read zeros; bit accounting catches an overrun
        return true;
    }
    size_t avail() const { return in_len_ - ip_; }

    inline void refill() {   // >= 56 valid bits afterwards (the buffer is padded: the load never leaves it)
        uint64_t w;
        memcpy(&w, in_.data() + ip_, 8);
        bb_ |= w << bc_;
        ip_ += (size_t)((63 - bc_) >> 3);
        bc_ |= 56;
    }
    // the next unconsumed bit sits at 8 * ip_ - bc_ of the input (refill keeps that invariant): past the end of the data?
    bool overrun() const { return 8 * (int64_t)ip_ - (int64_t)bc_ > 8 * (int64_t)in_len_; }
    // put whole unread bytes of the bit buffer back (before byte-wise reads: stored blocks, trailer)
    void unread_bits() {
        const int drop = bc_ & 7;   // to the byte boundary
        bb_ >>= drop;
        bc_ -= drop;
        ip_ -= (size_t)(bc_ >> 3);
        bb_ = 0;
        bc_ = 0;
    }

    void flush_crc() {
        if (op_ > crc_from_) {
            crc_ = crc_tab().update(crc_, win_.data() + crc_from_, op_ - crc_from_);
            isize_ += (uint32_t)(op_ - crc_from_);
            crc_from_ = op_;
        }
    }
    // room for more text (read() calls step() only when everything decoded so far has been handed out): when the buffer is
    // nearly full, keep the last WIN bytes as history
    void make_room() {
        if (op_ + 1024 <= win_.size() - 512) return;
        flush_crc();
        const size_t shift = op_ - WIN;
        memmove(win_.data(), win_.data() + shift, WIN);
        mstart_ = mstart_ > shift ? mstart_ - shift : 0;
        op_ = deliver_ = crc_from_ = WIN;
    }

    // one unit of progress: a header, a block header, a run of symbols, a piece of a stored block, a trailer.  0 = fine.
    int step() {
        switch (st_) {
            case S_HEADER: {
                if (!fill(1 << 16)) return 1;
                if (avail() == 0) {
                    // a file that ends before its first member header is no gzip file: FastqReader::init stops there
                    // ("igzip: Error invalid gzip header found", fastqreader.cpp:193-196)
                    if (!member_seen_) return 4;
                    at_eof = true;
                    return 0;
                }
                member_seen_ = true;
                const uint8_t* p = in_.data() + ip_;
                const size_t n = avail();
                if (n < 18 && !file_done_) return 4;
                if (n < 10 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return 4;
                const int flg = p[3];
                size_t at = 10;
                if (flg & 4) {
                    if (at + 2 > n) return 4;
                    at += 2 + ((size_t)p[at] | ((size_t)p[at + 1] << 8));
                }
                for (int pass = 0; pass < 2; pass++)
                    if (flg & (pass == 0 ? 8 : 16)) {
                        while (at < n && p[at]) at++;
                        at++;
                    }
                if (flg & 2) at += 2;
                if (at > n) return 4;   // (a header longer than 64 KiB of name / comment is not what FASTQ files carry)
                ip_ += at;
                bb_ = 0;
                bc_ = 0;
                crc_ = 0;
                isize_ = 0;
                mstart_ = op_;
                crc_from_ = op_;
                st_ = S_BLOCK;
                return 0;
            }
            case S_BLOCK: {
                if (!fill(1024)) return 1;   // a dynamic header is at most ~ 14 + 19*3 + 316*7 bits
                refill();
                last_block_ = (bb_ & 1u) != 0;
                const uint32_t type = (uint32_t)(bb_ >> 1) & 3u;
                bb_ >>= 3;
                bc_ -= 3;
                if (type == 0) {
                    unread_bits();
                    if (avail() < 4) return 4;
                    const uint8_t* p = in_.data() + ip_;
                    const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8), nlen = (uint32_t)p[2] | ((uint32_t)p[3] << 8);
                    if ((len ^ 0xFFFFu) != nlen) return 4;
                    ip_ += 4;
                    stored_left_ = len;
                    st_ = S_STORED;
                    return 0;
                }
                if (type == 3) return 4;
                uint8_t lens[320];
                int nlen, ndist;
                if (type == 1) {
                    nlen = 288;
                    ndist = 32;   // (3.2.6: 32 five-bit codes, of which 30 and 31 never occur - decoding one is an error)
                    for (int s = 0; s < 288; s++) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
                    for (int s = 0; s < 32; s++) lens[288 + s] = 5;
                } else {
                    nlen = (int)(bb_ & 31u) + 257;
                    ndist = (int)((bb_ >> 5) & 31u) + 1;
                    const int ncode = (int)((bb_ >> 10) & 15u) + 4;
                    bb_ >>= 14;
                    bc_ -= 14;
                    if (nlen > 286 || ndist > 30) return 4;
                    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint8_t cl[19] = {0};
                    for (int i = 0; i < ncode; i++) {
                        if (bc_ < 3) refill();
                        cl[order[i]] = (uint8_t)(bb_ & 7u);
                        bb_ >>= 3;
                        bc_ -= 3;
                    }
                    uint32_t ct[1 << 7];
                    if (!build_table(cl, 19, 2, ct, 7)) return 4;
                    int idx = 0;
                    const int total = nlen + ndist;
                    while (idx < total) {
                        refill();
                        const uint32_t e = ct[bb_ & 127u];
                        if (!e) return 4;
                        const int l = (int)((e >> 8) & 15u);
                        bb_ >>= l;
                        bc_ -= l;
                        const uint32_t sym = e >> 16;
                        if (sym < 16) { lens[idx++] = (uint8_t)sym; continue; }
                        int rep;
                        uint8_t val = 0;
                        if (sym == 16) {
                            if (idx == 0) return 4;
                            val = lens[idx - 1];
                            rep = 3 + (int)(bb_ & 3u);
                            bb_ >>= 2;
                            bc_ -= 2;
                        } else if (sym == 17) {
                            rep = 3 + (int)(bb_ & 7u);
                            bb_ >>= 3;
                            bc_ -= 3;
                        } else {
                            rep = 11 + (int)(bb_ & 127u);
                            bb_ >>= 7;
                            bc_ -= 7;
                        }
                        if (idx + rep > total) return 4;
                        while (rep--) lens[idx++] = val;
                    }
                    if (lens[256] == 0) return 4;   // no end-of-block code
                    if (overrun()) return 4;
                    // the distance lengths follow the literal / length ones directly
                    memmove(lens + 288, lens + nlen, (size_t)ndist);
                }
                if (!build_table(lens, nlen, 0, lt_, LROOT)) return 4;
                if (!build_table(lens + 288, ndist, 1, dt_, DROOT)) return 4;
                st_ = S_CODES;
                return 0;
            }
            case S_STORED: {
                if (stored_left_ == 0) { st_ = last_block_ ? S_TRAILER : S_BLOCK; return 0; }
                make_room();
                if (!fill(1)) return 1;
                if (avail() == 0) return 4;
                const size_t n = std::min<size_t>(std::min<size_t>(stored_left_, avail()), win_.size() - 512 - op_);
                memcpy(win_.data() + op_, in_.data() + ip_, n);
                op_ += n;
                ip_ += n;
                stored_left_ -= (uint32_t)n;
                return 0;
            }
            case S_CODES: return codes();
            case S_TRAILER: {
                unread_bits();
                if (!fill(8)) return 1;
                if (avail() < 8) return 4;
                flush_crc();
                const uint8_t* t = in_.data() + ip_;
                const uint32_t c = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                const uint32_t n = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                if (c != crc_ || n != isize_) return 4;
                ip_ += 8;
                st_ = S_HEADER;
                return 0;
            }
        }
        return 4;
    }

    // symbols of the current block until its end, the output buffer's end or the input's
    int codes() {
        make_room();
        if (!fill(in_cap_ / 2)) return 1;
        uint8_t* const w = win_.data();
        uint8_t* op = w + op_;
        uint8_t* const oend = w + win_.size() - 512 - 258;   // a match and the copy's overshoot fit behind it
        const uint8_t* const in = in_.data();
        // symbols may start while 16 input bytes remain in front of the padding (at the file's end the padding itself is
        // readable: then up to the last byte)
        const size_t ilimit = file_done_ ? in_len_ + 8 : (in_len_ >= 16 ? in_len_ - 16 : 0);
        size_t ip = ip_;
        uint64_t bb = bb_;
        int bc = bc_;
        const uint8_t* const mstart = w + mstart_;
        int rc = 0;
        bool done = false;
        const uint32_t lmask = (1u << LROOT) - 1u, dmask = (1u << DROOT) - 1u;
        while (op < oend && ip < ilimit) {
            {   // refill
                uint64_t x;
                memcpy(&x, in + ip, 8);
                bb |= x << bc;
                ip += (size_t)((63 - bc) >> 3);
                bc |= 56;
            }
            uint32_t e = lt_[bb & lmask];
            if (e & E_SUB) {
                bb >>= LROOT;
                bc -= LROOT;
                e = lt_[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
            }
            if (e & E_LIT) {
                bb >>= (e & 0xFFu);
                bc -= (int)(e & 0xFFu);
                *op++ = (uint8_t)(e >> 16);
                // a second and a third literal from the same refill (<= 15 bits each: 56 - 26 - 15 - 15 >= 0)
                e = lt_[bb & lmask];
                if ((e & (E_LIT | E_SUB)) == E_LIT) {
                    bb >>= (e & 0xFFu);
                    bc -= (int)(e & 0xFFu);
                    *op++ = (uint8_t)(e >> 16);
                    e = lt_[bb & lmask];
                    if ((e & (E_LIT | E_SUB)) == E_LIT) {
                        bb >>= (e & 0xFFu);
                        bc -= (int)(e & 0xFFu);
                        *op++ = (uint8_t)(e >> 16);
                    }
                }
                continue;
            }
            if (e & E_EOB) {
                bb >>= (e & 0xFFu);
                bc -= (int)(e & 0xFFu);
                done = true;
                break;
            }
            if (!(e & E_LEN)) { rc = 4; break; }
            {
                const uint64_t saved = bb;
                const uint32_t cl = (e >> 8) & 15u, tot = e & 0xFFu;
                bb >>= tot;
                bc -= (int)tot;
                uint32_t len = (e >> 16) + ((uint32_t)(saved >> cl) & ((1u << (tot - cl)) - 1u));
                uint32_t d = dt_[bb & dmask];
                if (d & E_SUB) {
                    bb >>= DROOT;
                    bc -= DROOT;
                    d = dt_[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))];
                }
                if (!(d & E_LEN)) { rc = 4; break; }
                const uint64_t saved2 = bb;
                const uint32_t dcl = (d >> 8) & 15u, dtot = d & 0xFFu;
                bb >>= dtot;
                bc -= (int)dtot;
                const uint32_t dist = (d >> 16) + ((uint32_t)(saved2 >> dcl) & ((1u << (dtot - dcl)) - 1u));
                if ((size_t)(op - mstart) < dist) { rc = 4; break; }   // (op - w >= WIN >= dist always: the subtraction below stays inside the buffer)
                const uint8_t* src = op - dist;
                uint8_t* const stop = op + len;
                if (dist >= 8) {
                    do {
                        uint64_t x;
                        memcpy(&x, src, 8);
                        memcpy(op, &x, 8);
                        src += 8;
                        op += 8;
                    } while (op < stop);
                } else if (dist == 1) {
                    const uint64_t x = 0x0101010101010101ull * (uint64_t)*src;
                    do {
                        memcpy(op, &x, 8);
                        op += 8;
                    } while (op < stop);
                } else {
                    do { *op++ = *src++; } while (op < stop);
                }
                op = stop;
            }
        }
        if (bc < 0) rc = 4;
        ip_ = ip;
        bb_ = bb;
        bc_ = bc;
        op_ = (size_t)(op - w);
        if (rc) return rc;
        if (overrun()) return 4;
        if (done) { st_ = last_block_ ? S_TRAILER : S_BLOCK; return 0; }
        if (ip >= ilimit && file_done_ && !done) return 4;   // the file ended inside a block
        return 0;
    }
};

}  // namespace fqgz
