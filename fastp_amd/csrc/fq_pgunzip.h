// fq_pgunzip.h - one gzip stream inflated by SEVERAL host threads (the file loop's ".gz" inputs that are not bgzip-written).
//
// Reference code this stands in for: FastqReader::readToBufIgzip (src/fastqreader.cpp:88-149), ISA-L's igzip on ONE reader
// thread per file.  fq_gunzip.h is the same thing done faster (0.7 GB/s of text per thread); a paired run on two ".fastq.gz"
// files - what sequencers deliver - is then bounded by two host threads at about 4 M reads/s each while the device side
// runs at ten times that.  A deflate stream has no marked block boundaries and every block may point 32 KiB back into text
// the decoder has not seen, so it is cut up the way pugz / rapidgzip do it:
//
//   1. the compressed bytes of a batch (threads x chunk) are read into memory; thread k looks for the first bit position at or
//      behind k * chunk where a dynamic-Huffman block header parses (BFINAL = 0, BTYPE = 2, HLIT / HDIST in range, a complete
//      code-length code, valid repeats, an end-of-block code, complete literal and distance codes): find_block
//   2. thread k decodes from its position to thread k+1's with the 32 KiB window in front of it UNKNOWN: symbols are 16 bits
//      wide, the window is preset with markers 0x8000 | i, a match that reaches into it copies markers like any other symbol
//      (decode_blocks<uint16_t>; chunk 0 knows its window and decodes bytes, decode_blocks<uint8_t>)
//   3. in sequence, cheap: chunk k must have ended exactly on the bit chunk k+1 started at (otherwise the candidate was no block
//      start: the batch ends in front of it and the next batch begins there - the result never depends on a guess); the last
//      32 KiB of chunk k, resolved against chunk k-1's window, are chunk k+1's window
//   4. every thread replaces the markers of its chunk (narrow: in place, 16 symbols at a time where there is none) and takes the
//      CRC-32 of its members' pieces; the pieces' CRCs are combined (crc32_combine: x^(8 n) mod P) and compared with the trailers
//
// Three batches are in flight: steps 1 - 3 of one (stage A), step 4 and the trailers of the one before it (stage B, a third of the
// threads), the caller taking the text of the one before that.  Contract as fq_gunzip.h's: RFC 1951 / 1952,
// members one behind the other, CRC-32 and ISIZE of every member checked, a damaged or cut-off stream is an error.
// tests/test_gunzip.py runs both inflaters on the same streams, with chunk sizes from a few hundred bytes upwards so that small
// files cross many chunk boundaries.
#pragma once
#include <time.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <memory>
#include <exception>
#include <mutex>
#include <thread>

#include "fq_gunzip.h"

namespace fqgz {

enum { PG_WIN = 32768, PG_PAD = 1024 };

// ---- CRC-32 of a concatenation ------------------------------------------------------------------------------------------------
// polynomials over GF(2) modulo the CRC polynomial, reflected (bit 31 = x^0)
inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
// crc(A ++ B) from crc(A), crc(B) and |B| (zlib's crc32() convention for the values)
inline uint32_t crc_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    static const struct Pow {
        uint32_t t[64];   // x^(2^k)
        Pow() {
            uint32_t p = 1u << 30;   // x^1
            t[0] = p;
            for (int k = 1; k < 64; k++) t[k] = p = crc_mulmod(p, p);
        }
    } pw;
    if (len_b == 0) return crc_a;
    uint32_t p = 1u << 31;   // x^0
    uint64_t n = len_b;
    for (int k = 3; n; n >>= 1, k++)   // x^(8 len_b)
        if (n & 1u) p = crc_mulmod(pw.t[k & 63], p);
    return crc_mulmod(p, crc_a) ^ crc_b;
}

// ---- bits of a buffer in memory (padded: PG_PAD readable zero bytes behind `len`) --------------------------------------------------------
struct BitIn {
    const uint8_t* in = nullptr;
    size_t len = 0, ip = 0;
    uint64_t bb = 0;
    int bc = 0;
    void seek(uint64_t bit) {
        ip = (size_t)(bit >> 3);
        bb = 0;
        bc = 0;
        refill();
        const int drop = (int)(bit & 7u);
        bb >>= drop;
        bc -= drop;
    }
    inline void refill() {
        uint64_t w;
        memcpy(&w, in + ip, 8);
        bb |= w << bc;
        ip += (size_t)((63 - bc) >> 3);
        bc |= 56;
    }
    uint64_t pos() const { return 8 * (uint64_t)ip - (uint64_t)bc; }
    bool overrun() const { return pos() > 8 * (uint64_t)len; }
};

// the code tables of a dynamic block whose 14 header bits come next; false: not a valid header
inline bool dynamic_tables(BitIn& b, uint32_t* lt, uint32_t* dt) {
    b.refill();
    const int nlen = (int)(b.bb & 31u) + 257, ndist = (int)((b.bb >> 5) & 31u) + 1, ncode = (int)((b.bb >> 10) & 15u) + 4;
    b.bb >>= 14;
    b.bc -= 14;
    if (nlen > 286 || ndist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < ncode; i++) {
        if (b.bc < 3) b.refill();
        cl[order[i]] = (uint8_t)(b.bb & 7u);
        b.bb >>= 3;
        b.bc -= 3;
    }
    uint32_t ct[1 << 7];
    if (!build_table(cl, 19, 2, ct, 7)) return false;
    uint8_t lens[320 + 32];
    int idx = 0;
    const int total = nlen + ndist;
    while (idx < total) {
        b.refill();
        const uint32_t e = ct[b.bb & 127u];
        if (!e) return false;
        const int l = (int)((e >> 8) & 15u);
        b.bb >>= l;
        b.bc -= l;
        const uint32_t sym = e >> 16;
        if (sym < 16) { lens[idx++] = (uint8_t)sym; continue; }
        int rep;
        uint8_t val = 0;
        if (sym == 16) {
            if (idx == 0) return false;
            val = lens[idx - 1];
            rep = 3 + (int)(b.bb & 3u);
            b.bb >>= 2;
            b.bc -= 2;
        } else if (sym == 17) {
            rep = 3 + (int)(b.bb & 7u);
            b.bb >>= 3;
            b.bc -= 3;
        } else {
            rep = 11 + (int)(b.bb & 127u);
            b.bb >>= 7;
            b.bc -= 7;
        }
        if (idx + rep > total) return false;
        while (rep--) lens[idx++] = val;
    }
    if (lens[256] == 0 || b.overrun()) return false;
    uint8_t dl[32];
    memcpy(dl, lens + nlen, (size_t)ndist);
    return build_table(lens, nlen, 0, lt, LROOT) && build_table(dl, ndist, 1, dt, DROOT);
}

inline void fixed_tables(uint32_t* lt, uint32_t* dt) {
    uint8_t lens[288], dl[32];
    for (int s = 0; s < 288; s++) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
    for (int s = 0; s < 32; s++) dl[s] = 5;
    build_table(lens, 288, 0, lt, LROOT);
    build_table(dl, 32, 1, dt, DROOT);
}

// the first bit position in [from, to) where a non-final dynamic block header parses; ~0 if there is none.
// 1 position in 8 passes the three type bits, 1 in ~10 of those the complete code-length code, a handful per megabyte reach the
// full parse, and what survives that and is not a block start is caught by step 3 above.
inline uint64_t find_block(const uint8_t* in, size_t len, uint64_t from, uint64_t to, uint32_t* lt, uint32_t* dt) {
    const uint64_t last = 8 * (uint64_t)len;
    for (uint64_t p = from; p < to && p + 17 + 12 <= last; p++) {
        uint64_t v;
        memcpy(&v, in + (p >> 3), 8);
        v >>= (p & 7u);                                    // >= 57 bits
        if ((v & 7u) != 4u) continue;                      // BFINAL 0, BTYPE 10
        if (((v >> 3) & 31u) > 29u || ((v >> 8) & 31u) > 29u) continue;
        const int ncode = (int)((v >> 13) & 15u) + 4;
        uint64_t c = v >> 17;                              // 40 bits = 13 code lengths; the rest from a second load
        uint32_t kraft = 0;
        for (int i = 0; i < ncode; i++) {
            if (i == 13) {
                memcpy(&c, in + ((p + 17 + 39) >> 3), 8);
                c >>= ((p + 17 + 39) & 7u);
            }
            const uint32_t l = (uint32_t)(c & 7u);
            c >>= 3;
            if (l) kraft += 128u >> l;
        }
        if (kraft != 128u) continue;                       // the code-length code must be complete (build_table, kind 2)
        BitIn b;
        b.in = in;
        b.len = len;
        b.seek(p + 3);
        if (dynamic_tables(b, lt, dt)) return p;
    }
    return ~0ull;
}

// ---- blocks from a bit position on, in memory -----------------------------------------------------------------------------------
struct MemberEnd {
    size_t out_at;           // the member's text ends in front of this symbol of the chunk
    uint32_t crc, isize;     // its trailer
};
enum { PG_STOP = 0, PG_NEED_INPUT = 1, PG_OUT_FULL = 2, PG_EOF = 3, PG_DAMAGED = 4 };

struct ChunkState {
    uint64_t bit = 0;        // where to go on: a member header (at_header, byte aligned) or a block header
    bool at_header = false;
    size_t op = PG_WIN;      // symbols out[PG_WIN, op) are text
    size_t mstart = 0;       // a distance must not reach below this symbol (the current member's start, as far as it is known)
    std::vector<MemberEnd> ends;
};

// T = uint8_t: the window out[0, PG_WIN) holds text.  T = uint16_t: it holds the markers 0x8000 | i.  Decodes block after
// block from s.bit; s is left at the last block (or member) boundary that was reached completely:
//   PG_STOP        a block header at a position >= stop_bit has been reached (the caller compares it with what it expected)
//   PG_NEED_INPUT  the input ends inside the next block and is not the end of the file
//   PG_OUT_FULL    the next block does not fit into `out`
//   PG_EOF         the file ends behind a member
//   PG_DAMAGED     RFC 1951 / 1952 violated at s.bit or behind it
template <class T>
int decode_blocks(const uint8_t* in, size_t len, bool final_input, uint64_t stop_bit, T* out, size_t ocap, ChunkState& s, uint32_t* lt, uint32_t* dt) {
    const uint32_t lmask = (1u << LROOT) - 1u, dmask = (1u << DROOT) - 1u;
    // (a dynamic header is at most 14 + 57 + 316 * 14 bits: the buffer's padding - PG_PAD zero bytes - covers a parse that runs
    // off the end of the data; b.overrun() then tells)
    for (;;) {
        if (s.at_header) {
            const size_t at0 = (size_t)(s.bit >> 3);
            if (at0 >= len) return final_input ? (at0 == len ? PG_EOF : PG_DAMAGED) : PG_NEED_INPUT;
            const uint8_t* p = in + at0;
            const size_t n = len - at0;
            const int cut = final_input ? PG_DAMAGED : PG_NEED_INPUT;   // what running out of bytes means here
            if (n < 18) return cut;
            if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return PG_DAMAGED;
            const int flg = p[3];
            size_t at = 10;
            if (flg & 4) {
                if (at + 2 > n) return cut;
                at += 2 + ((size_t)p[at] | ((size_t)p[at + 1] << 8));
            }
            for (int pass = 0; pass < 2; pass++)
                if (flg & (pass == 0 ? 8 : 16)) {
                    while (at < n && p[at]) at++;
                    at++;
                }
            if (flg & 2) at += 2;
            if (at > n) return cut;
            s.bit = 8 * (uint64_t)(at0 + at);
            s.at_header = false;
            s.mstart = s.op;
            continue;
        }
        if (s.bit >= stop_bit) return PG_STOP;
        if (s.bit + 3 > 8 * (uint64_t)len) return final_input ? PG_DAMAGED : PG_NEED_INPUT;
        BitIn b;
        b.in = in;
        b.len = len;
        b.seek(s.bit);
        const bool last_block = (b.bb & 1u) != 0;
        const uint32_t type = (uint32_t)(b.bb >> 1) & 3u;
        b.bb >>= 3;
        b.bc -= 3;
        size_t op = s.op;
        uint64_t end_bit;
        if (type == 0) {
            const size_t at = (size_t)((b.pos() + 7) >> 3);
            if (at + 4 > len) return final_input ? PG_DAMAGED : PG_NEED_INPUT;
            const uint32_t n = (uint32_t)in[at] | ((uint32_t)in[at + 1] << 8), nn = (uint32_t)in[at + 2] | ((uint32_t)in[at + 3] << 8);
            if ((n ^ 0xFFFFu) != nn) return PG_DAMAGED;
            if (at + 4 + n > len) return final_input ? PG_DAMAGED : PG_NEED_INPUT;
            if (op + n + 16 > ocap) return PG_OUT_FULL;
            for (uint32_t i = 0; i < n; i++) out[op + i] = (T)in[at + 4 + i];
            op += n;
            end_bit = 8 * (uint64_t)(at + 4 + n);
        } else if (type == 3) {
            return PG_DAMAGED;
        } else {
            if (type == 1) fixed_tables(lt, dt);
            else if (!dynamic_tables(b, lt, dt)) return (b.overrun() && !final_input) ? PG_NEED_INPUT : PG_DAMAGED;
            T* o = out + op;
            T* const oend = out + ocap - 258 - 16;
            const T* const mstart = out + s.mstart;
            const size_t ilimit = final_input ? len + 8 : (len >= 16 ? len - 16 : 0);
            size_t ip = b.ip;
            uint64_t bb = b.bb;
            int bc = b.bc;
            bool done = false;
            while (o < oend && ip < ilimit) {
                {
                    uint64_t x;
                    memcpy(&x, in + ip, 8);
                    bb |= x << bc;
                    ip += (size_t)((63 - bc) >> 3);
                    bc |= 56;
                }
                uint32_t e = lt[bb & lmask];
                if (e & E_SUB) {
                    bb >>= LROOT;
                    bc -= LROOT;
                    e = lt[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 8) & 15u)) - 1u))];
                }
                if (e & E_LIT) {
                    bb >>= (e & 0xFFu);
                    bc -= (int)(e & 0xFFu);
                    *o++ = (T)(e >> 16);
                    e = lt[bb & lmask];
                    if ((e & (E_LIT | E_SUB)) == E_LIT) {
                        bb >>= (e & 0xFFu);
                        bc -= (int)(e & 0xFFu);
                        *o++ = (T)(e >> 16);
                        e = lt[bb & lmask];
                        if ((e & (E_LIT | E_SUB)) == E_LIT) {
                            bb >>= (e & 0xFFu);
                            bc -= (int)(e & 0xFFu);
                            *o++ = (T)(e >> 16);
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
                if (!(e & E_LEN)) return PG_DAMAGED;
                const uint64_t saved = bb;
                const uint32_t cl = (e >> 8) & 15u, tot = e & 0xFFu;
                bb >>= tot;
                bc -= (int)tot;
                const uint32_t mlen = (e >> 16) + ((uint32_t)(saved >> cl) & ((1u << (tot - cl)) - 1u));
                uint32_t d = dt[bb & dmask];
                if (d & E_SUB) {
                    bb >>= DROOT;
                    bc -= DROOT;
                    d = dt[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 8) & 15u)) - 1u))];
                }
                if (!(d & E_LEN)) return PG_DAMAGED;
                const uint64_t saved2 = bb;
                const uint32_t dcl = (d >> 8) & 15u, dtot = d & 0xFFu;
                bb >>= dtot;
                bc -= (int)dtot;
                const uint32_t dist = (d >> 16) + ((uint32_t)(saved2 >> dcl) & ((1u << (dtot - dcl)) - 1u));
                if ((size_t)(o - mstart) < dist) return PG_DAMAGED;   // (o - out >= PG_WIN >= dist: the copy stays inside the buffer)
                const T* src = o - dist;
                T* const stop = o + mlen;
                constexpr uint32_t per8 = 8 / sizeof(T);
                if (dist >= per8) {
                    do {
                        uint64_t x;
                        memcpy(&x, src, 8);
                        memcpy(o, &x, 8);
                        src += per8;
                        o += per8;
                    } while (o < stop);
                } else {
                    // a period shorter than a word (runs of one quality character: distance 1): symbol by symbol until a
                    // multiple of the period that fills a word has been written, then word-wise from that far back
                    uint32_t d2 = dist;
                    while (d2 < per8) d2 += dist;
                    T* const lim = o + d2 < stop ? o + d2 : stop;
                    do { *o++ = *src++; } while (o < lim);
                    while (o < stop) {
                        uint64_t x;
                        memcpy(&x, o - d2, 8);
                        memcpy(o, &x, 8);
                        o += per8;
                    }
                }
                o = stop;
            }
            if (bc < 0) return final_input ? PG_DAMAGED : PG_NEED_INPUT;
            end_bit = 8 * (uint64_t)ip - (uint64_t)bc;
            if (end_bit > 8 * (uint64_t)len) return final_input ? PG_DAMAGED : PG_NEED_INPUT;
            if (!done) {
                if (o >= oend) return PG_OUT_FULL;
                return final_input ? PG_DAMAGED : PG_NEED_INPUT;   // the input ended inside the block
            }
            op = (size_t)(o - out);
        }
        if (last_block) {
            const size_t at = (size_t)((end_bit + 7) >> 3);
            if (at + 8 > len) return final_input ? PG_DAMAGED : PG_NEED_INPUT;
            const uint8_t* t = in + at;
            MemberEnd me;
            me.out_at = op;
            me.crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            me.isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            s.ends.push_back(me);
            s.bit = 8 * (uint64_t)(at + 8);
            s.at_header = true;
        } else {
            s.bit = end_bit;
        }
        s.op = op;
    }
}

// 16-bit symbols -> bytes: dst[i] = src[i] < 256 ? src[i] : window[src[i] & 0x7FFF].  dst may be the front of src's own buffer.
// Returns the lowest window position a marker pointed at (PG_WIN: there was none).  FASTQ text keeps its markers to the end
// of a chunk (a record's name is a match into the record before it, and so on back into the window; bases match bases
// anywhere), so every second group of 16 symbols holds one: the symbols go through a 64 K table - no branch to mispredict -
// and only groups without a marker are packed 16 at a time.  Markers are the negative 16-bit numbers: the lowest position is
// the signed minimum.
inline uint32_t narrow(uint8_t* dst, const uint16_t* src, size_t n, const uint8_t* window, uint8_t* lut /* 65536 bytes of scratch */) {
    for (int i = 0; i < 256; i++) lut[i] = (uint8_t)i;
    memcpy(lut + 0x8000, window, PG_WIN);
    size_t i = 0;
    int low = 0;
#if defined(__x86_64__)
    __m128i lo8 = _mm_setzero_si128();
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 8));
        const __m128i m = _mm_min_epi16(a, b);
        lo8 = _mm_min_epi16(lo8, m);
        if (_mm_movemask_epi8(m) & 0xAAAA) {   // a sign bit: a marker among the sixteen
            uint16_t t[16];
            _mm_storeu_si128((__m128i*)t, a);
            _mm_storeu_si128((__m128i*)(t + 8), b);
            uint8_t r[16];
            for (int k = 0; k < 16; k++) r[k] = lut[t[k]];
            memcpy(dst + i, r, 16);
        } else {
            _mm_storeu_si128((__m128i*)(dst + i), _mm_packus_epi16(a, b));
        }
    }
    {
        int16_t l[8];
        _mm_storeu_si128((__m128i*)l, lo8);
        for (int k = 0; k < 8; k++) low = std::min(low, (int)l[k]);
    }
#endif
    for (; i < n; i++) {
        uint16_t t;
        memcpy(&t, src + i, 2);
        low = std::min(low, (int)(int16_t)t);
        dst[i] = lut[t];
    }
    return low < 0 ? (uint32_t)(low + 32768) : (uint32_t)PG_WIN;
}

// ---- the reader -------------------------------------------------------------------------------------------------------------------
class ParallelGunzip {
  public:
    int fd = -1;
    int64_t fpos = 0, fsize = 0;   // file bytes consumed by the text handed out so far / the file's size (a regular file: pread)
    bool at_eof = false;           // every member has been delivered and the file has ended
    int threads = 4;
    size_t chunk = 2 << 20;        // compressed bytes per thread and batch
    // statistics (tests, FASTP_GPU_VERBOSE)
    int64_t batches = 0, chunks_used = 0, chunks_dropped = 0, marker_faults = 0;
    double t_phase[6] = {0, 0, 0, 0, 0, 0};   // seconds in: -, -, reading + finding + decoding, windows, resolving + CRC, trailers

    ~ParallelGunzip() {
        stop_ = true;
        free_.close();
        to_b_.close();
        to_c_.close();
        if (thread_a_.joinable()) thread_a_.join();
        if (thread_b_.joinable()) thread_b_.join();
    }

    // up to `want` bytes of text to dst; fewer only at the end of the file; < 0: error (*err: 1 reading failed, 4 damaged stream)
    int64_t read(uint8_t* dst, int64_t want, int* err) {
        int64_t made = 0;
        while (made < want) {
            if (cur_ && cur_->piece < cur_->pieces.size()) {
                Piece& p = cur_->pieces[cur_->piece];
                const int64_t n = std::min<int64_t>(want - made, (int64_t)(p.len - p.done));
                memcpy(dst + made, p.text + p.done, (size_t)n);
                p.done += (size_t)n;
                made += n;
                if (p.done == p.len) cur_->piece++;
                continue;
            }
            if (cur_) {   // the batch has been handed out: its end is what the caller has consumed
                fpos = cur_->end_byte;
                if (cur_->rc) { *err = cur_->rc; return -1; }
                if (cur_->eof) { at_eof = true; break; }
                free_.push(cur_idx_);
                cur_ = nullptr;
            }
            if (!started_) {
                if (fsize == 0) { *err = 4; return -1; }   // no member header at all: the reference's "invalid gzip header" (fastqreader.cpp:193-196)
                started_ = true;
                for (int i = 0; i < N_SLOTS; i++) free_.push(i);
                thread_a_ = std::thread([this] { main_a(); });
                thread_b_ = std::thread([this] { main_b(); });
            }
            cur_idx_ = to_c_.pop();
            if (cur_idx_ < 0) { *err = 4; return -1; }   // (only after the destructor has closed the queues)
            cur_ = &slot_[cur_idx_];
        }
        return made;
    }

  private:
    struct Piece {
        const uint8_t* text;
        size_t len, done;
    };
    struct Chunk {
        std::unique_ptr<uint16_t[]> sym;   // [window | symbols]; chunk 0: the same bytes used as uint8_t [window | text]
        size_t sym_cap = 0;            // (not value-initialised: pages are touched when symbols land on them)
        ChunkState st;
        uint64_t start = ~0ull;        // bit position of its first block (~0: no block start found in its range)
        int rc = PG_STOP;
        std::vector<uint32_t> seg_crc; // CRC-32 of the text between member ends (one more than st.ends: the open tail)
        uint32_t low_marker = PG_WIN;  // the furthest its text reaches back into the window in front of it (PG_WIN: not at all)
        std::vector<uint32_t> lt, dt;
        std::vector<uint8_t> lut;      // narrow()'s table
    };
    struct Batch {
        std::vector<uint8_t> in;
        std::vector<Chunk> ck;
        // stage A (read, find, decode, the chain) leaves:
        std::vector<int> used;                       // the chunks that count, in order
        std::vector<std::vector<uint8_t>> win;       // win[i] = the 32 KiB in front of used[i]
        std::vector<uint64_t> since_at;              // text bytes of the open member in front of used[i]
        int a_rc = 0;
        bool a_eof = false;
        // stage B (markers -> bytes, CRC-32, trailers) leaves:
        std::vector<Piece> pieces;
        size_t piece = 0;
        int rc = 0;
        bool eof = false;
        int64_t end_byte = 0;
    };
    // a queue of slot numbers between the stages; pop() = -1 once it is closed and empty
    struct Queue {
        std::mutex mu;
        std::condition_variable cv;
        std::deque<int> q;
        bool closed = false;
        void push(int v) {
            { std::lock_guard<std::mutex> lk(mu); q.push_back(v); }
            cv.notify_one();
        }
        int pop() {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return closed || !q.empty(); });
            if (q.empty()) return -1;
            const int v = q.front();
            q.pop_front();
            return v;
        }
        void close() {
            { std::lock_guard<std::mutex> lk(mu); closed = true; }
            cv.notify_all();
        }
    };
    // Three batches in flight: stage A of batch b+2 (its threads decode) beside stage B of batch b+1 (its threads resolve) beside the
    // caller copying batch b out.  Stage A needs of its predecessor only what ITS stage A left (position, window, member length),
    // stage B only what its predecessor's stage B left (the open member's CRC-32 and length).
    enum { N_SLOTS = 3 };
    Batch slot_[N_SLOTS];
    Batch* cur_ = nullptr;
    int cur_idx_ = -1;
    bool started_ = false;
    std::atomic<bool> stop_{false};
    std::thread thread_a_, thread_b_;
    Queue free_, to_b_, to_c_;
    // where the stream stands behind the last batch that was produced
    uint64_t bit_ = 0;             // absolute bit position in the file
    bool at_header_ = true;
    uint8_t window_[PG_WIN] = {0};
    uint64_t since_member_ = 0;    // text bytes of the open member so far
    // stage B's own
    uint32_t crc_ = 0;             // of the open member's text so far
    uint64_t isize_ = 0;
    static const Crc32& crc_tab() {
        static const Crc32 c;
        return c;
    }

    void main_a() {
        for (;;) {
            const int i = free_.pop();
            if (i < 0 || stop_) return;
            Batch& B = slot_[i];
            try {
                stage_a(B);
            } catch (...) {   // (memory for a chunk's symbols, a thread that cannot be started): an error of the run, not std::terminate
                B.used.clear();
                B.a_rc = 1;
            }
            const bool last = B.a_rc != 0 || B.a_eof;
            to_b_.push(i);
            if (last) return;
        }
    }
    void main_b() {
        for (;;) {
            const int i = to_b_.pop();
            if (i < 0 || stop_) return;
            Batch& B = slot_[i];
            try {
                stage_b(B);
            } catch (...) {
                B.pieces.clear();
                B.rc = 1;
            }
            const bool last = B.rc != 0 || B.eof;
            to_c_.push(i);
            if (last) return;
        }
    }

    static double now() {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
    }
    template <class F>
    static void fan_out(int n, F f) {
        if (n <= 0) return;
        std::vector<std::thread> th;
        th.reserve((size_t)n);
        // an exception of any f(k) (bad_alloc) leaves through the caller's thread once every thread has been joined
        std::mutex em;
        std::exception_ptr first;
        auto guarded = [&f, &em, &first](int k) {
            try {
                f(k);
            } catch (...) {
                std::lock_guard<std::mutex> g(em);
                if (!first) first = std::current_exception();
            }
        };
        for (int k = 1; k < n; k++) th.emplace_back([&guarded, k] { guarded(k); });   // (std::system_error here ends the process as it would the reference's own thread starts)
        guarded(0);
        for (auto& t : th) t.join();
        if (first) std::rethrow_exception(first);
    }

    void stage_a(Batch& B) {
        B.a_rc = 0;
        B.a_eof = false;
        for (;;) {
            const int rc = stage_a_once(B);
            if (rc != -2) { B.a_rc = rc; return; }
            // not one block of the first chunk fits: larger chunks (a block of gigabytes is no FASTQ file's)
            if (chunk >= ((size_t)1 << 30)) { B.a_rc = 4; B.used.clear(); return; }
            chunk *= 4;
        }
    }

    // 0 fine, 1 / 4 errors, -2: no progress with this chunk size
    int stage_a_once(Batch& B) {
        B.used.clear();
        B.win.clear();
        B.since_at.clear();
        const int T = std::max(1, threads);
        const int64_t base = (int64_t)(bit_ >> 3);          // file offset of B.in[0]
        const int64_t have = std::min<int64_t>(fsize - base, (int64_t)T * (int64_t)chunk);
        if (have < 0) return 4;   // (B.used is empty: stage B hands out nothing)
        const size_t len = (size_t)have;
        const bool final_input = base + have >= fsize;
        B.in.resize(len + PG_PAD);
        memset(B.in.data() + len, 0, PG_PAD);
        uint8_t* const in = B.in.data();
        if ((int)B.ck.size() != T) B.ck.resize((size_t)T);
        // output room per chunk: 8 x the compressed bytes (FASTQ: 3.5 - 5 x) and never less than two maximal blocks
        const size_t ocap = PG_WIN + std::max<size_t>(8 * chunk, (size_t)1 << 17) + 1024;
        std::atomic<int> io_err{0};
        double t0 = now(), t1;
        auto lap = [&](int k) { t1 = now(); t_phase[k] += t1 - t0; t0 = t1; };
        // ---- 1 + 2: one thread per chunk: read the slice, find the block start, decode up to the next chunk that has a start ----
        // (no barrier between the steps: a thread waits only for what it is about to touch - the slices behind its own before it
        // parses anything, the neighbours' block starts when its own decode has reached their range)
        const uint64_t bit0 = bit_ - 8 * (uint64_t)base;     // positions below are relative to B.in
        std::unique_ptr<std::atomic<int>[]> slice_read(new std::atomic<int>[(size_t)T + 1]), start_known(new std::atomic<int>[(size_t)T + 1]);
        for (int k = 0; k <= T; k++) { slice_read[k] = k == T; start_known[k] = k == T; }
        auto wait_for = [](std::atomic<int>& f) { while (!f.load(std::memory_order_acquire)) std::this_thread::yield(); };
        fan_out(T, [&](int k) { try {
            Chunk& c = B.ck[(size_t)k];
            c.start = ~0ull;
            c.rc = PG_STOP;
            {
                const int64_t a = std::min<int64_t>(have, (int64_t)k * (int64_t)chunk), e = std::min<int64_t>(have, a + (int64_t)chunk);
                int64_t got = 0;
                while (got < e - a) {
                    const ssize_t r = pread(fd, in + a + got, (size_t)(e - a - got), (off_t)(base + a + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { io_err = 1; break; }
                    got += r;
                }
                slice_read[k].store(1, std::memory_order_release);
            }
            const size_t want = (k ? ocap : ocap / 2) + 64;
            if (c.sym_cap < want) { c.sym.reset(new uint16_t[want]); c.sym_cap = want; }
            if (c.lt.empty()) { c.lt.resize(LTAB); c.dt.resize(DTAB); }
            for (int j = k + 1; j < T; j++) wait_for(slice_read[j]);   // a header parse or a block may run through any number of (small) slices
            if (!io_err) {
                if (k == 0) {
                    c.start = bit0;
                } else {
                    const uint64_t from = 8 * (uint64_t)k * chunk, to = std::min<uint64_t>(8 * (uint64_t)len, from + 8 * (uint64_t)chunk);
                    if (from < to) c.start = find_block(in, len, from, to, c.lt.data(), c.dt.data());
                }
            }
            start_known[k].store(1, std::memory_order_release);
            if (io_err || c.start == ~0ull) return;
            c.st = ChunkState();
            c.st.bit = c.start;
            uint8_t* const o8 = (uint8_t*)c.sym.get();
            uint16_t* const o16 = c.sym.get();
            if (k == 0) {
                memcpy(o8, window_, PG_WIN);
                c.st.at_header = at_header_;
                c.st.mstart = PG_WIN - (size_t)std::min<uint64_t>(PG_WIN, since_member_);
            } else {
                for (uint32_t i = 0; i < PG_WIN; i++) o16[i] = (uint16_t)(0x8000u | i);
            }
            auto run = [&](uint64_t stop) {
                return k == 0 ? decode_blocks<uint8_t>(in, len, final_input, stop, o8, ocap, c.st, c.lt.data(), c.dt.data())
                              : decode_blocks<uint16_t>(in, len, final_input, stop, o16, ocap, c.st, c.lt.data(), c.dt.data());
            };
            // first to the first block boundary in the next chunk's range (its start cannot lie in front of that one) ...
            c.rc = run(std::min<uint64_t>(8 * (uint64_t)(k + 1) * chunk, 8 * (uint64_t)len + 64));
            if (c.rc != PG_STOP) return;
            // ... then to the start of the next chunk that has one (none: as far as the input goes)
            uint64_t stop = ~0ull;
            for (int j = k + 1; j < T && stop == ~0ull; j++) {
                wait_for(start_known[j]);
                stop = B.ck[(size_t)j].start;
            }
            if (io_err) return;
            c.rc = run(stop);
        } catch (...) {   // no memory for this chunk's symbols: an error of the run; nobody may be left waiting for this thread's flags
            io_err = 1;
            slice_read[k].store(1, std::memory_order_release);
            start_known[k].store(1, std::memory_order_release);
        } });
        if (io_err) return 1;
        lap(2);
        // ---- 3: the chain: which chunks count, and the window in front of each ----
        std::vector<int>& used = B.used;
        std::vector<std::vector<uint8_t>>& win = B.win;
        int result = 0;
        bool eof = false;
        {
            uint64_t expect = bit0;
            const uint64_t since0 = since_member_;
            std::vector<uint8_t> w(window_, window_ + PG_WIN), lut(65536);
            for (int k = 0; k < T; k++) {
                Chunk& c = B.ck[(size_t)k];
                if (c.start == ~0ull) continue;
                if (c.start != expect) break;                 // the chunk in front did not end here: no block start after all
                used.push_back(k);
                win.push_back(w);
                const size_t n = c.st.op - PG_WIN;
                B.since_at.push_back(since_member_);
                since_member_ = c.st.ends.empty() ? since_member_ + n : (uint64_t)(c.st.op - c.st.ends.back().out_at);
                // the window behind it = the last 32 KiB of (w ++ its text)
                std::vector<uint8_t> nw(PG_WIN);
                if (n >= PG_WIN) {
                    if (k == 0) memcpy(nw.data(), (const uint8_t*)c.sym.get() + c.st.op - PG_WIN, PG_WIN);
                    else narrow(nw.data(), c.sym.get() + c.st.op - PG_WIN, PG_WIN, w.data(), lut.data());
                } else {
                    memcpy(nw.data(), w.data() + n, PG_WIN - n);
                    if (k == 0) memcpy(nw.data() + PG_WIN - n, (const uint8_t*)c.sym.get() + PG_WIN, n);
                    else narrow(nw.data() + PG_WIN - n, c.sym.get() + PG_WIN, n, w.data(), lut.data());
                }
                w.swap(nw);
                expect = c.st.bit;
                if (c.rc == PG_DAMAGED) { result = 4; break; }
                if (c.rc == PG_EOF) { eof = true; break; }
                if (c.rc != PG_STOP) break;                   // input or room ran out in front of the next chunk: the batch ends here
            }
            chunks_used += (int64_t)used.size();
            for (int k = 0; k < T; k++) chunks_dropped += B.ck[(size_t)k].start != ~0ull && std::find(used.begin(), used.end(), k) == used.end();
            const Chunk& l = B.ck[(size_t)used.back()];
            if (!result && !eof && l.st.bit == bit0 && l.st.at_header == at_header_) { since_member_ = since0; return -2; }   // (used is never empty: chunk 0 starts at bit0)
            memcpy(window_, w.data(), PG_WIN);
        }
        lap(3);
        const Chunk& l = B.ck[(size_t)used.back()];
        bit_ = l.st.bit + 8 * (uint64_t)base;
        at_header_ = l.st.at_header;
        B.end_byte = (int64_t)((bit_ + 7) >> 3);
        B.a_eof = eof && !result;
        batches++;
        return result;
    }

    void stage_b(Batch& B) {
        B.pieces.clear();
        B.piece = 0;
        std::vector<int>& used = B.used;
        std::vector<std::vector<uint8_t>>& win = B.win;
        int result = B.a_rc;
        double t0 = now(), t1;
        auto lap = [&](int k) { t1 = now(); t_phase[k] += t1 - t0; t0 = t1; };
        // ---- 4: markers -> bytes, CRC-32 of the members' pieces ----
        // (a chunk's share of this stage is a fifth of its decode: a third of the threads keep up with stage A of the next batch)
        const int nb = std::min<int>((int)used.size(), (std::max(1, threads) + 2) / 3);
        fan_out(nb, [&](int j) {
          for (int i = j; i < (int)used.size(); i += nb) {
            Chunk& c = B.ck[(size_t)used[(size_t)i]];
            uint8_t* text = (uint8_t*)c.sym.get() + (used[(size_t)i] == 0 ? PG_WIN : 0);
            const size_t n = c.st.op - PG_WIN;
            if (c.lut.empty()) c.lut.resize(65536);
            c.low_marker = used[(size_t)i] != 0 ? narrow(text, c.sym.get() + PG_WIN, n, win[(size_t)i].data(), c.lut.data()) : (uint32_t)PG_WIN;
            c.seg_crc.clear();
            size_t at = 0;
            for (size_t m = 0; m <= c.st.ends.size(); m++) {
                const size_t e = m < c.st.ends.size() ? c.st.ends[m].out_at - PG_WIN : n;
                c.seg_crc.push_back(crc_tab().update(0, text + at, e - at));
                at = e;
            }
          }
        });
        lap(4);
        // ---- 5: trailers ----
        for (size_t i = 0; i < used.size() && result != 4; i++) {
            Chunk& c = B.ck[(size_t)used[i]];
            const uint8_t* text = (const uint8_t*)c.sym.get() + (used[i] == 0 ? PG_WIN : 0);
            const size_t n = c.st.op - PG_WIN;
            // a member is a stream of its own: text must not be fetched from in front of it.  Inside a chunk decode_blocks sees to
            // that (mstart); what a chunk took from the window in front of it is known only now
            if (c.low_marker < PG_WIN - std::min<uint64_t>(PG_WIN, B.since_at[i])) { marker_faults++; result = 4; break; }
            size_t at = 0;
            for (size_t m = 0; m <= c.st.ends.size(); m++) {
                const size_t e = m < c.st.ends.size() ? c.st.ends[m].out_at - PG_WIN : n;
                crc_ = crc_concat(crc_, c.seg_crc[m], e - at);
                isize_ += e - at;
                at = e;
                if (m < c.st.ends.size()) {
                    if (crc_ != c.st.ends[m].crc || (uint32_t)isize_ != c.st.ends[m].isize) { result = 4; break; }
                    crc_ = 0;
                    isize_ = 0;
                }
            }
            if (result == 4) break;   // the text in front of the damaged member has been handed out; this chunk's is not
            if (n) B.pieces.push_back(Piece{text, n, 0});
        }
        B.rc = result;
        B.eof = B.a_eof && !result;
        lap(5);
    }
};

}  // namespace fqgz
