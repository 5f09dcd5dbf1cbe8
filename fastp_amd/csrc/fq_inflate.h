// fq_inflate.h - DEFLATE (RFC 1951) decoding of BGZF blocks on the device.
//
// The step before FASTQ parsing for compressed input: the reference's
// BgzfMtReader (src/bgzf.h:36-239) cuts a .gz written by bgzip into its independent <= 64 KiB gzip
// members by the BSIZE field of each header and hands them to a pool of igzip workers.  Here the host
// does the same header walk (fastp_gpu_bgzf_index, a few bytes per block) and ONE LANE inflates one
// block: blocks are independent, a lane always sees its own earlier stores (LZ77 copies read back
// what the same lane wrote), and a text chunk holds thousands of blocks, so the machine fills with
// lanes rather than with cooperating wavefronts.
//
// Per-lane canonical Huffman tables (symbols in code order, the textbook decoder of RFC 1951 3.2.2; the
// per-length code counts sit in registers) live in LDS, interleaved by lane ([entry][lane], so that the
// bank of an access depends on the lane only).  51 KB of LDS per wavefront-workgroup: three of them per CU.
// Direct-lookup tables were tried and removed: a lane's speed is set by the dependent-instruction latency of a
// single wavefront per SIMD, not by the decoder's instruction count, and their LDS footprint (one workgroup
// per CU) halved the number of blocks in flight.  The code lengths of a dynamic block pass
// through a per-lane global scratch row.
#pragma once
#include "fq_intrin.h"
#include "fq_types.h"

namespace fq {

// rare paths stay out of line: the symbol loop has to fit the instruction cache
#define FQ_COLD __device__ __attribute__((noinline))

enum {
    INF_MAXBITS = 15,
    INF_MAXL = 288,   // literal/length codes
    INF_MAXD = 32,    // distance codes (30 used)
    INF_LANES = 64,   // workgroup = one wavefront
    // per-lane table entries (u16 each): cnt[16] (construction scratch, then held in registers) | lensym[288] | distsym[32]
    INF_O_CNT = 0,
    INF_O_LSYM = 16,
    INF_O_DSYM = 16 + INF_MAXL,
    INF_ENTRIES = 16 + INF_MAXL + INF_MAXD,
    INF_SCRATCH = 320,  // code lengths of a dynamic block (bytes per lane)
    INF_SBUF = 32,      // dwords of compressed stream staged in LDS per lane
};

// status per block
enum { INF_OK = 0, INF_E_HEADER = 1, INF_E_BTYPE = 2, INF_E_STORED = 3, INF_E_CODE = 4, INF_E_DIST = 5, INF_E_OVERRUN = 6,
       INF_E_ISIZE = 7, INF_E_TABLE = 8, INF_E_CRC = 9 };

struct InflateArgs {
    const u8* comp;        // compressed chunk (readable for 16 bytes past the last member)
    const u32* pay_off;    // [n] offset of the block's deflate payload in comp
    const u32* pay_len;    // [n] payload bytes
    const u32* isize;      // [n] uncompressed size from the member trailer
    const u32* crc;        // [n] CRC-32 from the member trailer
    const u64* out_off;    // [n] where the block's text goes in out
    int n;
    u8* out;
    u64 out_cap;
    u8* scratch;           // [n][INF_SCRATCH]
    u32* status;           // [n]
    u32* first_bad;        // atomic min of the failing block indices
    int check_crc;
};

// unaligned 8-byte global accesses (gfx9+ global memory handles any alignment)
FQ_DEV u64 inf_ld8(const u8* p) {
    u64 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
FQ_DEV void inf_st8(u8* p, u64 v) { __builtin_memcpy(p, &v, 8); }

// LSB-first bit reader.  The compressed stream is staged through a per-lane LDS buffer ([dword][lane], INF_SBUF
// dwords = 256 bytes per lane): one burst of independent 16-byte global loads per 256 bytes of stream instead of
// a dependent global load every few symbols.  All stream offsets are relative to `base`, the 4-byte aligned
// address at or below the payload start (`skew` = the payload's offset inside its first dword).
struct InfBits {
    const u8* base;   // aligned stream origin
    u32 skew;         // payload start - base (0..3)
    u32 end;          // payload length in bytes
    u32 limit;        // dwords of stream that may be fetched (payload + trailer, the chunk is padded beyond)
    u32 next_dw;      // stream dword index of sbuf slot 0's successor batch (= first dword NOT yet staged)
    u32 rd;           // next staged dword to consume (index into the lane's buffer)
    u64 buf;
    int cnt;
    u32 taken;        // dwords moved into buf so far since the last seek (+ seek origin) - for position accounting
    u32* sbuf;
    int lane;
};

struct InfQuad {  // four dwords at a dword-aligned (not 16-byte aligned) address
    u32 x, y, z, w;
};
// every argument by value: a state struct whose address escapes into a call would live in scratch memory
FQ_DEV void inf_stage_at(const u8* base, u32 next_dw, u32 limit, u32* sbuf, int lane) {
    const InfQuad* src = (const InfQuad*)(base + 4 * (size_t)next_dw);
    InfQuad v[INF_SBUF / 4];
#pragma unroll
    for (int j = 0; j < INF_SBUF / 4; j++) {
        const InfQuad z = {0u, 0u, 0u, 0u};
        v[j] = (next_dw + 4u * j < limit) ? src[j] : z;
    }
#pragma unroll
    for (int j = 0; j < INF_SBUF / 4; j++) {
        sbuf[(4 * j + 0) * INF_LANES + lane] = v[j].x;
        sbuf[(4 * j + 1) * INF_LANES + lane] = v[j].y;
        sbuf[(4 * j + 2) * INF_LANES + lane] = v[j].z;
        sbuf[(4 * j + 3) * INF_LANES + lane] = v[j].w;
    }
}
FQ_DEV void inf_stage(InfBits& b) {  // fetch the next INF_SBUF dwords of the lane's stream into LDS
    inf_stage_at(b.base, b.next_dw, b.limit, b.sbuf, b.lane);
    b.next_dw += INF_SBUF;
    b.rd = 0;
}
// (re)start reading at payload byte offset `pos`
FQ_DEV void inf_seek(InfBits& b, u32 pos) {
    const u32 abs = b.skew + pos;      // byte offset from base
    b.next_dw = abs >> 2;
    b.taken = abs >> 2;
    inf_stage(b);
    const u32 drop = abs & 3u;         // bytes of the first dword that precede pos
    b.buf = (u64)(b.sbuf[b.lane] >> (8 * drop));
    b.cnt = 32 - 8 * (int)drop;
    b.rd = 1;
    b.taken++;
}
FQ_DEV void inf_start(InfBits& b, const u8* in, u32 end, u32* sbuf, int lane) {
    const size_t addr = (size_t)in;
    b.base = in - (addr & 3u);
    b.skew = (u32)(addr & 3u);
    b.end = end;
    b.limit = ((b.skew + end + 3u) >> 2) + 4u;   // the payload and up to 16 bytes behind it (trailer / padding)
    b.sbuf = sbuf;
    b.lane = lane;
    inf_seek(b, 0);
}
// payload bytes consumed so far (whole bytes still in the bit buffer count as not consumed)
FQ_DEV u32 inf_pos(const InfBits& b) { return b.taken * 4u - b.skew - (u32)(b.cnt >> 3); }

FQ_DEV void inf_refill(InfBits& b) {
    if (b.cnt <= 32) {  // past-the-end dwords read as zero / padding and are never consumed by a valid stream
        if (b.rd == (u32)INF_SBUF) inf_stage(b);
        b.buf |= (u64)b.sbuf[b.rd * INF_LANES + b.lane] << b.cnt;
        b.rd++;
        b.taken++;
        b.cnt += 32;
    }
}
FQ_DEV u32 inf_bits(InfBits& b, int n) {  // n <= 16
    inf_refill(b);
    const u32 v = (u32)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// table entry e of this lane
FQ_DEV u16& inf_t(u16* tab, int e, int lane) { return tab[e * INF_LANES + lane]; }

// The per-length code counts of one Huffman code in registers (15 counts <= 288, ten bits each): the decode
// loop walks the code lengths with constant indices, so it touches no memory until the symbol itself.
struct InfCounts {
    u32 r[5];
};
FQ_DEV void inf_load_counts(InfCounts& c, u16* tab, int lane, int cnt_o) {
#pragma unroll
    for (int k = 0; k < 5; k++) c.r[k] = 0;
#pragma unroll
    for (int l = 1; l <= INF_MAXBITS; l++) c.r[(l - 1) / 3] |= (u32)inf_t(tab, cnt_o + l, lane) << (10 * ((l - 1) % 3));
}

// canonical Huffman decode (RFC 1951 3.2.2) of the code at the low end of `bits`: symbols at sym_o[...] in code
// order; returns symbol | code length << 16, or -1 for an invalid code.  Out of line, everything by value.
FQ_DEV int inf_walk(u32 bits, InfCounts c, u16* tab, int lane, int sym_o) {
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= INF_MAXBITS; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = (int)((c.r[(len - 1) / 3] >> (10 * ((len - 1) % 3))) & 0x3FFu);
        if (code - count < first) return (int)inf_t(tab, sym_o + index + (code - first), lane) | (len << 16);
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}
FQ_DEV int inf_decode(InfBits& b, const InfCounts& c, u16* tab, int lane, int sym_o) {
    inf_refill(b);
    const int r = inf_walk((u32)b.buf, c, tab, lane, sym_o);
    if (r < 0) return -1;
    const int len = r >> 16;
    b.buf >>= len;
    b.cnt -= len;
    return r & 0xFFFF;
}

// build count[] / symbol[] from n code lengths (read through `len_at`); returns false for an over-subscribed set
template <class F>
FQ_DEV bool inf_construct(u16* tab, int lane, int cnt_o, int sym_o, int n, F len_at, bool allow_incomplete) {
    for (int l = 0; l <= INF_MAXBITS; l++) inf_t(tab, cnt_o + l, lane) = 0;
    for (int s = 0; s < n; s++) inf_t(tab, cnt_o + (int)len_at(s), lane)++;
    int left = 1;
    for (int l = 1; l <= INF_MAXBITS; l++) {
        left <<= 1;
        left -= (int)inf_t(tab, cnt_o + l, lane);
        if (left < 0) return false;
    }
    // offsets of each length in the symbol table: kept in registers as a running sum while filling,
    // one pass per code length (n <= 288, 15 lengths: 4320 steps worst case, once per deflate block)
    int offs = 0;
    for (int l = 1; l <= INF_MAXBITS; l++) {
        const int c = (int)inf_t(tab, cnt_o + l, lane);
        if (!c) continue;
        int k = 0;
        for (int s = 0; s < n && k < c; s++)
            if ((int)len_at(s) == l) inf_t(tab, sym_o + offs + k++, lane) = (u16)s;
        offs += c;
    }
    const bool complete = left == 0;
    // an incomplete code is only legal for a distance code with a single symbol (RFC 1951 3.2.7)
    return complete || allow_incomplete;
}

// length / distance base values and extra bits (RFC 1951 3.2.5), computed rather than tabulated
FQ_DEV void inf_len_base(int sym, int& base, int& extra) {  // sym = 257..285
    const int i = sym - 257;
    if (i < 8) { base = 3 + i; extra = 0; return; }
    if (i == 28) { base = 258; extra = 0; return; }
    extra = (i - 4) >> 2;
    base = 3 + ((4 + (i & 3)) << extra);
}
FQ_DEV void inf_dist_base(int sym, int& base, int& extra) {  // sym = 0..29
    if (sym < 4) { base = 1 + sym; extra = 0; return; }
    extra = (sym - 2) >> 1;
    base = 1 + ((2 + (sym & 1)) << extra);
}

// store the low n (< 8) bytes of v
FQ_DEV void inf_st_tail(u8* p, u64 v, int n) {
    for (int i = 0; i < n; i++) {
        p[i] = (u8)v;
        v >>= 8;
    }
}

// LZ77 copy of len bytes from dst - dist to dst (the regions may overlap: the pattern repeats).  A byte loop
// would pay one memory round trip per byte (every load may alias the store before it); here
//   dist >= 8 : eight bytes per step, the loads of a 32-byte group issued before its stores when dist >= 32
//   dist <  8 : the dist-byte pattern is read ONCE, widened to a periodic 8-byte word, and only stored
// Stores never go past dst + len (the next block's text belongs to another lane).
FQ_DEV void inf_copy(u8* dst, u32 dist, int len) {
    const u8* src = dst - dist;
    if (dist >= 32u) {
        int i = 0;
        for (; i + 32 <= len; i += 32) {
            const u64 a = inf_ld8(src + i), b = inf_ld8(src + i + 8), c = inf_ld8(src + i + 16), d = inf_ld8(src + i + 24);
            inf_st8(dst + i, a);
            inf_st8(dst + i + 8, b);
            inf_st8(dst + i + 16, c);
            inf_st8(dst + i + 24, d);
        }
        if (i < len) {  // < 32 bytes left: reads may run past src + len (still inside the lane's own text), stores are exact
            const u64 a = inf_ld8(src + i), b = inf_ld8(src + i + 8), c = inf_ld8(src + i + 16), d = inf_ld8(src + i + 24);
            const u64 w[4] = {a, b, c, d};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int left = len - i - 8 * k;
                if (left >= 8) inf_st8(dst + i + 8 * k, w[k]);
                else if (left > 0) inf_st_tail(dst + i + 8 * k, w[k], left);
            }
        }
        return;
    }
    if (dist >= 8u) {
        int i = 0;
        for (; i + 8 <= len; i += 8) inf_st8(dst + i, inf_ld8(src + i));
        if (i < len) inf_st_tail(dst + i, inf_ld8(src + i), len - i);
        return;
    }
    // dist 1..7: periodic word.  step = the largest multiple of dist that fits in 8 bytes
    u64 pat = 0;   // byte loads: independent of each other (one round trip), and nothing past dst is touched
    for (u32 k = 0; k < dist; k++) pat |= (u64)src[k] << (8 * k);
    u64 word = pat;
    for (u32 filled = dist; filled < 8u; filled += dist) word |= pat << (8 * filled);
    const int step = (int)((8u / dist) * dist);
    int i = 0;
    for (; i + 8 <= len; i += step) inf_st8(dst + i, word);
    // the word starts a period at every multiple of step, so the tail is its low bytes
    if (i < len) inf_st_tail(dst + i, word, len - i);
}

FQ_DEV u32 inf_crc32_update(u32 crc, u32 byte) {  // bitwise, reflected 0xEDB88320: no table, no memory traffic
    crc ^= byte;
#pragma unroll
    for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    return crc;
}

FQ_DEV u32 inflate_block(const InflateArgs& a, u16* tab, int lane, int g) {
    InfBits b;
    inf_start(b, a.comp + a.pay_off[g], a.pay_len[g], (u32*)(tab + INF_ENTRIES * INF_LANES), lane);
    u8* out = a.out + a.out_off[g];
    const u32 cap = a.isize[g];
    if (a.out_off[g] + cap > a.out_cap) return INF_E_ISIZE;
    u32 opos = 0;
    u8* lens = a.scratch + (size_t)g * INF_SCRATCH;
    int last;
    do {
        last = (int)inf_bits(b, 1);
        const int type = (int)inf_bits(b, 2);
        if (type == 0) {  // stored: to the byte boundary, LEN, NLEN, bytes
            const int drop = b.cnt & 7;
            b.buf >>= drop;
            b.cnt -= drop;
            const u32 len = inf_bits(b, 16);
            const u32 nlen = inf_bits(b, 16);
            if ((len ^ 0xFFFFu) != nlen) return INF_E_STORED;
            if (opos + len > cap) return INF_E_ISIZE;
            // the bit buffer holds whole bytes now: hand them back, then copy from the stream
            const u32 at = inf_pos(b);
            if (at + len > b.end) return INF_E_OVERRUN;
            const u8* in = b.base + b.skew;
            u32 i = 0;
            for (; i + 8 <= len; i += 8) inf_st8(out + opos + i, inf_ld8(in + at + i));
            for (; i < len; i++) out[opos + i] = in[at + i];
            opos += len;
            inf_seek(b, at + len);
            continue;
        }
        if (type == 3) return INF_E_BTYPE;
        InfCounts lc, dc;
        if (type == 1) {  // fixed codes (3.2.6)
            auto dl = [](int) -> u32 { return 5u; };
            inf_construct(tab, lane, INF_O_CNT, INF_O_DSYM, 30, dl, true);
            inf_load_counts(dc, tab, lane, INF_O_CNT);
            auto ll = [](int s) -> u32 { return s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u; };
            inf_construct(tab, lane, INF_O_CNT, INF_O_LSYM, 288, ll, false);
            inf_load_counts(lc, tab, lane, INF_O_CNT);
        } else {  // dynamic codes (3.2.7)
            const int nlen = (int)inf_bits(b, 5) + 257;
            const int ndist = (int)inf_bits(b, 5) + 1;
            const int ncode = (int)inf_bits(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return INF_E_TABLE;
            for (int i = 0; i < 19; i++) lens[i] = 0;
            for (int i = 0; i < ncode; i++) {
                // order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15, five bits per entry in two constants
                const u64 lo = 0x022CAA324E804A30ull;  // entries 0..11
                const u64 hi = 0x00000003C2E1346Cull;  // entries 12..18
                const int idx = i < 12 ? (int)((lo >> (5 * i)) & 31u) : (int)((hi >> (5 * (i - 12))) & 31u);
                lens[idx] = (u8)inf_bits(b, 3);
            }
            auto cl = [=](int s) -> u32 { return (u32)lens[s]; };
            if (!inf_construct(tab, lane, INF_O_CNT, INF_O_LSYM, 19, cl, false)) return INF_E_TABLE;
            InfCounts cc;
            inf_load_counts(cc, tab, lane, INF_O_CNT);
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = inf_decode(b, cc, tab, lane, INF_O_LSYM);
                if (sym < 0) return INF_E_CODE;
                if (sym < 16) {
                    lens[idx++] = (u8)sym;
                } else {
                    int rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) return INF_E_TABLE;
                        val = lens[idx - 1];
                        rep = 3 + (int)inf_bits(b, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)inf_bits(b, 3);
                    } else {
                        rep = 11 + (int)inf_bits(b, 7);
                    }
                    if (idx + rep > nlen + ndist) return INF_E_TABLE;
                    while (rep--) lens[idx++] = (u8)val;
                }
            }
            if (lens[256] == 0) return INF_E_TABLE;  // no end-of-block code
            // the count rows are shared construction scratch: each code's counts move to registers before the next is built
            auto dl = [=](int s) -> u32 { return (u32)lens[nlen + s]; };
            if (!inf_construct(tab, lane, INF_O_CNT, INF_O_DSYM, ndist, dl, true)) return INF_E_TABLE;
            inf_load_counts(dc, tab, lane, INF_O_CNT);
            auto ll = [=](int s) -> u32 { return (u32)lens[s]; };
            if (!inf_construct(tab, lane, INF_O_CNT, INF_O_LSYM, nlen, ll, true)) return INF_E_TABLE;
            inf_load_counts(lc, tab, lane, INF_O_CNT);
        }
        // ---- the symbols of this block ----
        // literals collect in a register and leave eight at a time (one store instead of eight); they are flushed
        // before anything that reads or orders against them: a match, the end of the block
        u64 lit = 0;
        int nlit = 0;
        for (;;) {
            int sym = inf_decode(b, lc, tab, lane, INF_O_LSYM);
            if (sym < 0) return INF_E_CODE;
            if (sym < 256) {
                if (opos >= cap) return INF_E_ISIZE;
                lit |= (u64)(u32)sym << (8 * nlit);
                nlit++;
                opos++;
                if (nlit == 8) {
                    inf_st8(out + opos - 8, lit);
                    lit = 0;
                    nlit = 0;
                }
                continue;
            }
            if (nlit) {
                inf_st_tail(out + opos - (u32)nlit, lit, nlit);
                lit = 0;
                nlit = 0;
            }
            if (sym == 256) break;
            if (sym > 285) return INF_E_CODE;
            int base, extra;
            inf_len_base(sym, base, extra);
            const int len = base + (int)inf_bits(b, extra);
            const int ds = inf_decode(b, dc, tab, lane, INF_O_DSYM);
            if (ds < 0 || ds > 29) return INF_E_DIST;
            inf_dist_base(ds, base, extra);
            // up to 13 extra bits: within inf_bits' 16
            const u32 dist = (u32)base + inf_bits(b, extra);
            if (dist > opos) return INF_E_DIST;
            if (opos + (u32)len > cap) return INF_E_ISIZE;
            inf_copy(out + opos, dist, len);
            opos += (u32)len;
        }
        if (inf_pos(b) > b.end) return INF_E_OVERRUN;
    } while (!last);
    if (opos != cap) return INF_E_ISIZE;
    if (a.check_crc) {
        u32 crc = 0xFFFFFFFFu;
        for (u32 i = 0; i < cap; i++) crc = inf_crc32_update(crc, out[i]);
        if ((crc ^ 0xFFFFFFFFu) != a.crc[g]) return INF_E_CRC;
    }
    return INF_OK;
}

FQ_DEV void inflate_body(const InflateArgs& a, u16* tab) {
    const int lane = lane_id();
    const int g = block_id() * block_threads() + thread_id();
    if (g >= a.n) return;
    const u32 st = inflate_block(a, tab, lane, g);
    a.status[g] = st;
    if (st != INF_OK) g_atomic_min_u32(a.first_bad, (u32)g);
}

}  // namespace fq
