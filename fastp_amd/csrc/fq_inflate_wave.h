// fq_inflate_wave.h - one WAVEFRONT per BGZF block (the cooperative version of fq_inflate.h; same arguments, same
// status codes, same checks).  The stage replaced is one igzip call of the reference's BgzfMtReader worker
// (/root/reference/src/bgzf.h:165-195).
//
// A block's text (<= 64 KiB) is assembled in LDS, so an LZ77 copy never waits for global memory, and leaves as one
// coalesced stream at the end.  Huffman decoding is made parallel by speculation:
//   * lane k decodes the token that WOULD start at bit cur + k of the stream (literal, or length + distance with
//     their extra bits: two table lookups), for k = 0..63;
//   * the tokens that really occur are the chain 0 -> 0 + bits(0) -> ..., walked with v_readlane (a few cycles per
//     token, no LDS round trip); the walk also hands each token its output position;
//   * literals are stored by their lanes at once; each match is copied by the whole wave.
// The code lengths of a dynamic block (RFC 1951 3.2.7) are decoded by the same speculate-and-walk step, and the
// decoding tables (10-bit direct table for literals/lengths, 8-bit for distances, canonical walk for longer codes)
// are built by the wave with ballots.
#pragma once
#include "fq_deflate.h"   // CrcLds
#include "fq_inflate.h"   // InflateArgs, status codes, length / distance bases

namespace fq {

enum {
    IW_LROOT = 10, IW_DROOT = 8, IW_CROOT = 7,
    IW_RING = 4096,        // bytes of compressed stream staged in LDS (two halves of 2 KiB)
    IW_LONG = 0x8000,      // table entry: code longer than the root, walk the canonical arrays
};

struct IwLds {
    u8 out[65536];
    u32 ring[IW_RING / 4];
    u16 ltab[1 << IW_LROOT], dtab[1 << IW_DROOT];
    u16 ctab[1 << IW_CROOT];
    u16 cnt[3][16];        // per-length code counts: 0 literal/length, 1 distance, 2 code-length alphabet
    u16 lsym[288], dsym[32];
    u8 lens[320 + 64];
    CrcLds crc;
};

// ---- the compressed stream ------------------------------------------------------------------------------------
// offsets are relative to `base`, the 16-byte aligned address at or below the payload; the ring holds stream bytes
// [loaded - IW_RING, loaded)
struct IwIn {
    const u8* base;
    u32 limit;     // 16-byte loads may START below this stream offset
    u32 loaded;
};
FQ_DEV void iw_fill_half(IwLds& S, IwIn& in, int lane) {
    const u32 at = in.loaded + 32u * (u32)lane;
    u32x4 a, b;
    a.x = a.y = a.z = a.w = 0u;
    b = a;
    if (at < in.limit) a = *(const u32x4*)(in.base + at);
    if (at + 16u < in.limit) b = *(const u32x4*)(in.base + at + 16u);
    u32* r = S.ring + ((at & (IW_RING - 1)) >> 2);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
    r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    in.loaded += IW_RING / 2;
}
// make the ring cover [cur_byte - 3, cur_byte + 28)
FQ_DEV void iw_need(IwLds& S, IwIn& in, u32 cur_bit, int lane) {
    const u32 b = cur_bit >> 3;
    if (b >= in.loaded) {   // a jump (stored block): restart two halves below
        wave_sync();
        in.loaded = b & ~(u32)(IW_RING / 2 - 1);
        iw_fill_half(S, in, lane);
        iw_fill_half(S, in, lane);
        wave_sync();
    }
    while (b + IW_RING / 2 > in.loaded) {
        wave_sync();
        iw_fill_half(S, in, lane);
        wave_sync();
    }
}
// 57+ stream bits starting at bit `o`
FQ_DEV u64 iw_peek(const IwLds& S, u32 o) {
    const u32 b = o >> 3, d = (b & ~3u) & (IW_RING - 1);
    const u32 w0 = S.ring[d >> 2], w1 = S.ring[((d + 4u) & (IW_RING - 1)) >> 2], w2 = S.ring[((d + 8u) & (IW_RING - 1)) >> 2];
    const u32 sh = (b & 3u) * 8u + (o & 7u);   // 0..31
    const u64 lo = (u64)w0 | ((u64)w1 << 32);
    return sh ? (lo >> sh) | ((u64)w2 << (64u - sh)) : lo;
}

// ---- decoding tables ------------------------------------------------------------------------------------------
// n code lengths at S.lens[first ..): counts, canonical codes, the direct table `tab` (root bits; entry =
// symbol << 4 | length, 0 = no such code, IW_LONG = longer than the root) and the symbols in code order.
// Returns false for an over-subscribed set, or an incomplete one unless allow_incomplete.
template <class T>
FQ_DEV bool iw_build(IwLds& S, int first, int n, int which, T* tab, int root, u16* symtab, bool allow_incomplete, int lane) {
    u16* cnt = S.cnt[which];
    if (lane < 16) cnt[lane] = 0;
    for (int i = lane; i < (1 << root); i += 64) tab[i] = 0;
    wave_sync();
    // codes in symbol order: 64 symbols per step, ranks inside a step by ballots
    u32 next[16];    // symbols seen so far per length (uniform)
#pragma unroll
    for (int l = 0; l < 16; l++) next[l] = 0;
    u32 my_rank[5], my_len[5];
    const int steps = (n + 63) >> 6;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        my_rank[j] = 0;
        my_len[j] = 0;
        if (j < steps) {
            const int s = j * 64 + lane;
            const u32 l = s < n ? (u32)S.lens[first + s] : 0u;
            my_len[j] = l;
#pragma unroll
            for (int L = 1; L < 16; L++) {
                const u64 m = ballot(l == (u32)L);
                if (l == (u32)L) my_rank[j] = next[L] + (u32)popc64(m & ((1ull << lane) - 1ull));
                next[L] += (u32)popc64(m);
            }
        }
    }
    // Kraft check, first code and first symbol-table index of each length
    u32 firstc[16], offs[16];
    int left = 1;
    u32 code = 0, at = 0;
    bool over = false;
#pragma unroll
    for (int L = 1; L < 16; L++) {
        left <<= 1;
        left -= (int)next[L];
        if (left < 0) over = true;
        firstc[L] = code;
        offs[L] = at;
        code = (code + next[L]) << 1;
        at += next[L];
    }
    if (lane < 16) {
        u32 c = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) c = lane == L ? next[L] : c;
        cnt[lane] = (u16)c;
    }
    if (over) return false;
    if (left > 0 && !allow_incomplete) return false;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        if (j >= steps) continue;
        const u32 l = my_len[j];
        if (!l) continue;
        const int s = j * 64 + lane;
        u32 fc = 0, of = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            fc = l == (u32)L ? firstc[L] : fc;
            of = l == (u32)L ? offs[L] : of;
        }
        symtab[of + my_rank[j]] = (u16)s;
        const u32 c = fc + my_rank[j];
        const u32 rev = brev32(c) >> (32u - l);
        if (l <= (u32)root) {
            for (u32 k = rev; k < (1u << root); k += 1u << l) tab[k] = (T)(((u32)s << 4) | l);
        } else {
            tab[rev & ((1u << root) - 1u)] = (T)IW_LONG;
        }
    }
    wave_sync();
    return true;
}

// canonical walk (RFC 1951 3.2.2) for a code longer than the root: symbol | length << 16, or -1
FQ_DEV int iw_walk(const u16* cnt, const u16* symtab, u32 bits) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = cnt[len];
        if (code - count < first) return (int)symtab[index + (code - first)] | (len << 16);
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// token kinds
enum { IW_T_BAD = 0, IW_T_LIT = 1, IW_T_MATCH = 2, IW_T_EOB = 3 };

// the token that would start at the low end of w: kind | bits << 2 | (literal or match length) << 8, distance
FQ_DEV u32 iw_token(const IwLds& S, u64 w, u32& dist) {
    dist = 0;
    u32 e = S.ltab[(u32)w & ((1u << IW_LROOT) - 1u)];
    u32 sym, nb;
    if (e == (u32)IW_LONG) {
        const int r = iw_walk(S.cnt[0], S.lsym, (u32)w);
        if (r < 0) return IW_T_BAD;
        sym = (u32)r & 0xFFFFu;
        nb = (u32)r >> 16;
    } else {
        if (!e) return IW_T_BAD;
        sym = e >> 4;
        nb = e & 15u;
    }
    if (sym < 256u) return IW_T_LIT | (nb << 2) | (sym << 8);
    if (sym == 256u) return IW_T_EOB | (nb << 2);
    if (sym > 285u) return IW_T_BAD;
    w >>= nb;
    int base, extra;
    inf_len_base((int)sym, base, extra);
    const u32 mlen = (u32)base + ((u32)w & ((1u << extra) - 1u));
    w >>= extra;
    nb += (u32)extra;
    e = S.dtab[(u32)w & ((1u << IW_DROOT) - 1u)];
    u32 ds, dn;
    if (e == (u32)IW_LONG) {
        const int r = iw_walk(S.cnt[1], S.dsym, (u32)w);
        if (r < 0) return IW_T_BAD;
        ds = (u32)r & 0xFFFFu;
        dn = (u32)r >> 16;
    } else {
        if (!e) return IW_T_BAD;
        ds = e >> 4;
        dn = e & 15u;
    }
    if (ds > 29u) return IW_T_BAD;
    w >>= dn;
    nb += dn;
    inf_dist_base((int)ds, base, extra);
    dist = (u32)base + ((u32)w & ((1u << extra) - 1u));
    nb += (u32)extra;
    return IW_T_MATCH | (nb << 2) | (mlen << 8);
}

FQ_DEV u32 inflate_wave_block(const InflateArgs& a, IwLds& S, int g, int lane) {
    const u8* pay = a.comp + a.pay_off[g];
    const u32 pay_len = a.pay_len[g];
    IwIn in;
    const u32 skew = (u32)((size_t)pay & 15u);
    in.base = pay - skew;
    in.limit = skew + pay_len + 8u;       // the member trailer; a 16-byte load starting below it ends inside the chunk's padding
    in.loaded = 0;
    const u32 cap = a.isize[g];
    if (cap > 65536u || a.out_off[g] + cap > a.out_cap) return INF_E_ISIZE;
    wave_sync();
    iw_fill_half(S, in, lane);
    iw_fill_half(S, in, lane);
    wave_sync();
    u32 cur = skew * 8u;                      // bit position in the stream
    const u32 end_bit = (skew + pay_len) * 8u;
    u32 opos = 0;
    u32 last;
    do {
        iw_need(S, in, cur, lane);
        u64 w = iw_peek(S, cur);
        last = (u32)w & 1u;
        const u32 type = ((u32)w >> 1) & 3u;
        cur += 3u;
        if (type == 0u) {   // stored: to the byte boundary, LEN, ~LEN, the bytes (straight from the payload)
            cur = (cur + 7u) & ~7u;
            iw_need(S, in, cur, lane);
            w = iw_peek(S, cur);
            const u32 len = (u32)w & 0xFFFFu, nlen = ((u32)w >> 16) & 0xFFFFu;
            if ((len ^ 0xFFFFu) != nlen) return INF_E_STORED;
            cur += 32u;
            if (opos + len > cap) return INF_E_ISIZE;
            if (cur + 8u * len > end_bit) return INF_E_OVERRUN;
            const u8* src = in.base + (cur >> 3);
            for (u32 i = (u32)lane; i < len; i += 64u) S.out[opos + i] = src[i];
            opos += len;
            cur += 8u * len;
            continue;
        }
        if (type == 3u) return INF_E_BTYPE;
        int nlen, ndist;
        if (type == 1u) {   // fixed codes (3.2.6)
            nlen = 288;
            ndist = 30;
            for (int s = lane; s < 288; s += 64) S.lens[s] = (u8)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
            if (lane < 30) S.lens[288 + lane] = 5;
            wave_sync();
        } else {            // dynamic codes (3.2.7)
            iw_need(S, in, cur, lane);
            w = iw_peek(S, cur);
            nlen = (int)((u32)w & 31u) + 257;
            ndist = (int)(((u32)w >> 5) & 31u) + 1;
            const int ncode = (int)(((u32)w >> 10) & 15u) + 4;
            cur += 14u;
            if (nlen > 286 || ndist > 30) return INF_E_TABLE;
            // the code-length code: 19 three-bit lengths in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
            iw_need(S, in, cur, lane);
            if (lane < 19) S.lens[320 + lane] = 0;
            wave_sync();
            if (lane < ncode) {
                const u64 lo = 0x022CAA324E804A30ull;  // entries 0..11, five bits each
                const u64 hi = 0x00000003C2E1346Cull;  // entries 12..18
                const int idx = lane < 12 ? (int)((lo >> (5 * lane)) & 31u) : (int)((hi >> (5 * (lane - 12))) & 31u);
                S.lens[320 + idx] = (u8)(iw_peek(S, cur + 3u * (u32)lane) & 7u);
            }
            cur += 3u * (u32)ncode;
            wave_sync();
            // (the code-length alphabet needs no symbol table: every code fits the root; S.lsym is rebuilt below)
            if (!iw_build(S, 320, 19, 2, S.ctab, IW_CROOT, S.lsym, false, lane)) return INF_E_TABLE;
            // the nlen + ndist code lengths: speculate at every bit offset, walk the chain
            int idx = 0;
            u32 prev = 0;
            const int total = nlen + ndist;
            while (idx < total) {
                iw_need(S, in, cur, lane);
                const u64 v = iw_peek(S, cur + (u32)lane);
                const u32 e = S.ctab[(u32)v & ((1u << IW_CROOT) - 1u)];
                const u32 sym = e >> 4, cl = e & 15u;
                u32 nb = cl, ext = 0;
                if (sym == 16u) { ext = ((u32)(v >> cl) & 3u); nb += 2u; }
                else if (sym == 17u) { ext = ((u32)(v >> cl) & 7u); nb += 3u; }
                else if (sym == 18u) { ext = ((u32)(v >> cl) & 127u); nb += 7u; }
                const u32 packed = e ? (nb | (sym << 8) | (ext << 16)) : 0u;
                u32 pos = 0;
                while (pos < 64u && idx < total) {
                    const u32 t = read_lane(packed, pos);
                    if (!t) return INF_E_CODE;
                    const u32 s = (t >> 8) & 31u, x = t >> 16;
                    int rep = 1;
                    u32 val = s;
                    if (s == 16u) {
                        if (idx == 0) return INF_E_TABLE;
                        val = prev;
                        rep = 3 + (int)x;
                    } else if (s == 17u) {
                        val = 0;
                        rep = 3 + (int)x;
                    } else if (s == 18u) {
                        val = 0;
                        rep = 11 + (int)x;
                    }
                    if (idx + rep > total) return INF_E_TABLE;
                    for (int k = lane; k < rep; k += 64) S.lens[idx + k] = (u8)val;
                    idx += rep;
                    prev = val;
                    pos += t & 0xFFu;
                }
                cur += pos;
            }
            wave_sync();
            if (S.lens[256] == 0) return INF_E_TABLE;   // no end-of-block code
        }
        if (!iw_build(S, nlen, ndist, 1, S.dtab, IW_DROOT, S.dsym, true, lane)) return INF_E_TABLE;
        if (!iw_build(S, 0, nlen, 0, S.ltab, IW_LROOT, S.lsym, type == 2u, lane)) return INF_E_TABLE;
        // ---- the symbols of this block ----
        for (;;) {
            iw_need(S, in, cur, lane);
            u32 dist;
            const u32 tk = iw_token(S, iw_peek(S, cur + (u32)lane), dist);
            u32 pos = 0, run = opos, my_out = 0;
            u64 lits = 0, matches = 0;
            bool eob = false;
            while (pos < 64u) {
                const u32 t = read_lane(tk, pos);
                const u32 kind = t & 3u;
                if (kind == IW_T_BAD) return INF_E_CODE;
                if ((u32)lane == pos) my_out = run;
                if (kind == IW_T_LIT) { lits |= 1ull << pos; run += 1u; }
                else if (kind == IW_T_MATCH) { matches |= 1ull << pos; run += t >> 8; }
                pos += (t >> 2) & 63u;
                if (kind == IW_T_EOB) { eob = true; break; }
            }
            if (run > cap) return INF_E_ISIZE;
            if ((lits >> lane) & 1ull) S.out[my_out] = (u8)(tk >> 8);
            wave_order();
            while (matches) {
                const u32 j = (u32)ffs64(matches) - 1u;
                matches &= matches - 1ull;
                const u32 mlen = read_lane(tk, j) >> 8, md = read_lane(dist, j), at = read_lane(my_out, j);
                if (md > at) return INF_E_DIST;
                const u8* src = S.out + at - md;
                if (md >= mlen) {
                    for (u32 k = (u32)lane; k < mlen; k += 64u) S.out[at + k] = src[k];
                } else {   // the pattern of md bytes repeats
                    const float inv = 1.0f / (float)md;
                    for (u32 k = (u32)lane; k < mlen; k += 64u) {
                        u32 q = (u32)((float)k * inv);
                        u32 r = k - q * md;
                        if ((int)r < 0) r += md;
                        if (r >= md) r -= md;
                        S.out[at + k] = src[r];
                    }
                }
                wave_order();
            }
            opos = run;
            cur += pos;
            if (eob) break;
        }
        if (cur > end_bit) return INF_E_OVERRUN;
    } while (!last);
    if (opos != cap) return INF_E_ISIZE;
    wave_sync();
    if (a.check_crc) {
        const u32 crc = def_crc32(S.crc, S.out, cap, lane);
        if (crc != a.crc[g]) return INF_E_CRC;
    }
    // the text leaves as one stream: 8 bytes per lane and step (any destination alignment)
    u8* dst = a.out + a.out_off[g];
    const u64* so = (const u64*)S.out;
    for (u32 i = (u32)lane; i * 8u + 8u <= cap; i += 64u) inf_st8(dst + 8u * i, so[i]);
    if ((u32)lane < (cap & 7u)) dst[(cap & ~7u) + (u32)lane] = S.out[(cap & ~7u) + (u32)lane];
    return INF_OK;
}

FQ_DEV void inflate_wave_body(const InflateArgs& a, u32* ldsw) {
    IwLds& S = *(IwLds*)ldsw;
    const int lane = lane_id();
    if (a.check_crc) def_crc_setup(S.crc, lane);
    for (int g = block_id(); g < a.n; g += grid_blocks()) {
        const u32 st = inflate_wave_block(a, S, g, lane);
        if (lane == 0) {
            a.status[g] = st;
            if (st != INF_OK) g_atomic_min_u32(a.first_bad, (u32)g);
        }
        wave_sync();
    }
}

}  // namespace fq
