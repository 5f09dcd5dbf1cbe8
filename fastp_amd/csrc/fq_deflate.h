// fq_deflate.h - output text -> gzip members on the device (second half).
// The reference compresses each worker's pack as one independent gzip member with libdeflate and places the
// members at ordered offsets (/root/reference/src/writerthread.cpp:118-168, src/writer.cpp:110-133; the member
// layout is what makes that parallel, src/common.h:27-30).  Same idea one level finer: the stream is cut into
// blocks of DEF_BLOCK bytes, each becomes one BGZF-framed gzip member (readable by any gzip, and by the
// reference's BgzfMtReader / fastp_gpu_inflate_bgzf block by block), compressed by ONE WAVEFRONT:
//   match    64 positions per step: 4-byte hash -> most recent earlier position (LDS table), plus distance 1
//            (quality runs); the longest of the two is the lane's candidate
//   select   greedy parse of the step by ballots (one iteration per chosen match, literals come for free)
//   count    literal/length and distance histograms in LDS
//   codes    length-limited Huffman codes (rank sort by the wave, two-queue merge + Kraft repair by one lane),
//            the code-length alphabet the same way (RFC 1951 3.2.7)
//   emit     64 tokens per step: bit lengths scanned across the wave, bits OR-ed into an LDS window, whole
//            dwords flushed to the member
//   crc      CRC-32 by 64 lanes x 1 KiB slices combined with precomputed zero-shift operators
// Whichever is smaller, the dynamic block or a stored block, is written.  The bytes differ from libdeflate's
// (DEFLATE has no canonical encoding); what is checked is that every inflater returns the text.
#pragma once
#include "fq_intrin.h"
#include "fq_types.h"

namespace fq {

enum {
    DEF_BLOCK = 65280,            // text bytes per member (bgzip's own block size: 0xff00)
    DEF_SLOT = 65536 + 64,        // scratch bytes per member; the member starts at +2 so its DEFLATE stream is dword aligned
    DEF_HASH_BITS = 12,           // 8 KB of LDS: ~19 KB per block in all, eight blocks per CU in flight
    DEF_WIN = 256,                // LDS bit window, dwords
    DEF_PROBE = 36,               // bytes a lane compares for its own candidate
    DEF_NLL = 288, DEF_ND = 32, DEF_NCL = 20
};

struct DeflateArgs {
    const u8* text;
    u64 nbytes;          // of this round
    int nblocks;
    u8* slots;           // [nblocks][DEF_SLOT]
    u32* tokens;         // [nblocks][DEF_BLOCK]
    u32* sizes;          // [nblocks] member bytes
    u64* offs;           // [nblocks] where each member goes in out (relative to out_base), [nblocks] = total
    u8* out;
    u64 out_base, out_cap;
};

// CRC-32 of up to 64 KiB by one wavefront: the byte table and the operators "append 1024 * 2^k zero bytes"
struct CrcLds {
    u32 tab[256];
    u32 mat[6][32];
};

struct DefLds {
    u16 head[1 << DEF_HASH_BITS];
    u32 hist_ll[DEF_NLL], hist_d[DEF_ND], hist_cl[DEF_NCL];
    u32 code_ll[DEF_NLL], code_d[DEF_ND], code_cl[DEF_NCL];   // bit-reversed code | length << 16
    u8 len_ll[DEF_NLL], len_d[DEF_ND], len_cl[DEF_NCL];
    u16 sorted[DEF_NLL], par_leaf[DEF_NLL], par_int[DEF_NLL];
    u32 wint[DEF_NLL];
    u8 dep_int[DEF_NLL];
    u32 blc[16], nxt[16];
    u8 cl_sym[DEF_NLL + DEF_ND], cl_ext[DEF_NLL + DEF_ND];
    u32 cl_n, hlit, hdist, hclen, dyn_bits;
    CrcLds crc;
    u32 win[DEF_WIN];
};

FQ_DEV u32 def_ld4(const u8* p) {
    u32 w;
    __builtin_memcpy(&w, p, 4);
    return w;
}
FQ_DEV int def_ctz32(u32 v) { return ffs32(v) - 1; }
FQ_DEV int def_log2(u32 v) { return 31 - clz32(v); }

// bytes two positions have in common, at most maxl (both readable for maxl bytes); eight bytes per dependent step
FQ_DEV u64 def_ld8(const u8* p) {
    u64 w;
    __builtin_memcpy(&w, p, 8);
    return w;
}
FQ_DEV u32 def_match(const u8* a, const u8* b, u32 maxl) {
    u32 l = 0;
    while (l + 8u <= maxl) {
        const u64 x = def_ld8(a + l) ^ def_ld8(b + l);
        if (x) return l + (u32)((ffs64(x) - 1) >> 3);
        l += 8u;
    }
    while (l < maxl && a[l] == b[l]) l++;
    return l;
}

// RFC 1951 3.2.5: length 3..258 -> symbol 257..285 + extra bits; distance 1..32768 -> symbol 0..29 + extra bits
FQ_DEV u32 def_len_sym(u32 len, u32& eb, u32& ev) {
    if (len == 258u) { eb = 0; ev = 0; return 285u; }
    const u32 v = len - 3u;
    if (v < 8u) { eb = 0; ev = 0; return 257u + v; }
    const u32 e = (u32)def_log2(v) - 2u;
    eb = e;
    ev = v & ((1u << e) - 1u);
    return 257u + 4u * e + 4u + ((v >> e) & 3u);
}
FQ_DEV u32 def_dist_sym(u32 dist, u32& eb, u32& ev) {
    const u32 v = dist - 1u;
    if (v < 4u) { eb = 0; ev = 0; return v; }
    const u32 e = (u32)def_log2(v) - 1u;
    eb = e;
    ev = v & ((1u << e) - 1u);
    return 2u * e + 2u + ((v >> e) & 1u);
}

// ---- CRC-32 (reflected 0xEDB88320) ----------------------------------------------------------------------------
FQ_DEV u32 def_gf2_times(const u32* mat, u32 vec) {
    u32 sum = 0;
    for (int i = 0; vec; vec >>= 1, i++)
        if (vec & 1u) sum ^= mat[i];
    return sum;
}
// byte table + the operators "append 1024 * 2^k zero bytes" (k = 0..5), built once per workgroup by the wave
FQ_DEV void def_crc_setup(CrcLds& S, int lane) {
    for (int i = lane; i < 256; i += 64) {
        u32 c = (u32)i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
        S.tab[i] = c;
    }
    u32* m = S.mat[0];
    if (lane < 32) m[lane] = lane == 0 ? 0xEDB88320u : 1u << (lane - 1);   // one zero bit
    wave_sync();
    // 13 squarings: 2^13 bits = 1024 bytes; then one more per level
    for (int sq = 0; sq < 13 + 5; sq++) {
        u32* src = sq < 13 ? S.mat[0] : S.mat[sq - 13];
        u32* dst = sq < 13 ? S.mat[0] : S.mat[sq - 12];
        u32 r = 0;
        if (lane < 32) r = def_gf2_times(src, src[lane]);
        wave_sync();
        if (lane < 32) dst[lane] = r;
        wave_sync();
    }
}
FQ_DEV u32 def_crc32(const CrcLds& S, const u8* in, u32 n, int lane) {
    if (n == 0u) return 0u;
    // the text right-aligned in a virtual 64 KiB message of leading zero bytes (which leave a zero register unchanged);
    // the 0xFFFFFFFF preset enters where the text starts
    const u32 pad = 65536u - n;
    u32 reg = 0;
    const u32 v0 = (u32)lane * 1024u;
    for (u32 v = v0 < pad ? pad : v0; v < v0 + 1024u; v++) {
        if (v == pad) reg = 0xFFFFFFFFu;
        reg = S.tab[(reg ^ in[v - pad]) & 0xFFu] ^ (reg >> 8);
    }
    for (int k = 0; k < 6; k++) {
        const u32 right = shfl(reg, lane + (1 << k));
        if ((lane & ((2 << k) - 1)) == 0) reg = def_gf2_times(S.mat[k], reg) ^ right;
    }
    return ~shfl(reg, 0);
}

// ---- length-limited Huffman code of n symbols ---------------------------------------------------------------
FQ_DEV u32 def_bitrev(u32 c, u32 len) { return brev32(c) >> (32u - len); }

FQ_DEV void def_build(DefLds& S, const u32* freq, int n, int maxbits, u8* len, u32* code, int lane) {
    int used = 0;
    for (int s = lane; s < n; s += 64) {
        len[s] = 0;
        code[s] = 0;
        const u32 f = freq[s];
        if (f) {
            used++;
            int r = 0;
            for (int t = 0; t < n; t++) {
                const u32 g = freq[t];
                r += (int)(g != 0u && (g < f || (g == f && t < s)));
            }
            S.sorted[r] = (u16)s;   // ascending (frequency, symbol)
        }
    }
    for (int sh = 1; sh < 64; sh <<= 1) used += (int)shfl_xor((u32)used, sh);
    const int m = used;
    wave_sync();
    if (lane == 0) {
        if (m == 0) {
            len[0] = len[1] = 1;    // no symbol used: still a complete (two-code) set for every inflater
        } else if (m == 1) {
            const int s = S.sorted[0];
            len[s] = 1;
            len[s == 0 ? 1 : 0] = 1;
        } else {
            // two-queue merge: leaves in sorted order, internal nodes come out in non-decreasing weight
            int li = 0, ii = 0;
            for (int ni = 0; ni < m - 1; ni++) {
                u32 w = 0;
                for (int k = 0; k < 2; k++) {
                    const bool leaf = li < m && (ii >= ni || freq[S.sorted[li]] <= S.wint[ii]);
                    if (leaf) { w += freq[S.sorted[li]]; S.par_leaf[li++] = (u16)ni; }
                    else { w += S.wint[ii]; S.par_int[ii++] = (u16)ni; }
                }
                S.wint[ni] = w;
            }
            S.dep_int[m - 2] = 0;
            for (int k = m - 3; k >= 0; k--) S.dep_int[k] = (u8)(S.dep_int[S.par_int[k]] + 1);
            u32 kraft = 0;
            for (int i = 0; i < m; i++) {
                int d = S.dep_int[S.par_leaf[i]] + 1;
                if (d > maxbits) d = maxbits;
                len[S.sorted[i]] = (u8)d;
                kraft += 1u << (maxbits - d);
            }
            const u32 full = 1u << maxbits;
            while (kraft > full) {   // clamped depths over-subscribe: lengthen the rarest symbol among the longest that can grow
                int best = -1, bl = 0;
                for (int i = 0; i < m; i++) {
                    const int l = len[S.sorted[i]];
                    if (l < maxbits && l > bl) { bl = l; best = i; }
                }
                len[S.sorted[best]] = (u8)(bl + 1);
                kraft -= 1u << (maxbits - bl - 1);
            }
            while (kraft < full) {   // room left: shorten the most frequent symbol of the longest length (adds the smallest unit)
                int L = 0, best = 0;
                for (int i = 0; i < m; i++) {
                    const int l = len[S.sorted[i]];
                    if (l >= L) { L = l; best = i; }
                }
                len[S.sorted[best]] = (u8)(L - 1);
                kraft += 1u << (maxbits - L);
            }
        }
        // canonical codes (RFC 1951 3.2.2), stored bit-reversed for the LSB-first stream
        for (int b = 0; b < 16; b++) S.blc[b] = 0;
        for (int s = 0; s < n; s++) S.blc[len[s]]++;
        S.blc[0] = 0;
        u32 c = 0;
        for (int b = 1; b <= maxbits; b++) {
            c = (c + S.blc[b - 1]) << 1;
            S.nxt[b] = c;
        }
        for (int s = 0; s < n; s++) {
            const u32 l = len[s];
            if (l) code[s] = def_bitrev(S.nxt[l]++, l) | (l << 16);
        }
    }
    wave_sync();
}

// ---- the bit window -----------------------------------------------------------------------------------------
// win[0] holds the partly filled dword at bit cursor `cur` (bits of the DEFLATE stream written so far)
FQ_DEV void def_put_serial(DefLds& S, u32& cur_local, u32 base_bits, u32 v, u32 nb) {   // one lane, nb <= 16
    const u32 off = cur_local - base_bits;
    S.win[off >> 5] |= v << (off & 31u);
    if ((off & 31u) + nb > 32u) S.win[(off >> 5) + 1u] |= v >> (32u - (off & 31u));
    cur_local += nb;
}
// flush the whole dwords of the window [base dword of old_cur, new_cur) to the stream, keep the partial one
FQ_DEV void def_flush(DefLds& S, u32* stream, u32 old_cur, u32 new_cur, int lane) {
    wave_sync();
    const u32 w0 = old_cur >> 5, w1 = new_cur >> 5;   // dwords [w0, w1) are complete
    const u32 nfull = w1 - w0;
    u32 v[DEF_WIN / 64];
    for (int k = 0; k < DEF_WIN / 64; k++) v[k] = S.win[k * 64 + lane];
    const u32 partial = S.win[nfull < DEF_WIN ? nfull : DEF_WIN - 1];
    wave_sync();
    for (int k = 0; k < DEF_WIN / 64; k++) {
        const u32 i = (u32)(k * 64 + lane);
        if (i < nfull) stream[w0 + i] = v[k];
        S.win[i] = 0;
    }
    wave_sync();
    if (lane == 0) S.win[0] = partial;
    wave_sync();
}

FQ_DEV void deflate_block(const DeflateArgs& a, DefLds& S, int blk, int lane) {
    const u64 start = (u64)blk * DEF_BLOCK;
    const u32 n = (u32)(a.nbytes - start < (u64)DEF_BLOCK ? a.nbytes - start : (u64)DEF_BLOCK);
    // (Staging the block's text in LDS was measured and dropped: 93 KB per block leaves one wavefront per CU, and a
    // single wavefront's time is set by instruction latency either way - 3.5 ms per block against 4.1 from global memory
    // with five blocks per CU in flight.)
    const u8* in = a.text + start;
    u8* member = a.slots + (size_t)blk * DEF_SLOT + 2;
    u32* stream = (u32*)(member + 18);
    u32* tok = a.tokens + (size_t)blk * DEF_BLOCK;

    for (int i = lane; i < (1 << DEF_HASH_BITS); i += 64) S.head[i] = 0xFFFFu;
    for (int i = lane; i < DEF_NLL; i += 64) S.hist_ll[i] = 0;
    if (lane < DEF_ND) S.hist_d[lane] = 0;
    if (lane < DEF_NCL) S.hist_cl[lane] = 0;
    for (int i = lane; i < DEF_WIN; i += 64) S.win[i] = 0;
    wave_sync();

    // ---- match, select, count ----
    u32 ntok = 0, carry = 0;
    for (u32 base = 0; base < n; base += 64u) {
        const u32 p = base + (u32)lane;
        const u32 cnt = n - base < 64u ? n - base : 64u;
        u32 mlen = 0, mdist = 0, h = 0xFFFFFFFFu;
        // A lane compares at most DEF_PROBE bytes: in FASTQ text most positions sit inside long matches (names, quality
        // runs) that an earlier position takes, so full-length compares by all 64 lanes were most of the kernel's time.
        // The match that is actually chosen is extended by the whole wave below.
        u32 maxl = 0;
        if ((u32)lane < cnt && p + 4u <= n) {
            h = (def_ld4(in + p) * 2654435761u) >> (32 - DEF_HASH_BITS);
            if ((u32)lane >= carry) {   // positions an earlier match covers only feed the hash table
                const u32 c = S.head[h];
                maxl = n - p < 258u ? n - p : 258u;
                const u32 probe = maxl < (u32)DEF_PROBE ? maxl : (u32)DEF_PROBE;
                if (c != 0xFFFFu && p - c <= 32768u) {
                    const u32 l = def_match(in + p, in + c, probe);
                    if (l >= 4u || (l == 3u && p - c < 4096u)) { mlen = l; mdist = p - c; }
                }
                if (p >= 1u) {
                    const u32 l = def_match(in + p, in + p - 1, probe);
                    if (l >= 3u && l > mlen) { mlen = l; mdist = 1u; }
                }
            }
        }
        wave_sync();   // every lookup before any insert of this step
        if (h != 0xFFFFFFFFu) S.head[h] = (u16)p;
        wave_sync();
        const u64 mmask = ballot(mlen >= 3u);
        const u64 live = cnt == 64u ? ~0ull : (1ull << cnt) - 1ull;
        u64 starts = 0, mstart = 0;
        u32 pos = carry;
        while (pos < cnt) {
            const u64 m = mmask & (~0ull << pos);
            if (!m) {
                starts |= (~0ull << pos) & live;
                pos = cnt;
                break;
            }
            const u32 j = (u32)ffs64(m) - 1u;
            starts |= (~0ull << pos) & ((2ull << j) - 1ull);   // literals [pos, j) and the match at j
            mstart |= 1ull << j;
            u32 L = shfl(mlen, (int)j);
            const u32 mx = shfl(maxl, (int)j);
            if (L >= (u32)DEF_PROBE && L < mx) {   // the probe was cut short: 64 more bytes per step, all lanes
                const u32 pj = base + j, dj = shfl(mdist, (int)j);
                const u8* t = in;
                while (L < mx) {   // 4 bytes per lane: one step covers what is left of a 258-byte match
                    const u32 k = L + 4u * (u32)lane;
                    u32 same = 0;   // bytes of this lane's dword that match (from its low end), capped by mx
                    if (k < mx) {
                        const u32 lim = mx - k < 4u ? mx - k : 4u;
                        if (lim == 4u) {
                            const u32 x = def_ld4(t + pj + k) ^ def_ld4(t + pj - dj + k);
                            same = x ? (u32)(def_ctz32(x) >> 3) : 4u;
                        } else {
                            while (same < lim && t[pj + k + same] == t[pj - dj + k + same]) same++;
                        }
                    }
                    const u64 stop = ballot(same < 4u);          // lanes where the match ends (or the limit is)
                    const u32 f = stop ? (u32)ffs64(stop) - 1u : 64u;
                    L += 4u * f + (f < 64u ? shfl(same, (int)f) : 0u);
                    if (f < 64u) break;
                }
                if (L > mx) L = mx;
                if ((u32)lane == j) mlen = L;
            }
            pos = j + L;
        }
        carry = pos >= 64u ? pos - 64u : 0u;
        if ((starts >> lane) & 1ull) {
            const u32 idx = ntok + (u32)popc64(starts & ((1ull << lane) - 1ull));
            if ((mstart >> lane) & 1ull) {
                u32 eb, ev;
                tok[idx] = 0x80000000u | ((mlen - 3u) << 15) | (mdist - 1u);
                lds_add_u32(&S.hist_ll[def_len_sym(mlen, eb, ev)], 1u);
                lds_add_u32(&S.hist_d[def_dist_sym(mdist, eb, ev)], 1u);
            } else {
                const u32 b = in[p];
                tok[idx] = b;
                lds_add_u32(&S.hist_ll[b], 1u);
            }
        }
        ntok += (u32)popc64(starts);
    }
    wave_sync();
    if (lane == 0) S.hist_ll[256] = 1;   // end of block
    wave_sync();

    // ---- codes ----
    def_build(S, S.hist_ll, 286, 15, S.len_ll, S.code_ll, lane);
    def_build(S, S.hist_d, 30, 15, S.len_d, S.code_d, lane);
    if (lane == 0) {
        // the two length sequences, run-length coded with 16 / 17 / 18 (RFC 1951 3.2.7)
        u32 hlit = 286, hdist = 30;
        while (hlit > 257u && S.len_ll[hlit - 1] == 0) hlit--;
        while (hdist > 1u && S.len_d[hdist - 1] == 0) hdist--;
        const u32 total = hlit + hdist;
        u32 k = 0, i = 0;
        while (i < total) {
            const u32 v = i < hlit ? S.len_ll[i] : S.len_d[i - hlit];
            u32 run = 1;
            while (i + run < total && (i + run < hlit ? S.len_ll[i + run] : S.len_d[i + run - hlit]) == v) run++;
            i += run;
            if (v == 0u) {
                while (run >= 11u) { const u32 t = run < 138u ? run : 138u; S.cl_sym[k] = 18; S.cl_ext[k++] = (u8)(t - 11u); run -= t; }
                if (run >= 3u) { S.cl_sym[k] = 17; S.cl_ext[k++] = (u8)(run - 3u); run = 0; }
                while (run) { S.cl_sym[k] = 0; S.cl_ext[k++] = 0; run--; }
            } else {
                S.cl_sym[k] = (u8)v; S.cl_ext[k++] = 0; run--;
                while (run >= 3u) { const u32 t = run < 6u ? run : 6u; S.cl_sym[k] = 16; S.cl_ext[k++] = (u8)(t - 3u); run -= t; }
                while (run) { S.cl_sym[k] = (u8)v; S.cl_ext[k++] = 0; run--; }
            }
        }
        S.cl_n = k;
        S.hlit = hlit;
        S.hdist = hdist;
        for (u32 q = 0; q < k; q++) S.hist_cl[S.cl_sym[q]]++;
    }
    wave_sync();
    def_build(S, S.hist_cl, 19, 7, S.len_cl, S.code_cl, lane);
    const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (lane == 0) {
        u32 hclen = 19;
        while (hclen > 4u && S.len_cl[order[hclen - 1]] == 0) hclen--;
        S.hclen = hclen;
        u32 bits = 3u + 5u + 5u + 4u + 3u * hclen;
        for (u32 q = 0; q < S.cl_n; q++) {
            const u32 s = S.cl_sym[q];
            bits += S.len_cl[s] + (s == 16u ? 2u : s == 17u ? 3u : s == 18u ? 7u : 0u);
        }
        for (u32 s = 0; s < 286u; s++) {
            u32 extra = 0;
            if (s >= 265u && s < 285u) extra = (s - 261u) >> 2;
            bits += S.hist_ll[s] * (S.len_ll[s] + extra);
        }
        for (u32 s = 0; s < 30u; s++) bits += S.hist_d[s] * (S.len_d[s] + (s >= 4u ? (s - 2u) >> 1 : 0u));
        S.dyn_bits = bits;
    }
    wave_sync();
    const u32 dyn_bytes = (S.dyn_bits + 7u) >> 3;
    u32 dbytes;
    if (dyn_bytes >= n + 5u) {
        // ---- stored block: BFINAL = 1, BTYPE = 00, LEN, ~LEN, the bytes ----
        u8* d = member + 18;
        if (lane == 0) {
            d[0] = 1;
            d[1] = (u8)(n & 0xFFu);
            d[2] = (u8)(n >> 8);
            d[3] = (u8)(~n & 0xFFu);
            d[4] = (u8)((~n >> 8) & 0xFFu);
        }
        for (u32 i = (u32)lane; i < n; i += 64u) d[5 + i] = in[i];
        dbytes = n + 5u;
    } else {
        // ---- dynamic block header (one lane; at most ~4.5 Kbit, the window holds 8 Kbit) ----
        u32 cur = 0;
        if (lane == 0) {
            def_put_serial(S, cur, 0, 1u | (2u << 1), 3);
            def_put_serial(S, cur, 0, S.hlit - 257u, 5);
            def_put_serial(S, cur, 0, S.hdist - 1u, 5);
            def_put_serial(S, cur, 0, S.hclen - 4u, 4);
            for (u32 q = 0; q < S.hclen; q++) def_put_serial(S, cur, 0, S.len_cl[order[q]], 3);
            for (u32 q = 0; q < S.cl_n; q++) {
                const u32 s = S.cl_sym[q], c = S.code_cl[s];
                def_put_serial(S, cur, 0, c & 0xFFFFu, c >> 16);
                if (s >= 16u) def_put_serial(S, cur, 0, S.cl_ext[q], s == 16u ? 2u : s == 17u ? 3u : 7u);
            }
        }
        cur = shfl(cur, 0);
        def_flush(S, stream, 0, cur, lane);
        // ---- tokens, 64 per step, then the end-of-block code ----
        u32 t_next = (u32)lane < ntok ? tok[lane] : 0u;   // the next step's tokens are in flight while this step's bits are placed
        for (u32 t0 = 0; t0 <= ntok; t0 += 64u) {
            const u32 i = t0 + (u32)lane;
            u64 bits = 0;
            u32 nb = 0;
            const u32 t = t_next;
            t_next = i + 64u < ntok ? tok[i + 64u] : 0u;
            if (i < ntok) {
                if (!(t & 0x80000000u)) {
                    const u32 c = S.code_ll[t];
                    bits = c & 0xFFFFu;
                    nb = c >> 16;
                } else {
                    u32 eb, ev;
                    const u32 ls = def_len_sym(((t >> 15) & 0xFFu) + 3u, eb, ev);
                    u32 c = S.code_ll[ls];
                    bits = c & 0xFFFFu;
                    nb = c >> 16;
                    bits |= (u64)ev << nb;
                    nb += eb;
                    const u32 ds = def_dist_sym((t & 0x7FFFu) + 1u, eb, ev);
                    c = S.code_d[ds];
                    bits |= (u64)(c & 0xFFFFu) << nb;
                    nb += c >> 16;
                    bits |= (u64)ev << nb;
                    nb += eb;
                }
            } else if (i == ntok) {
                const u32 c = S.code_ll[256];
                bits = c & 0xFFFFu;
                nb = c >> 16;
            }
            u32 incl = nb;
            for (int sh = 1; sh < 64; sh <<= 1) {
                const u32 o = shfl(incl, lane - sh);
                if (lane >= sh) incl += o;
            }
            const u32 total = shfl(incl, 63);
            if (nb) {
                const u32 off = (cur & 31u) + incl - nb;
                const u32 w = off >> 5, sh = off & 31u;
                lds_or_u32(&S.win[w], (u32)(bits << sh));
                if (sh + nb > 32u) lds_or_u32(&S.win[w + 1u], (u32)(bits >> (32u - sh)));
                if (sh + nb > 64u) lds_or_u32(&S.win[w + 2u], (u32)(bits >> (64u - sh)));
            }
            def_flush(S, stream, cur, cur + total, lane);
            cur += total;
        }
        // the last partial dword
        dbytes = (cur + 7u) >> 3;
        if (lane == 0) {
            const u32 w = S.win[0];
            u8* tail = (u8*)(stream + (cur >> 5));
            for (u32 b = 0; b < dbytes - 4u * (cur >> 5); b++) tail[b] = (u8)(w >> (8u * b));
        }
    }
    // ---- gzip framing with the BGZF extra field (total size - 1), CRC-32 and ISIZE ----
    const u32 crc = def_crc32(S.crc, in, n, lane);
    if (lane == 0) {
        const u32 bsize = 18u + dbytes + 8u;
        const u8 hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        for (int i = 0; i < 16; i++) member[i] = hdr[i];
        member[16] = (u8)((bsize - 1u) & 0xFFu);
        member[17] = (u8)((bsize - 1u) >> 8);
        u8* tr = member + 18 + dbytes;
        for (int i = 0; i < 4; i++) tr[i] = (u8)(crc >> (8 * i));
        for (int i = 0; i < 4; i++) tr[4 + i] = (u8)(n >> (8 * i));
        a.sizes[blk] = bsize;
    }
    wave_sync();
}

FQ_DEV void deflate_body(const DeflateArgs& a, u32* ldsw) {
    DefLds& S = *(DefLds*)ldsw;
    const int lane = lane_id();
    def_crc_setup(S.crc, lane);
    for (int blk = block_id(); blk < a.nblocks; blk += grid_blocks()) deflate_block(a, S, blk, lane);
}

// offsets of the members: one workgroup, lanes own runs of blocks (the shape of fmt_scan_body)
FQ_DEV void deflate_scan_body(const DeflateArgs& a, u64* lds) {
    const int tid = thread_id(), nt = block_threads();
    const int per = (a.nblocks + nt - 1) / nt;
    const int b0 = tid * per < a.nblocks ? tid * per : a.nblocks, b1 = b0 + per < a.nblocks ? b0 + per : a.nblocks;
    u64 sum = 0;
    for (int b = b0; b < b1; b++) sum += a.sizes[b];
    lds[tid] = sum;
    block_sync();
    if (tid == 0) {
        u64 run = 0;
        for (int i = 0; i < nt; i++) {
            const u64 v = lds[i];
            lds[i] = run;
            run += v;
        }
        a.offs[a.nblocks] = run;
    }
    block_sync();
    u64 run = lds[tid];
    for (int b = b0; b < b1; b++) {
        a.offs[b] = run;
        run += a.sizes[b];
    }
}

// members to their place in the output (a workgroup per member)
FQ_DEV void deflate_gather_body(const DeflateArgs& a) {
    for (int blk = block_id(); blk < a.nblocks; blk += grid_blocks()) {
        const u32 sz = a.sizes[blk];
        const u64 at = a.out_base + a.offs[blk];
        if (at + sz > a.out_cap) continue;   // the host reports the needed size
        const u8* src = a.slots + (size_t)blk * DEF_SLOT + 2;
        u8* dst = a.out + at;
        for (u32 i = 4u * (u32)thread_id(); i + 4u <= sz; i += 4u * (u32)block_threads()) {
            u32 w;
            __builtin_memcpy(&w, src + i, 4);
            __builtin_memcpy(dst + i, &w, 4);
        }
        if (thread_id() == 0)
            for (u32 i = sz & ~3u; i < sz; i++) dst[i] = src[i];
    }
}

}  // namespace fq
