// Scalar stand-in for the reference's Google-Highway translation unit.
//
// TEST INFRASTRUCTURE ONLY.  Highway 1.3.0 is an un-vendored dependency of the
// reference (ci.yml pins it) and is absent from this image, so the reference
// oracle binary links this file in place of src/simd.cpp.  It implements the
// five entry points declared in src/simd.h:12-31 with the obvious byte loops;
// the reference's own testSimd (simd.cpp:326-564) asserts its vector code is
// equivalent to exactly these scalar semantics, so callers see no difference.
#include "src/simd.h"
#include <cstdint>

namespace fastp_simd {

void countQualityMetrics(const char* qualstr, const char* seqstr, int len,
                         char qualThreshold, int& lowQualNum, int& nBaseNum,
                         int& totalQual) {
    int low = 0, nb = 0, tot = 0;
    const uint8_t thr = (uint8_t)qualThreshold;
    for (int k = 0; k < len; ++k) {
        const uint8_t q = (uint8_t)qualstr[k];
        tot += (int)q - 33;
        low += (q < thr);
        nb += (seqstr[k] == 'N');
    }
    lowQualNum = low; nBaseNum = nb; totalQual = tot;
}

void reverseComplement(const char* src, char* dst, int len) {
    for (int k = 0; k < len; ++k) {
        char out = 'N';
        switch (src[k] & ~0x20) {       // fold case: a/A, c/C, g/G, t/T
            case 'A': out = 'T'; break;
            case 'T': out = 'A'; break;
            case 'C': out = 'G'; break;
            case 'G': out = 'C'; break;
            default: break;
        }
        // only true letters fold; '!' (0x21)&~0x20 = 0x01 etc. never hit a case
        dst[len - 1 - k] = out;
    }
}

int countAdjacentDiffs(const char* data, int len) {
    int n = 0;
    for (int k = 1; k < len; ++k) n += (data[k] != data[k - 1]);
    return n;
}

int countMismatches(const char* a, const char* b, int len) {
    int n = 0;
    for (int k = 0; k < len; ++k) n += (a[k] != b[k]);
    return n;
}

int countMismatchesBounded(const char* a, const char* b, int len, int limit) {
    int n = 0;
    for (int k = 0; k < len; ++k) {
        n += (a[k] != b[k]);
        if (n > limit) return n;
    }
    return n;
}

bool testSimd() {
    // minimal self-check so `fastp_ref test` still exercises this shim
    char out[8];
    reverseComplement("ACGTNacgt", out, 4);
    if (!(out[0] == 'A' && out[1] == 'C' && out[2] == 'G' && out[3] == 'T')) return false;
    if (countMismatches("AAAA", "AATA", 4) != 1) return false;
    if (countMismatchesBounded("AAAA", "TTTT", 4, 1) <= 1) return false;
    if (countAdjacentDiffs("AACC", 4) != 1) return false;
    int lo, nb, tot;
    countQualityMetrics("!5?I", "ANNA", 4, '5', lo, nb, tot);
    return lo == 1 && nb == 2 && tot == (0 + 20 + 30 + 40);
}

}  // namespace fastp_simd
