// FastqReader::read alone over a plain FASTQ file (the patched copy of the reference reader): FASTP_GPU=1 switches the
// memchr scan of fastp_gpu_reader_scan_eol on.  Build: see profiles/r03_reader_hook_microbench.txt
// reader-only harness: FastqReader::read over a file, reads per second (FASTP_GPU=1 switches the memchr scan on)
#include <chrono>
#include <cstdio>
#include "src/fastqreader.h"
#include "src/read.h"
#include <mutex>
#include <string>
std::string command;
std::mutex logmtx;
int main(int argc, char** argv) {
    FastqReader r(argv[1], true, false, 0);
    long n = 0, bases = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (Read* x = r.read()) { n++; bases += x->length(); delete x; }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%ld reads (%ld bases) in %.3f s = %.2f Mreads/s per reader thread\n", n, bases, dt, n / dt / 1e6);
    return 0;
}
