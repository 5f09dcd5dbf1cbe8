// tools/gunzip_bench.cpp - the stream's host inflaters (fastp_amd/csrc/fq_gunzip.h: one thread; fq_pgunzip.h: several threads
// on one stream) next to zlib's on one gzip file, all taking the text 16 MiB at a time as the stream's trips do.  CRC-32 /
// ISIZE checked by all.
//   g++ -O2 -std=c++17 -pthread tools/gunzip_bench.cpp -lz -o /tmp/gunzip_bench && /tmp/gunzip_bench reads.fq.gz [threads ...]
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <time.h>
#include <zlib.h>

#include <memory>
#include <vector>

#include "../fastp_amd/csrc/fq_pgunzip.h"

static double now() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb)) return 1;
    std::vector<uint8_t> out(16 << 20);
    for (int rep = 0; rep < 3; rep++) {
        for (int a = 2; a < argc; a++) {
            std::unique_ptr<fqgz::ParallelGunzip> g(new fqgz::ParallelGunzip());
            g->fd = fd;
            g->fsize = sb.st_size;
            g->threads = atoi(argv[a]);
            if (getenv("PG_CHUNK")) g->chunk = (size_t)atol(getenv("PG_CHUNK"));
            const double t0 = now();
            int64_t total = 0;
            uint64_t sum = 0;
            for (;;) {
                int err = 0;
                const int64_t n = g->read(out.data(), (int64_t)out.size(), &err);
                if (n < 0) { printf("fq_pgunzip: error %d (%lld batches, %lld chunks used, %lld dropped, %lld marker faults)\n", err, (long long)g->batches, (long long)g->chunks_used, (long long)g->chunks_dropped, (long long)g->marker_faults); return 1; }
                total += n;
                sum += out[0] + out[(size_t)(n > 0 ? n - 1 : 0)];
                if (n < (int64_t)out.size()) break;
            }
            const double dt = now() - t0;
            printf("fq_pgunzip.h x%-2d: %lld bytes of text in %.3f s = %7.1f MB/s  (check %llu; %lld batches, %lld chunks used, %lld dropped)\n", g->threads,
                   (long long)total, dt, total / dt / 1e6, (unsigned long long)sum, (long long)g->batches, (long long)g->chunks_used, (long long)g->chunks_dropped);
            printf("                  seconds in: read + find + decode %.3f, windows %.3f, resolve + CRC %.3f, trailers %.3f\n", g->t_phase[2], g->t_phase[3],
                   g->t_phase[4], g->t_phase[5]);
        }
        {
            std::unique_ptr<fqgz::Gunzip> g(new fqgz::Gunzip());
            g->fd = fd;
            g->fsize = sb.st_size;
            const double t0 = now();
            int64_t total = 0;
            uint64_t sum = 0;
            for (;;) {
                int err = 0;
                const int64_t n = g->read(out.data(), (int64_t)out.size(), &err);
                if (n < 0) { printf("fq_gunzip: error %d\n", err); return 1; }
                total += n;
                sum += out[0] + out[(size_t)(n > 0 ? n - 1 : 0)];
                if (n < (int64_t)out.size()) break;
            }
            const double dt = now() - t0;
            printf("fq_gunzip.h : %lld bytes of text in %.3f s = %7.1f MB/s  (check %llu)\n", (long long)total, dt, total / dt / 1e6, (unsigned long long)sum);
        }
        {
            std::vector<uint8_t> in(4 << 20);
            z_stream z;
            memset(&z, 0, sizeof(z));
            inflateInit2(&z, 31);
            const double t0 = now();
            int64_t total = 0, fpos = 0;
            uint64_t sum = 0;
            bool end = false;
            while (!end) {
                z.next_out = out.data();
                z.avail_out = (uInt)out.size();
                while (z.avail_out) {
                    if (!z.avail_in) {
                        const ssize_t r = pread(fd, in.data(), in.size(), fpos);
                        if (r <= 0) { end = true; break; }
                        fpos += r;
                        z.next_in = in.data();
                        z.avail_in = (uInt)r;
                    }
                    const int rc = inflate(&z, Z_NO_FLUSH);
                    if (rc == Z_STREAM_END) inflateReset(&z);
                    else if (rc != Z_OK && rc != Z_BUF_ERROR) { printf("zlib: error %d\n", rc); return 1; }
                }
                const int64_t n = (int64_t)out.size() - z.avail_out;
                total += n;
                sum += out[0] + out[(size_t)(n > 0 ? n - 1 : 0)];
            }
            const double dt = now() - t0;
            inflateEnd(&z);
            printf("zlib %s : %lld bytes of text in %.3f s = %7.1f MB/s  (check %llu)\n", zlibVersion(), (long long)total, dt, total / dt / 1e6, (unsigned long long)sum);
        }
    }
    return 0;
}
