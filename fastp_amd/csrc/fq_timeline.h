// FASTP_GPU_TIMELINE=1: one stderr line per start-up / shut-down step with the milliseconds since the process was
// started (from /proc/self/stat's start time: 10 ms steps, which is what a 0.5 s start-up needs).  Profiling only: no
// result depends on it, nothing is printed without the variable.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>

namespace fq {

inline double timeline_now_s() {
    timespec ts;
    clock_gettime(CLOCK_BOOTTIME, &ts);   // the clock /proc's start time counts in
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

inline double timeline_process_start_s() {
    double t = timeline_now_s();
    if (FILE* f = fopen("/proc/self/stat", "r")) {
        char buf[1024];
        const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        buf[n] = 0;
        // field 22 (starttime, clock ticks since boot) counted from behind the command's closing parenthesis
        const char* p = nullptr;
        for (size_t i = 0; i < n; i++) if (buf[i] == ')') p = buf + i;
        if (p) {
            int field = 2;
            for (p++; *p; p++) {
                if (*p != ' ') continue;
                if (++field == 22) { t = (double)strtoull(p + 1, nullptr, 10) / (double)sysconf(_SC_CLK_TCK); break; }
            }
        }
    }
    return t;
}

inline void timeline(const char* what) {
    static const bool on = getenv("FASTP_GPU_TIMELINE") != nullptr;
    if (!on) return;
    static const double t0 = timeline_process_start_s();
    fprintf(stderr, "fastp_gpu timeline %9.1f ms  %s\n", (timeline_now_s() - t0) * 1e3, what);
}

}  // namespace fq
