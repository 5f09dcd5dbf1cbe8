/* slowread.c - TEST INFRASTRUCTURE: an LD_PRELOAD shim that makes fread() on ONE file slow, so that a test can make
 * one of reference fastp's two reader threads lag behind the other deterministically (tests/test_ref_binding.py).
 *   SLOWREAD_PATH  the file whose reads are delayed (compared with /proc/self/fd/N's target)
 *   SLOWREAD_US    microseconds slept per fread call on it
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

size_t fread(void* ptr, size_t size, size_t n, FILE* f) {
    static size_t (*real)(void*, size_t, size_t, FILE*) = NULL;
    if (!real) real = (size_t(*)(void*, size_t, size_t, FILE*))dlsym(RTLD_NEXT, "fread");
    const char* want = getenv("SLOWREAD_PATH");
    if (want && f) {
        char link[64], target[4096];
        snprintf(link, sizeof(link), "/proc/self/fd/%d", fileno(f));
        ssize_t k = readlink(link, target, sizeof(target) - 1);
        if (k > 0) {
            target[k] = 0;
            if (strcmp(target, want) == 0) {
                const char* us = getenv("SLOWREAD_US");
                usleep(us ? (useconds_t)atoi(us) : 100000);
            }
        }
    }
    return real(ptr, size, n, f);
}
