/* fastp_gpu_host.h - the HOST side of the boundary in C++ (C linkage): what a patched fastp worker
 * does with the engine's per-read records after fastp_gpu_submit_host.
 *
 * Reference code this stands in for (OpenGene/fastp v1.3.6):
 *   the part of PairEndProcessor::processPairEnd / SingleEndProcessor::processSingleEnd that
 *   touches strings - output routing src/peprocessor.cpp:518-621, src/seprocessor.cpp:280-290,
 *   Read::appendToString src/read.cpp:119-154, OverlapAnalysis::merge's string assembly
 *   src/overlapanalysis.cpp:148-179, the UMI name edit src/umiprocessor.cpp:19-81 and
 *   FilterResult::addAdapterTrimmed with its insertion caps src/filterresult.cpp:115-180.
 * It never takes a trimming or filtering decision: those come from the device records.
 * fastp_amd/hostloop.py is the same logic in Python; tests run both against the golden files.
 */
#ifndef FASTP_GPU_HOST_H
#define FASTP_GPU_HOST_H

#include "fastp_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* a ReadPack (src/read.h:43-69) as the reader thread leaves it: four strings per read */
typedef struct fastp_gpu_reads {
    int32_t n;
    const char* const* name;   const int32_t* name_len;
    const char* const* seq;    const char* const* qual;  const int32_t* len;
    const char* const* strand; const int32_t* strand_len;
} fastp_gpu_reads;

/* FASTP_GPU_UMI_* and the FASTP_GPU_OUT1 .. FASTP_GPU_N_OUTPUTS stream indices are declared in fastp_gpu.h */

typedef struct fastp_gpu_host_options {
    int32_t want_failed;      /* --failed_out given   */
    int32_t want_unpaired1;   /* --unpaired1 given    */
    int32_t want_unpaired2;   /* --unpaired2 given    */
    int32_t umi_loc;          /* FASTP_GPU_UMI_*      */
    int32_t umi_len;
    const char* umi_prefix;   /* may be NULL          */
    const char* umi_delimiter;/* NULL = ":"           */
} fastp_gpu_host_options;

typedef struct fastp_gpu_host fastp_gpu_host;  /* a worker's output strings + FilterResult's adapter maps */

int fastp_gpu_host_create(const fastp_gpu_params* params, const fastp_gpu_host_options* opts, fastp_gpu_host** out);
void fastp_gpu_host_destroy(fastp_gpu_host* h);

/* apply the results of one pack (host pointers in `res`); r2 == NULL for single-end */
int fastp_gpu_host_apply(fastp_gpu_host* h, const fastp_gpu_reads* r1, const fastp_gpu_reads* r2,
                         const fastp_gpu_results* res);

/* accumulated text of one output stream (what goes to WriterThread::input), NULL if not wanted */
const char* fastp_gpu_host_output(fastp_gpu_host* h, int which, size_t* len);
void fastp_gpu_host_clear_outputs(fastp_gpu_host* h);

/* FilterResult::mAdapter1 / mAdapter2, in std::map order */
int64_t fastp_gpu_host_adapter_entries(fastp_gpu_host* h, int is_r2);
int fastp_gpu_host_adapter_entry(fastp_gpu_host* h, int is_r2, int64_t index, const char** seq, int32_t* len,
                                 int64_t* count);

/* FilterResult::addAdapterTrimmed (src/filterresult.cpp:124-180) for a host that cuts the adapter strings out of
 * its own text (fastp_gpu_stream.h): the one-read form, and the pair form of trimByOverlapAnalysis - read 2's string
 * is not recorded when a cap refused read 1's.  Empty strings are not recorded (:125). */
int fastp_gpu_host_add_adapter(fastp_gpu_host* h, int is_r2, const char* adapter, int32_t len);
int fastp_gpu_host_add_adapter_pair(fastp_gpu_host* h, const char* a1, int32_t len1, const char* a2, int32_t len2);

#ifdef __cplusplus
}
#endif
#endif /* FASTP_GPU_HOST_H */
