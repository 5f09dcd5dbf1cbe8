/* fastp_gpu_stream.h - FASTQ files in -> output FASTQ streams out, every per-read step on the device,
 * as ONE host-side loop in C++ behind a C ABI (the step either side of the hot path bound together:
 * SURVEY.md 8f rank 1 + the path + 8f rank 2).
 *
 * Reference code this stands in for (OpenGene/fastp v1.3.6) when a maintainer binds it into the
 * reference's own reader / writer threads (oracle/patches/gpu_worker.cpp):
 *   PairEndProcessor::readerTask      src/peprocessor.cpp:725-888   } raw file bytes instead of
 *   PairEndProcessor::interleavedReaderTask  src/peprocessor.cpp:890-1013 (config.interleaved)
 *   SingleEndProcessor::readerTask    src/seprocessor.cpp:327-442   } FastqReader::read
 *   FastqReader::getLine / read       src/fastqreader.cpp:240-368   } (src/fastqreader.cpp:88-149 fills 8 MiB blocks)
 *   processorTask -> processPairEnd / processSingleEnd  src/peprocessor.cpp:1021-1033, :362-708
 *   the output strings the loop body hands to WriterThread::input  src/peprocessor.cpp:652-686,
 *   src/seprocessor.cpp:299-304  (Read::appendToString src/read.cpp:119-154)
 *   WriterThread::inputPwrite's per-pack gzip members  src/writerthread.cpp:118-168
 *   FilterResult::addAdapterTrimmed  src/filterresult.cpp:124-180 (replayed on the host from the parsed text)
 *
 *   BgzfMtReader (bgzip-written ".gz" inputs)  src/bgzf.h:36-239  } fastp_gpu_inflate_bgzf on the compressed bytes
 *
 * What runs where: the host reads raw chunks of the input files into page-locked memory (a small pool of
 * positional reads), copies them to HBM and gets text back; line splitting + packing
 * (fastp_gpu_parse_fastq), the worker loop (fastp_gpu_submit_device), record formatting for every output
 * stream (fastp_gpu_format_streams) and, for ".gz" outputs, the gzip members (fastp_gpu_deflate_bgzf) run
 * on the device.  File reading, the device work and file writing of neighbouring chunks overlap.
 * The result - every output stream byte for byte, the counter block, the adapter maps - is what ONE
 * worker thread of the reference produces from the same files (`-w 1`: Duplicate and the
 * overrepresentation sampling see the reads in file order, insert sizes are taken from every pair).
 *
 * A read longer than the context's max_len does not stop the run (the reference grows its buffers,
 * Stats::extendBuffer src/stats.cpp:65-83): the stream finishes the chunks before it, carries counters,
 * duplicate bitmaps and stream positions into a context with a larger max_len and goes on ("re-plan").
 * A malformed record ends the stream where FastqReader::read would return NULL (src/fastqreader.cpp:338-362):
 * the records before it are processed, *truncated is set.
 */
#ifndef FASTP_GPU_STREAM_H
#define FASTP_GPU_STREAM_H

#include "fastp_gpu.h"
#include "fastp_gpu_host.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one chunk's bytes of one output stream, in stream order; called from ONE thread of the stream (its writer
 * thread), for every wanted stream without a file descriptor, also with len == 0 (a WriterThread takes one
 * string per pack and thread in turn).  Return 0, or non-zero to stop the run (FASTP_GPU_E_INVALID). */
typedef int (*fastp_gpu_stream_emit_fn)(void* user, int stream, const char* data, int64_t len);

typedef struct fastp_gpu_stream_config {
    const char* in1;            /* FASTQ file: a regular file (read with pread, several pieces at a time) or a pipe /
                                 * FIFO / "/dev/stdin" (--stdin: read in sequence).  A name that ends in ".gz" is a
                                 * gzip stream, as for FastqReader::init (src/fastqreader.cpp:169-199): a bgzip-written
                                 * one (isBgzf, src/bgzf.h:17-27) goes to the device compressed and is inflated there
                                 * (fastp_gpu_bgzf_index + fastp_gpu_inflate_bgzf in place of BgzfMtReader), any other
                                 * is inflated on the host, several threads per file (fq_pgunzip.h; a pipe: one thread,
                                 * fq_gunzip.h; readToBufIgzip :88-149 has ISA-L on the reference's reader thread)   */
    const char* in2;            /* second file of a paired run, NULL for single-end                        */
    int64_t chunk_bytes;        /* text per file and trip; 0 = 16 MiB (FASTP_GPU_STREAM_CHUNK_MB)          */
    int32_t io_threads;         /* positional reads / writes in flight; 0 = 8 (FASTP_GPU_STREAM_IO_THREADS) */
    int32_t device;             /* HIP device ordinal                                                      */
    int64_t reads_to_process;   /* --reads_to_process (src/peprocessor.cpp:775-778): 0 = all               */
    fastp_gpu_format_options format;        /* --failed_out / --unpaired1/2 given, UMI name edit           */
    int32_t want[FASTP_GPU_N_OUTPUTS];      /* the stream is produced (out1/out2: the run has outputs)     */
    int32_t compress[FASTP_GPU_N_OUTPUTS];  /* its bytes are BGZF gzip members made on the device           */
    int32_t out_fd[FASTP_GPU_N_OUTPUTS];    /* >= 0: written with pwrite from out_offset on; -1: emit()     */
    int64_t out_offset[FASTP_GPU_N_OUTPUTS];
    fastp_gpu_stream_emit_fn emit;
    void* user;
    fastp_gpu_host* host;       /* FilterResult's adapter maps are replayed into this object (may be NULL) */
    int32_t interleaved;        /* --interleaved_in: a paired run whose mates alternate in in1 (in2 = NULL), as
                                 * FastqReaderPair::read takes them (src/fastqreader.cpp:470-478) in
                                 * PairEndProcessor::interleavedReaderTask (src/peprocessor.cpp:890-1013)          */
    int32_t phred64;            /* --phred64: quality characters are converted as FastqReader::read does
                                 * (fastp_gpu_phred64_to_33 after the parser)                                        */
    int32_t want_overlapped;    /* --overlapped_out (params->overlapped_out): its stream - the bases of read 1 the reference
                                 * prints for an overlapped pair, src/peprocessor.cpp:488-495 - is assembled on the host from
                                 * the records and the chunk's text and handed to emit() as stream FASTP_GPU_OVERLAPPED (one
                                 * call per chunk, in order, also with len == 0); emit must be given                     */
} fastp_gpu_stream_config;

typedef struct fastp_gpu_stream_stats {
    int64_t units;              /* reads (SE) / pairs (PE) processed                                       */
    int64_t chunks;
    int64_t replans;            /* contexts re-created for a longer read                                   */
    int32_t max_len;            /* the last context's max_len                                              */
    int32_t truncated;          /* 1 = a malformed record ended the stream early (units before it are out) */
    int64_t bytes_in[2];
    int64_t bytes_out[FASTP_GPU_N_OUTPUTS];   /* as written (compressed size for compressed streams)       */
    double wall_s, setup_s, wait_read_s, parse_s, engine_s, format_s, deflate_s, d2h_s, wait_write_s, write_s, replay_s;
    double inflate_s;           /* BGZF inputs: the device inflate + the text's copy to the host                */
    int64_t bytes_file[2];      /* bytes read from each input file (bytes_in counts TEXT)                        */
    int32_t input_kind[2];      /* 0 plain text, 1 gzip inflated on the host, 2 BGZF inflated on the device      */
    int64_t bytes_overlapped;   /* bytes handed to emit() for --overlapped_out's stream                          */
} fastp_gpu_stream_stats;

typedef struct fastp_gpu_stream fastp_gpu_stream;

/* params as for fastp_gpu_create (max_len = the expected read length: the stream raises it itself when the
 * first chunk, or any later one, holds a longer read; strings are copied).  Creates the engine context. */
int fastp_gpu_stream_create(const fastp_gpu_params* params, const fastp_gpu_stream_config* cfg, fastp_gpu_stream** out);
/* the whole run: returns when every byte has been handed to its file descriptor / emit callback */
int fastp_gpu_stream_run(fastp_gpu_stream* s);
/* the run's counter block in the layout of the LAST context (cycles = its max_len; the blocks of the contexts a
 * re-plan replaced are folded in).  fastp_gpu_stream_layout first, then counters with n = layout.total. */
int fastp_gpu_stream_layout(const fastp_gpu_stream* s, fastp_gpu_counter_layout* out);
int fastp_gpu_stream_counters(fastp_gpu_stream* s, int64_t* out, int64_t n);
int fastp_gpu_stream_get_stats(const fastp_gpu_stream* s, fastp_gpu_stream_stats* out);
const char* fastp_gpu_stream_last_error(const fastp_gpu_stream* s);   /* s may be NULL: the creating thread's last error */
void fastp_gpu_stream_destroy(fastp_gpu_stream* s);

/* The host inflater the stream reads non-bgzip ".gz" inputs with (fastp_amd/csrc/fq_gunzip.h: what ISA-L's igzip is to
 * FastqReader::readToBufIgzip, src/fastqreader.cpp:88-149), on its own: the text of every member of the gzip file `path`
 * into out[0, capacity), taken from the inflater `piece` bytes at a time (0 = 1 MiB; the tests vary it to move the
 * hand-over points).  FASTP_GPU_E_INVALID for a damaged stream (bad code set, distance in front of the member, CRC-32 /
 * ISIZE mismatch, no gzip header behind a member, end of file inside one), FASTP_GPU_E_OVERFLOW if capacity is too small.
 * Needs no device. */
int fastp_gpu_stream_gunzip_file(const char* path, uint8_t* out, int64_t capacity, int64_t piece, int64_t* out_len);
/* The same through the inflater the stream uses on regular files by default (fastp_amd/csrc/fq_pgunzip.h): `threads` host
 * threads on the ONE gzip stream - block starts found by parsing candidate headers every `chunk_bytes` of compressed data
 * (0 = 2 MiB), chunks decoded with their 32 KiB window unknown (16-bit symbols, markers), windows and markers resolved in
 * order, a chunk whose predecessor did not end on its first bit thrown away; CRC-32 / ISIZE of every member checked from the
 * chunks' partial CRCs.  Same results and errors as fastp_gpu_stream_gunzip_file at any `threads` / `chunk_bytes` (the tests
 * use chunks of a few kilobytes so that small files cross many boundaries).  threads = 1, chunk_bytes = 0: fq_gunzip.h. */
int fastp_gpu_stream_gunzip_file_mt(const char* path, uint8_t* out, int64_t capacity, int64_t piece, int threads, int64_t chunk_bytes,
                                    int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* FASTP_GPU_STREAM_H */
