// gpu_worker.h - the hooks a GPU-enabled build of reference fastp calls (TEST INFRASTRUCTURE: this is the
// reference-side binding of include/fastp_gpu.h + include/fastp_gpu_host.h, compiled INTO the reference by
// oracle/build_ref_gpu.sh so that the drop-in boundary is exercised end to end: fastp's own CLI, reader and writer
// threads, Stats / FilterResult objects and JSON / HTML reporters around the engine).
//
// oracle/patches/apply_gpu_worker.py inserts one-line calls into copies of the reference's sources:
//   src/peprocessor.cpp  top of PairEndProcessor::readerTask       -> fastp_gpu_stream_reader_pe   (stream mode)
//   src/seprocessor.cpp  top of SingleEndProcessor::readerTask     -> fastp_gpu_stream_reader_se
//   src/peprocessor.cpp  top of PairEndProcessor::interleavedReaderTask -> fastp_gpu_stream_reader_interleaved
//   src/peprocessor.cpp  top of PairEndProcessor::processPairEnd   -> fastp_gpu_worker_pe
//   src/seprocessor.cpp  top of SingleEndProcessor::processSingleEnd -> fastp_gpu_worker_se
//   both                 end of ::processorTask (before setConsumerFinished) -> fastp_gpu_worker_drain_pe / _se
//   both                 before "merge stats" in ::process()        -> fastp_gpu_worker_finish_pe / _se
//   src/evaluator.cpp    top of Evaluator::computeOverRepSeq         -> fastp_gpu_worker_overrep
//   src/evaluator.cpp    in front of evalAdapterAndReadNum's counting loop -> fastp_gpu_worker_adapter_kmers
// The engine is used when the environment has FASTP_GPU=1; otherwise the hooks return "not handled" and the
// reference's own loop runs.
#ifndef FASTP_GPU_WORKER_H
#define FASTP_GPU_WORKER_H
#include <map>
#include <string>

class PairEndProcessor;
class SingleEndProcessor;
class ThreadConfig;
struct ReadPack;

// STREAM MODE - the hook at the top of readerTask (src/peprocessor.cpp:725, src/seprocessor.cpp:327): with FASTP_GPU=1 and
// plain FASTQ files the read-1 reader thread runs include/fastp_gpu_stream.h's loop over both files (raw chunks -> device
// parser -> worker loop -> device formatter -> the WriterThreads' files) and closes the input lists when it is done; the
// worker threads stay idle.  1 = the run was taken, -1 = the reference's reader runs and the worker hooks below see packs.
int fastp_gpu_stream_reader_pe(PairEndProcessor* p, bool isLeft);
int fastp_gpu_stream_reader_se(SingleEndProcessor* p);
int fastp_gpu_stream_reader_interleaved(PairEndProcessor* p);

// PACK MODE (FASTP_GPU_STREAM=0, or an option set the stream loop does not take: --overlapped_out, phred64, interleaved /
// piped / gz input).  1 = the pack was taken by the engine (packed into the current window of packs; its outputs reach the writers when the
// window's records arrive - gpu_worker.cpp); -1 = engine disabled: run the reference's own loop body on this pack.
// A read the packer refuses (letters outside ACGTN, quality characters outside '!'..'~', longer than the evaluated
// read length) stops the run with a message.
int fastp_gpu_worker_pe(PairEndProcessor* p, ReadPack* left, ReadPack* right, ThreadConfig* config);
int fastp_gpu_worker_se(SingleEndProcessor* p, ReadPack* pack, ThreadConfig* config);
// at the end of processorTask (before the consumer side of the input lists is closed and, for the last thread, the
// writers are told the input is complete): the packs this thread has handed to the engine and not yet got back
void fastp_gpu_worker_drain_pe(PairEndProcessor* p, ThreadConfig* config);
void fastp_gpu_worker_drain_se(SingleEndProcessor* p, ThreadConfig* config);
// in processorTask's idle branch (no pack to consume): a thread with nothing to do still hands out the packs whose
// records have arrived - the window slots are freed by the LAST thread to do so
void fastp_gpu_worker_idle_pe(PairEndProcessor* p, ThreadConfig* config);
void fastp_gpu_worker_idle_se(SingleEndProcessor* p, ThreadConfig* config);
// load the engine's counter block into the worker threads' Stats / FilterResult objects, Duplicate's totals and
// the insert-size histogram, right before the reference merges them and writes its reports
void fastp_gpu_worker_finish_pe(PairEndProcessor* p, ThreadConfig** configs);
void fastp_gpu_worker_finish_se(SingleEndProcessor* p, ThreadConfig** configs);
// Evaluator::computeOverRepSeq (-p) through fastp_gpu_eval_overrep: 1 = hotseqs filled by the engine, -1 = not handled
int fastp_gpu_worker_overrep(const std::string& filename, std::map<std::string, long>& hotseqs, int seqlen);

// Evaluator::evalAdapterAndReadNum's ten-mer histogram (evaluator.cpp:384-396) through fastp_gpu_eval_adapter_kmers:
// 1 = counts filled by the engine, -1 = not handled (the reference's own counting loop runs)
class Read;
class Evaluator;
int fastp_gpu_worker_adapter_kmers(Evaluator* ev, Read** reads, long records, int shiftTail, unsigned int* counts);

// FastqReader::getLine (fastqreader.cpp:240-262): with the engine on, the worker threads are no longer what bounds a run -
// the reader is, and most of its time goes into looking for the end of a line one character at a time.  The hook sits in
// front of that loop and returns the position the loop would stop at (the first '\r' or '\n' in [from, to), or `to`),
// found with memchr; the loop behind it then has nothing left to do.  FASTP_GPU off: returns `from`, the loop runs.
int fastp_gpu_reader_scan_eol(const char* buf, int from, int to);

// Duplicate::Duplicate (src/duplicate.cpp:46-52): the size of the bitmap the reference allocates and clears on the host.  With
// the engine on, Duplicate's bits live in HBM and the reference's own checkPair / checkRead never run: a token size then.
long fastp_gpu_worker_dup_bytes(long bytes);

#endif
