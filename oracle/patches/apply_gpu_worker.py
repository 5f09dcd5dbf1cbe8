#!/usr/bin/env python3
"""Insert the GPU worker hooks into COPIES of two reference sources (TEST INFRASTRUCTURE, see gpu_worker.h).

    apply_gpu_worker.py <reference root> <output dir>

Writes <output dir>/peprocessor.cpp, seprocessor.cpp, evaluator.cpp and fastqreader.cpp: the reference's files with thirteen one-line insertions,
each placed by an anchor (the function signature / the comment that opens the merge of the per-thread results).
Nothing else of the reference is touched or reproduced here; every other source is compiled where it lies.
"""
import os
import re
import sys

ref, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def patch(name, inserts):
    src = open(os.path.join(ref, "src", name)).read()
    src = '#include "gpu_worker.h"\n' + src
    for anchor, text, where in inserts:
        if where == "before_each":   # the same insertion in front of every match
            ms = list(re.finditer(anchor, src))
            if not ms:
                sys.exit(f"apply_gpu_worker: anchor not found in {name}: {anchor}")
            for m in reversed(ms):
                src = src[:m.start()] + text + src[m.start():]
            continue
        m = re.search(anchor, src)
        if not m:
            sys.exit(f"apply_gpu_worker: anchor not found in {name}: {anchor}")
        pos = m.end() if where == "after" else m.start()
        src = src[:pos] + text + src[pos:]
    open(os.path.join(out, name), "w").write(src)


patch("peprocessor.cpp", [
    (r"void PairEndProcessor::readerTask\(bool isLeft\)\s*\{",
     "\n    if(fastp_gpu_stream_reader_pe(this, isLeft) > 0) return;   // GPU stream mode: raw chunks -> device parser / engine / formatter -> the writers' files\n", "after"),
    (r"void PairEndProcessor::interleavedReaderTask\(\)\s*\{",
     "\n    if(fastp_gpu_stream_reader_interleaved(this) > 0) return;   // GPU stream mode on --interleaved_in\n", "after"),
    (r"bool PairEndProcessor::processPairEnd\(ReadPack\* leftPack, ReadPack\* rightPack, ThreadConfig\* config\)\s*\{",
     "\n    if(fastp_gpu_worker_pe(this, leftPack, rightPack, config) > 0) return true;   // GPU engine (FASTP_GPU=1)\n", "after"),
    (r"[ \t]*// merge stats and filter results",
     "    fastp_gpu_worker_finish_pe(this, configs);   // engine counters -> Stats / FilterResult / Duplicate / insert sizes\n", "before"),
    (r"[ \t]*inputLeft->setConsumerFinished\(\);\s*inputRight->setConsumerFinished\(\);",
     "    fastp_gpu_worker_drain_pe(this, config);   // packs still in flight on the engine\n", "before"),
    (r"\} else \{(?=\s*std::unique_lock<std::mutex> lk\(mBackpressureMtx\);\s*mBackpressureCV\.wait_for\(lk, std::chrono::milliseconds\(1\)\);\s*\}\s*\}\s*fastp_gpu_worker_drain_pe)",
     "\n            fastp_gpu_worker_idle_pe(this, config);   // nothing to consume: hand out what has come back meanwhile", "after"),
])
patch("seprocessor.cpp", [
    (r"void SingleEndProcessor::readerTask\(\)\s*\{",
     "\n    if(fastp_gpu_stream_reader_se(this) > 0) return;   // GPU stream mode\n", "after"),
    (r"bool SingleEndProcessor::processSingleEnd\(ReadPack\* pack, ThreadConfig\* config\)\s*\{",
     "\n    if(fastp_gpu_worker_se(this, pack, config) > 0) return true;   // GPU engine (FASTP_GPU=1)\n", "after"),
    (r"[ \t]*// merge stats and read filter results",
     "    fastp_gpu_worker_finish_se(this, configs);   // engine counters -> Stats / FilterResult / Duplicate\n", "before"),
    (r"[ \t]*input->setConsumerFinished\(\);",
     "    fastp_gpu_worker_drain_se(this, config);   // packs still in flight on the engine\n", "before"),
    (r"\} else \{(?=\s*std::unique_lock<std::mutex> lk\(mBackpressureMtx\);\s*mBackpressureCV\.wait_for\(lk, std::chrono::milliseconds\(1\)\);\s*\}\s*\}\s*fastp_gpu_worker_drain_se)",
     "\n            fastp_gpu_worker_idle_se(this, config);   // nothing to consume: hand out what has come back meanwhile", "after"),
])
patch("evaluator.cpp", [
    (r"void Evaluator::computeOverRepSeq\(string filename, map<string, long>& hotseqs, int seqlen\)\s*\{",
     "\n    if(fastp_gpu_worker_overrep(filename, hotseqs, seqlen) > 0) return;   // counted on the device (FASTP_GPU=1)\n", "after"),
    (r"for\(int i=0; i<records; i\+\+\) \{\s*Read\* r = loadedReads\[i\];\s*const char\* data = r->mSeq->c_str\(\);\s*int key = -1;",
     "if(fastp_gpu_worker_adapter_kmers(this, loadedReads, records, shiftTail, counts) < 0)   // else: counted on the device (FASTP_GPU=1)\n    ", "before"),
])
patch("fastqreader.cpp", [
    # FastqReader::getLine's two scans for the end of the line (the loop stays and finds it has nothing to do)
    (r"[ \t]*while\(end < mBufDataLen\) \{\s*if\(mFastqBuf\[end\] != '\\r' && mFastqBuf\[end\] != '\\n'\)\s*end\+\+;",
     "\tend = fastp_gpu_reader_scan_eol(mFastqBuf, end, mBufDataLen);   // memchr instead of one character at a time (FASTP_GPU=1)\n", "before_each"),
])
patch("duplicate.cpp", [
    (r"[ \t]*mBufLenInBits = mBufLenInBytes << 3;",
     "    mBufLenInBytes = fastp_gpu_worker_dup_bytes(mBufLenInBytes);   // FASTP_GPU=1: Duplicate's bitmaps live in HBM, a token buffer here\n", "before"),
])
print("patched duplicate.cpp, peprocessor.cpp, seprocessor.cpp, evaluator.cpp, fastqreader.cpp ->", out)
