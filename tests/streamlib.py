"""ctypes mirror of include/fastp_gpu_stream.h for the tests: FASTQ files -> output files through
fastp_gpu_stream_create / _run / _counters of a given library (the emulator build in the CPU suite, the
HIP library under -m gpu), the adapter maps through the host glue object the stream replays into."""
import ctypes as C
import os

import numpy as np

from fastp_amd import abi, hostloop
import cpphost
import refjson

N_OUT = 6
STREAM_NAMES = ["out1", "out2", "failed", "merged", "unpaired1", "unpaired2"]
EMIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)


class FormatOptions(C.Structure):
    _fields_ = [("want_failed", C.c_int32), ("want_unpaired1", C.c_int32), ("want_unpaired2", C.c_int32), ("umi_loc", C.c_int32),
                ("umi_len", C.c_int32), ("umi_prefix", C.c_char_p), ("umi_delimiter", C.c_char_p), ("corrections_capacity", C.c_int32)]


class StreamConfig(C.Structure):
    _fields_ = [("in1", C.c_char_p), ("in2", C.c_char_p), ("chunk_bytes", C.c_int64), ("io_threads", C.c_int32), ("device", C.c_int32),
                ("reads_to_process", C.c_int64), ("format", FormatOptions), ("want", C.c_int32 * N_OUT), ("compress", C.c_int32 * N_OUT),
                ("out_fd", C.c_int32 * N_OUT), ("out_offset", C.c_int64 * N_OUT), ("emit", EMIT_FN), ("user", C.c_void_p),
                ("host", C.c_void_p), ("interleaved", C.c_int32), ("phred64", C.c_int32), ("want_overlapped", C.c_int32)]


class StreamStats(C.Structure):
    _fields_ = [("units", C.c_int64), ("chunks", C.c_int64), ("replans", C.c_int64), ("max_len", C.c_int32), ("truncated", C.c_int32),
                ("bytes_in", C.c_int64 * 2), ("bytes_out", C.c_int64 * N_OUT)] + \
               [(k, C.c_double) for k in ("wall_s", "setup_s", "wait_read_s", "parse_s", "engine_s", "format_s", "deflate_s", "d2h_s",
                                          "wait_write_s", "write_s", "replay_s", "inflate_s")] + \
               [("bytes_file", C.c_int64 * 2), ("input_kind", C.c_int32 * 2), ("bytes_overlapped", C.c_int64)]


class StreamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fastp_gpu_stream error {code}: {msg}")
        self.code = code


def run_files(lib, params: abi.Params, in1: str, in2, outdir: str, want=("out1", "out2", "failed"), chunk_bytes=0, umi=None,
              compress=(), emit=False, reads_to_process=0, device=0, interleaved=False, phred64=False):
    """returns (outputs: dict name -> bytes, counters, layout, AdapterMaps, StreamStats)"""
    paired = bool(params.paired)
    lib.fastp_gpu_stream_last_error.restype = C.c_char_p
    lib.fastp_gpu_stream_last_error.argtypes = [C.c_void_p]
    host = cpphost.CppHost(lib, params, "failed" in want, "unpaired1" in want, umi)
    cfg = StreamConfig()
    cfg.in1 = in1.encode()
    cfg.in2 = in2.encode() if in2 else None
    cfg.chunk_bytes = chunk_bytes
    cfg.device = device
    cfg.reads_to_process = reads_to_process
    cfg.interleaved = int(interleaved)
    cfg.phred64 = int(phred64)
    cfg.format.want_failed = int("failed" in want)
    cfg.format.want_unpaired1 = int("unpaired1" in want)
    cfg.format.want_unpaired2 = int("unpaired2" in want)
    if umi is not None:
        cfg.format.umi_loc, cfg.format.umi_len = cpphost.UMI_LOC[umi.loc], umi.umi_len
        cfg.format.umi_prefix = umi.prefix or None
        cfg.format.umi_delimiter = umi.delimiter
    cfg.want_overlapped = int(bool(params.overlapped_out) and "overlapped" in want)
    fds, paths, collected = {}, {}, {q: bytearray() for q in range(N_OUT + 1)}
    for q, name in enumerate(STREAM_NAMES):
        cfg.out_fd[q] = -1
        if name not in want or (not paired and q in (1, 3, 4, 5)):
            continue
        cfg.want[q] = 1
        cfg.compress[q] = int(name in compress)
        if not emit:
            paths[q] = os.path.join(outdir, name + (".fq.gz" if name in compress else ".fq"))
            fds[q] = os.open(paths[q], os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            cfg.out_fd[q] = fds[q]

    def on_emit(user, stream, data, n):
        collected[stream] += C.string_at(data, n) if n else b""
        return 0
    cb = EMIT_FN(on_emit)
    cfg.emit = cb
    cfg.host = host.h
    s = C.c_void_p()
    lib.fastp_gpu_stream_create.argtypes = [C.POINTER(abi.Params), C.POINTER(StreamConfig), C.POINTER(C.c_void_p)]
    rc = lib.fastp_gpu_stream_create(C.byref(params), C.byref(cfg), C.byref(s))
    if rc != 0:
        host.close()
        for fd in fds.values():
            os.close(fd)
        raise StreamError(rc, (lib.fastp_gpu_stream_last_error(None) or b"").decode())
    try:
        lib.fastp_gpu_stream_run.argtypes = [C.c_void_p]
        rc = lib.fastp_gpu_stream_run(s)
        if rc != 0:
            raise StreamError(rc, (lib.fastp_gpu_stream_last_error(s) or b"").decode())
        lay = abi.CounterLayout()
        lib.fastp_gpu_stream_layout.argtypes = [C.c_void_p, C.POINTER(abi.CounterLayout)]
        assert lib.fastp_gpu_stream_layout(s, C.byref(lay)) == 0
        ctr = np.zeros(lay.total, dtype=np.int64)
        lib.fastp_gpu_stream_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        rc = lib.fastp_gpu_stream_counters(s, ctr.ctypes.data, lay.total)
        if rc != 0:
            raise StreamError(rc, (lib.fastp_gpu_stream_last_error(s) or b"").decode())
        st = StreamStats()
        lib.fastp_gpu_stream_get_stats.argtypes = [C.c_void_p, C.POINTER(StreamStats)]
        lib.fastp_gpu_stream_get_stats(s, C.byref(st))
        amaps = host.adapter_maps()
    finally:
        lib.fastp_gpu_stream_destroy.argtypes = [C.c_void_p]
        lib.fastp_gpu_stream_destroy(s)
        host.close()
        for fd in fds.values():
            os.close(fd)
    outs = {}
    for q, name in enumerate(STREAM_NAMES):
        if cfg.want[q]:
            outs[name] = bytes(collected[q]) if emit else open(paths[q], "rb").read()
    if cfg.want_overlapped:      # the host-assembled stream always arrives through the emit callback (stream 6)
        outs["overlapped"] = bytes(collected[N_OUT])
        assert st.bytes_overlapped == len(outs["overlapped"])
    return outs, ctr, lay, amaps, st


def as_outputs(outs: dict, paired: bool):
    o = hostloop.Outputs(paired, True, True, True)
    o.out1 = outs.get("out1", b"")
    o.out2 = outs.get("out2") if paired else None
    o.failed = outs.get("failed")
    o.merged = outs.get("merged", b"")
    o.unpaired1, o.unpaired2 = outs.get("unpaired1"), outs.get("unpaired2")
    if "overlapped" in outs:
        o.overlapped = outs["overlapped"]
    return o


def report(ctr, lay, params, amaps):
    return refjson.build(ctr, lay, params, amaps)
