"""Python host mirror of the engine: thin ctypes layer over libfastp_gpu.so (the C ABI of
include/fastp_gpu.h).  No computation happens here, and there is NO CPU fallback: if the HIP
library is missing or no MI355X is visible, construction fails loudly."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_PKG = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_PKG, "libfastp_gpu.so")
_LIBS: dict[str, C.CDLL] = {}

EXPORTS = [
    "fastp_gpu_default_params", "fastp_gpu_seq_stride", "fastp_gpu_qual_stride", "fastp_gpu_cycles_for",
    "fastp_gpu_counter_layout_for", "fastp_gpu_counter_layout_for_params", "fastp_gpu_create", "fastp_gpu_destroy", "fastp_gpu_last_error",
    "fastp_gpu_pack_reads", "fastp_gpu_pack_reads_x", "fastp_gpu_bgzf_index", "fastp_gpu_inflate_bgzf", "fastp_gpu_parse_fastq", "fastp_gpu_parse_exotic", "fastp_gpu_phred64_to_33", "fastp_gpu_host_writes_overlapped", "fastp_gpu_format_fastq", "fastp_gpu_format_streams", "fastp_gpu_deflate_bgzf", "fastp_gpu_device_alloc", "fastp_gpu_device_free", "fastp_gpu_device_upload", "fastp_gpu_device_download", "fastp_gpu_eval_seq_len", "fastp_gpu_eval_adapter_kmers", "fastp_gpu_eval_overrep", "fastp_gpu_submit_host", "fastp_gpu_submit_device", "fastp_gpu_synchronize",
    "fastp_gpu_counters_device", "fastp_gpu_counters", "fastp_gpu_kernel_time",
    "fastp_gpu_counters_export", "fastp_gpu_counters_import",
    "fastp_gpu_dup_scan_bytes", "fastp_gpu_submit_pass1_device", "fastp_gpu_dup_bitmap_bytes", "fastp_gpu_dup_bitmap_export",
    "fastp_gpu_dup_prefix_set", "fastp_gpu_prefix_or_images", "fastp_gpu_submit_pass2_device", "fastp_gpu_stream_set_origin", "fastp_gpu_overrep_device",
    "fastp_gpu_device", "fastp_gpu_reset", "fastp_gpu_plan",
    "fastp_gpu_host_alloc", "fastp_gpu_host_free", "fastp_gpu_submit_host_async", "fastp_gpu_wait", "fastp_gpu_poll",
    "fastp_gpu_comm_id", "fastp_gpu_comm_init", "fastp_gpu_comm_init_local", "fastp_gpu_comm_destroy", "fastp_gpu_allreduce",
    "fastp_gpu_exchange_dup_prefix", "fastp_gpu_comm_last_error", "fastp_gpu_dup_bitmap_import", "fastp_gpu_warmup",
    # include/fastp_gpu_host.h
    "fastp_gpu_host_create", "fastp_gpu_host_destroy", "fastp_gpu_host_apply", "fastp_gpu_host_output", "fastp_gpu_host_clear_outputs",
    "fastp_gpu_host_adapter_entries", "fastp_gpu_host_adapter_entry", "fastp_gpu_host_add_adapter", "fastp_gpu_host_add_adapter_pair",
    # include/fastp_gpu_stream.h
    "fastp_gpu_stream_create", "fastp_gpu_stream_run", "fastp_gpu_stream_layout", "fastp_gpu_stream_counters", "fastp_gpu_stream_get_stats",
    "fastp_gpu_stream_last_error", "fastp_gpu_stream_destroy", "fastp_gpu_stream_gunzip_file", "fastp_gpu_stream_gunzip_file_mt",
]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fastp_gpu error {code}: {msg}")
        self.code = code


def load_library(path: str | None = None) -> C.CDLL:
    path = path or os.environ.get("FASTP_GPU_LIB") or DEFAULT_LIB
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'`).  fastp_amd has no CPU fallback.")
    L = C.CDLL(path)
    L.fastp_gpu_default_params.argtypes = [C.POINTER(abi.Params), C.c_int, C.c_int]
    L.fastp_gpu_seq_stride.restype = C.c_size_t
    L.fastp_gpu_seq_stride.argtypes = [C.c_int]
    L.fastp_gpu_qual_stride.restype = C.c_size_t
    L.fastp_gpu_qual_stride.argtypes = [C.c_int]
    L.fastp_gpu_cycles_for.restype = C.c_int
    L.fastp_gpu_cycles_for.argtypes = [C.POINTER(abi.Params)]
    L.fastp_gpu_counter_layout_for.argtypes = [C.c_int, C.c_int, C.POINTER(abi.CounterLayout)]
    L.fastp_gpu_counter_layout_for_params.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.CounterLayout)]
    L.fastp_gpu_counter_layout_for_params.restype = None
    L.fastp_gpu_create.restype = C.c_int
    L.fastp_gpu_create.argtypes = [C.POINTER(abi.Params), C.c_int, C.POINTER(C.c_void_p)]
    L.fastp_gpu_destroy.argtypes = [C.c_void_p]
    L.fastp_gpu_last_error.restype = C.c_char_p
    L.fastp_gpu_last_error.argtypes = [C.c_void_p]
    L.fastp_gpu_pack_reads.restype = C.c_int
    L.fastp_gpu_pack_reads.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.fastp_gpu_pack_reads_x.restype = C.c_int
    L.fastp_gpu_pack_reads_x.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
    L.fastp_gpu_submit_host.restype = C.c_int
    L.fastp_gpu_submit_host.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.POINTER(abi.Results)]
    L.fastp_gpu_submit_device.restype = C.c_int
    L.fastp_gpu_submit_device.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.POINTER(abi.Results), C.c_void_p]
    L.fastp_gpu_synchronize.restype = C.c_int
    L.fastp_gpu_synchronize.argtypes = [C.c_void_p]
    L.fastp_gpu_counters_device.restype = C.c_int
    L.fastp_gpu_counters_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p]
    L.fastp_gpu_counters.restype = C.c_int
    L.fastp_gpu_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    for fn in (L.fastp_gpu_counters_export, L.fastp_gpu_counters_import):
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.fastp_gpu_kernel_time.restype = C.c_int
    L.fastp_gpu_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    _LIBS[path] = L
    return L


def counter_layout(params: abi.Params, lib=None) -> abi.CounterLayout:
    L = lib or load_library()
    lay = abi.CounterLayout()
    L.fastp_gpu_counter_layout_for_params(C.byref(params), C.byref(lay))
    return lay


def pack_ascii(lib, max_len, seq, qual, lens, exotic=None):
    """ASCII rows [n, stride] -> packed rows via the library's host packer.  `exotic`: uint8 [n], gets 1 where a read
    holds a letter outside ACGTN (fastp_gpu_pack_reads_x); without it such a read is an error."""
    n = int(len(lens))
    ss, qs = int(lib.fastp_gpu_seq_stride(max_len)), int(lib.fastp_gpu_qual_stride(max_len))
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    lens32 = np.ascontiguousarray(lens, dtype=np.int32)
    stride = seq.shape[1] if n else 0
    sp = (seq.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
    qp = (qual.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(stride)).astype(np.uint64)
    so = np.zeros((n, ss), dtype=np.uint8)
    qo = np.zeros((n, qs), dtype=np.uint8)
    lo = np.zeros(n, dtype=np.uint16)
    bad = C.c_int32(-1)
    if exotic is not None:
        rc = lib.fastp_gpu_pack_reads_x(max_len, n, sp.ctypes.data, qp.ctypes.data, lens32.ctypes.data, so.ctypes.data,
                                        qo.ctypes.data, lo.ctypes.data, C.byref(bad), exotic.ctypes.data)
    else:
        rc = lib.fastp_gpu_pack_reads(max_len, n, sp.ctypes.data, qp.ctypes.data, lens32.ctypes.data, so.ctypes.data,
                                      qo.ctypes.data, lo.ctypes.data, C.byref(bad))
    if rc != 0:
        raise EngineError(rc, f"fastp_gpu_pack_reads failed at read {bad.value}")
    return so, qo, lo


class GpuEngine:
    """One fastp run on one GPU: Stats x4 + FilterResult + Duplicate + insert-size histogram live in HBM."""

    def __init__(self, params: abi.Params, device: int = 0, lib_path: str | None = None):
        self.lib = load_library(lib_path)
        self.params = params
        h = C.c_void_p()
        rc = self.lib.fastp_gpu_create(C.byref(params), device, C.byref(h))
        if rc != 0:
            raise EngineError(rc, (self.lib.fastp_gpu_last_error(None) or b"").decode())
        self.h = h
        self.layout = counter_layout(params, self.lib)

    def close(self):
        if getattr(self, "h", None):
            self.lib.fastp_gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, (self.lib.fastp_gpu_last_error(self.h) or b"").decode())

    # -- packed host buffers -> results (H2D + kernels + D2H) ---------------------------------
    def submit_packed(self, s1, q1, l1, s2=None, q2=None, l2=None, flags=abi.BATCH_STAT_ISIZE, corr_capacity=None, exotic=None):
        # exotic: (units int32 ascending, [text1, text2], [off1, off2]) - the raw sequence bytes of the units with letters
        # outside ACGTN (fastp_gpu_batch.exotic_*)
        n = int(len(l1))
        paired = s2 is not None
        r1 = np.zeros(n, dtype=abi.READ_RESULT_DTYPE)
        r2 = np.zeros(n if paired else 0, dtype=abi.READ_RESULT_DTYPE)
        pr = np.zeros(n if paired else 0, dtype=abi.PAIR_RESULT_DTYPE)
        if corr_capacity is None:
            corr_capacity = getattr(self, "corr_capacity", None)   # (tests: a caller's list that is too small)
        if corr_capacity is None:
            corr_capacity = max(1024, n * 32)
        corr = np.zeros(corr_capacity, dtype=abi.CORRECTION_DTYPE)
        ncorr = C.c_int32(0)
        b = abi.Batch()
        b.n, b.flags = n, flags
        b.seq1, b.qual1, b.len1 = s1.ctypes.data, q1.ctypes.data, l1.ctypes.data
        if paired:
            b.seq2, b.qual2, b.len2 = s2.ctypes.data, q2.ctypes.data, l2.ctypes.data
        if exotic is not None and len(exotic[0]):
            xu, xt, xo = exotic
            b.n_exotic = len(xu)
            b.exotic_unit = xu.ctypes.data
            for m in range(2 if paired else 1):
                b.exotic_text[m] = xt[m].ctypes.data
                b.exotic_off[m] = xo[m].ctypes.data
                b.exotic_text_bytes[m] = xt[m].nbytes
        res = abi.Results()
        res.r1 = r1.ctypes.data
        res.r2 = r2.ctypes.data if paired else None
        res.pair = pr.ctypes.data if paired else None
        res.corrections = corr.ctypes.data
        res.corrections_capacity = corr_capacity
        res.n_corrections = C.addressof(ncorr)
        nfasta = int(self.params.n_adapter_fasta)
        ev = np.zeros(max(16, n * 2 * min(nfasta, 8)) if nfasta else 0, dtype=abi.ADAPTER_EVENT_DTYPE)
        nev = C.c_int32(0)
        if nfasta:
            res.adapter_events = ev.ctypes.data
            res.adapter_events_capacity = len(ev)
            res.n_adapter_events = C.addressof(nev)
        self._check(self.lib.fastp_gpu_submit_host(self.h, C.byref(b), C.byref(res)))
        # --adapter_fasta trims of this batch, per read in adapter order (the device emits them unordered)
        self.last_adapter_events = np.sort(ev[:nev.value].copy(), order=["read", "adapter"])
        return r1, (r2 if paired else None), (pr if paired else None), corr[:min(ncorr.value, corr_capacity)].copy()

    # -- ASCII rows (what the FASTQ decoder produces) -> results -------------------------------
    def process(self, seq1, qual1, len1, seq2=None, qual2=None, len2=None, flags=abi.BATCH_STAT_ISIZE):
        ml = self.params.max_len
        n = int(len(len1))
        mask = np.zeros(n, dtype=np.uint8)
        s1, q1, l1 = pack_ascii(self.lib, ml, seq1, qual1, len1, mask)
        s2 = q2 = l2 = None
        if seq2 is not None:
            s2, q2, l2 = pack_ascii(self.lib, ml, seq2, qual2, len2, mask)
        exotic = None
        if mask.any():   # letters outside ACGTN: the units' raw rows travel with the batch
            xu = np.flatnonzero(mask).astype(np.int32)
            rows = [np.ascontiguousarray(np.asarray(a, dtype=np.uint8)[xu]) for a in ((seq1, seq2) if seq2 is not None else (seq1,))]
            offs = [(np.arange(len(xu), dtype=np.uint32) * np.uint32(r.shape[1])) for r in rows]
            exotic = (xu, rows, offs)
        if seq2 is not None:
            return self.submit_packed(s1, q1, l1, s2, q2, l2, flags, exotic=exotic)
        return self.submit_packed(s1, q1, l1, flags=flags, exotic=exotic)

    # -- device-resident batches (bench / multi-GPU hosts) ---------------------------------------
    def submit_device(self, batch: abi.Batch, results: abi.Results, stream=None):
        self._check(self.lib.fastp_gpu_submit_device(self.h, C.byref(batch), C.byref(results), stream))

    # ---- sharded runs (include/fastp_gpu.h "Sharded runs") ----
    def dup_scan_bytes(self, n: int) -> int:
        fn = self.lib.fastp_gpu_dup_scan_bytes
        fn.restype, fn.argtypes = C.c_int64, [C.c_void_p, C.c_int32]
        return int(fn(self.h, n))

    def submit_pass1_device(self, batch: abi.Batch, scan_ptr: int, results, stream=None):
        fn = self.lib.fastp_gpu_submit_pass1_device
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.c_void_p, C.POINTER(abi.Results), C.c_void_p]
        self._check(fn(self.h, C.byref(batch), scan_ptr, C.byref(results) if results is not None else None, stream))

    def submit_pass2_device(self, batch: abi.Batch, scan_ptr: int, results: abi.Results, stream=None):
        fn = self.lib.fastp_gpu_submit_pass2_device
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.POINTER(abi.Batch), C.c_void_p, C.POINTER(abi.Results), C.c_void_p]
        self._check(fn(self.h, C.byref(batch), scan_ptr, C.byref(results), stream))

    def prefix_or_images(self, images_ptr: int, n_images: int, bytes_each: int):
        fn = self.lib.fastp_gpu_prefix_or_images
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64]
        self._check(fn(self.h, images_ptr, n_images, bytes_each))

    def dup_bitmap_bytes(self) -> int:
        fn = self.lib.fastp_gpu_dup_bitmap_bytes
        fn.restype, fn.argtypes = C.c_int64, [C.c_void_p]
        return int(fn(self.h))

    def dup_bitmap_export(self, dst_ptr: int):
        fn = self.lib.fastp_gpu_dup_bitmap_export
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p]
        self._check(fn(self.h, dst_ptr))

    def dup_prefix_set(self, images_ptr, n_images: int):
        fn = self.lib.fastp_gpu_dup_prefix_set
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]
        self._check(fn(self.h, images_ptr, n_images))

    def stream_set_origin(self, units_before: int, post_reads_before: int):
        fn = self.lib.fastp_gpu_stream_set_origin
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int64, C.c_int64]
        self._check(fn(self.h, units_before, post_reads_before))

    def overrep_device(self, batch: abi.Batch, results: abi.Results, stream=None):
        fn = self.lib.fastp_gpu_overrep_device
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(abi.Batch), C.POINTER(abi.Results), C.c_void_p]
        self._check(fn(self.h, C.byref(batch), C.byref(results), stream))

    # ---- BGZF input (include/fastp_gpu.h) ----
    def bgzf_index(self, host_bytes: np.ndarray, max_blocks: int, max_text_bytes: int, check=True):
        """header walk over HOST bytes -> (info, pay_off, pay_len, isize, crc, out_off) numpy arrays"""
        fn = self.lib.fastp_gpu_bgzf_index
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64] + [C.c_void_p] * 5 + [C.POINTER(abi.InflateInfo)]
        arrs = [np.zeros(max(1, max_blocks), dtype=np.uint32) for _ in range(4)] + [np.zeros(max(1, max_blocks), dtype=np.uint64)]
        info = abi.InflateInfo()
        rc = fn(host_bytes.ctypes.data, host_bytes.size, max_blocks, max_text_bytes, *[a.ctypes.data for a in arrs], C.byref(info))
        if check and rc != 0:
            raise EngineError(rc, f"not a BGZF member at block {info.first_bad}")
        info.rc = rc
        return (info,) + tuple(a[:info.n_blocks] for a in arrs)

    def inflate_bgzf(self, comp_ptr, n_blocks, pay_off_ptr, pay_len_ptr, isize_ptr, crc_ptr, out_off_ptr, out_ptr, out_cap,
                     check_crc=True, check=True):
        fn = self.lib.fastp_gpu_inflate_bgzf
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_int32)]
        bad = C.c_int32(-1)
        rc = fn(self.h, comp_ptr, n_blocks, pay_off_ptr, pay_len_ptr, isize_ptr, crc_ptr, out_off_ptr, out_ptr, out_cap,
                int(check_crc), C.byref(bad))
        if check:
            self._check(rc)
        return rc, int(bad.value)

    def parse_fastq(self, text_ptr: int, nbytes: int, is_last: bool, max_records: int, seq_ptr: int, qual_ptr: int,
                    len_ptr: int, line_off_ptr: int, line_len_ptr: int, check=True) -> abi.ParseInfo:
        """FASTQ text resident on the device -> packed rows on the device (fastp_gpu_parse_fastq)"""
        info = abi.ParseInfo()
        fn = self.lib.fastp_gpu_parse_fastq
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                       C.c_void_p, C.c_void_p, C.POINTER(abi.ParseInfo)]
        rc = fn(self.h, text_ptr, nbytes, int(is_last), max_records, seq_ptr, qual_ptr, len_ptr, line_off_ptr,
                line_len_ptr, C.byref(info))
        if check:
            self._check(rc)
        info.rc = rc
        return info

    def parse_exotic(self) -> np.ndarray:
        """records of the last parse_fastq call with letters outside ACGTN (ascending), fastp_gpu_parse_exotic"""
        fn = self.lib.fastp_gpu_parse_exotic
        fn.restype = C.c_int32
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        n = int(fn(self.h, None, 0))
        out = np.zeros(max(n, 1), dtype=np.int32)
        fn(self.h, out.ctypes.data, n)
        return out[:n]

    def format_fastq(self, n, m1: abi.FormatIn, m2, corrections_ptr, n_corrections_ptr, out1_ptr, out1_cap, out2_ptr,
                     out2_cap, check=True):
        """result records + parsed text on the device -> out1 / out2 FASTQ text on the device"""
        fn = self.lib.fastp_gpu_format_fastq
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int32, C.POINTER(abi.FormatIn), C.POINTER(abi.FormatIn), C.c_void_p, C.c_void_p,
                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        lens = (C.c_int64 * 2)()
        rc = fn(self.h, n, C.byref(m1), C.byref(m2) if m2 is not None else None, corrections_ptr, n_corrections_ptr,
                out1_ptr, out1_cap, out2_ptr, out2_cap, lens)
        if check:
            self._check(rc)
        return rc, int(lens[0]), int(lens[1])

    def format_streams(self, n, m1: abi.FormatIn, m2, pair_ptr, corrections_ptr, n_corrections_ptr,
                       opts: abi.FormatOptions | None, out_ptrs, out_caps, check=True):
        """every output stream of the worker loop on the device (out1, out2, failed, merged, unpaired1, unpaired2);
        corrections are patched into the mates' text.  Returns (rc, [needed bytes per stream])"""
        fn = self.lib.fastp_gpu_format_streams
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_int32, C.POINTER(abi.FormatIn), C.POINTER(abi.FormatIn), C.c_void_p, C.c_void_p,
                       C.c_void_p, C.POINTER(abi.FormatOptions), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                       C.POINTER(C.c_int64)]
        outs = (C.c_void_p * abi.N_OUTPUTS)(*[p or None for p in out_ptrs])
        caps = (C.c_int64 * abi.N_OUTPUTS)(*out_caps)
        lens = (C.c_int64 * abi.N_OUTPUTS)()
        rc = fn(self.h, n, C.byref(m1), C.byref(m2) if m2 is not None else None, pair_ptr, corrections_ptr,
                n_corrections_ptr, C.byref(opts) if opts is not None else None, outs, caps, lens)
        if check:
            self._check(rc)
        return rc, [int(x) for x in lens]

    def deflate_bgzf(self, text_ptr, nbytes, out_ptr, out_capacity, write_eof=False, check=True):
        """device text -> BGZF-framed gzip members on the device; returns (rc, bytes written / needed)"""
        fn = self.lib.fastp_gpu_deflate_bgzf
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        n = C.c_int64(0)
        rc = fn(self.h, text_ptr, nbytes, int(write_eof), out_ptr, out_capacity, C.byref(n))
        if check:
            self._check(rc)
        return rc, int(n.value)

    # ---- the Evaluator pre-pass on the device (csrc/fq_eval.h); device pointers of one mate's packed rows ----
    def eval_seq_len(self, len_ptr, n):
        fn = self.lib.fastp_gpu_eval_seq_len
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        out = C.c_int32(0)
        self._check(fn(self.h, len_ptr, n, C.byref(out)))
        return int(out.value)

    def eval_adapter_kmers(self, seq_ptr, qual_ptr, len_ptr, n, trim_tail1, counts_ptr):
        """4^10 ten-mer histogram into DEVICE uint32[1 << 20] at counts_ptr; returns the records used"""
        fn = self.lib.fastp_gpu_eval_adapter_kmers
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                       C.POINTER(C.c_int64)]
        rec = C.c_int64(0)
        self._check(fn(self.h, seq_ptr, qual_ptr, len_ptr, n, trim_tail1, counts_ptr, C.byref(rec)))
        return int(rec.value)

    def eval_overrep(self, seq_ptr, qual_ptr, len_ptr, n, seq_len, max_seqs=1 << 16, text_capacity=1 << 22, check=True):
        """Evaluator::computeOverRepSeq: returns (rc, [(sequence bytes, count)] in std::map order)"""
        fn = self.lib.fastp_gpu_eval_overrep
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int64,
                       C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)]
        text = C.create_string_buffer(max(1, text_capacity))
        off = (C.c_int64 * (max_seqs + 1))()
        cnt = (C.c_int64 * max(1, max_seqs))()
        ns = C.c_int32(0)
        rc = fn(self.h, seq_ptr, qual_ptr, len_ptr, n, seq_len, text, text_capacity, off, cnt, max_seqs, C.byref(ns))
        if check:
            self._check(rc)
        if rc != 0:
            return rc, int(ns.value)
        raw = text.raw
        return rc, [(raw[off[i]:off[i + 1]], int(cnt[i])) for i in range(ns.value)]

    def synchronize(self):
        self._check(self.lib.fastp_gpu_synchronize(self.h))

    def reset(self):
        """a new run on the same context (fresh Stats / FilterResult / Duplicate)"""
        self._check(self.lib.fastp_gpu_reset(self.h))
        self._shard_stream_done = (0, 0)   # multigpu.run_shard's running stream origin belongs to the run that just ended

    # ---- collectives behind the C ABI (RCCL; csrc/fq_comm.cpp) ----
    def _comm_check(self, rc):
        if rc != 0:
            self.lib.fastp_gpu_comm_last_error.restype = C.c_char_p
            raise EngineError(rc, (self.lib.fastp_gpu_comm_last_error() or b"").decode())

    def comm_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        self._comm_check(self.lib.fastp_gpu_comm_id(buf))
        return bytes(buf)

    def comm_init(self, comm_id: bytes, nranks: int, rank: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(comm_id)
        self.lib.fastp_gpu_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self._comm_check(self.lib.fastp_gpu_comm_init(self.h, buf, nranks, rank))

    def _ctx_array(self):
        arr = (C.c_void_p * 1)(self.h)
        return arr

    def allreduce(self):
        """Stats::merge / FilterResult::merge over the communicator (this process owns one rank)"""
        self.lib.fastp_gpu_allreduce.argtypes = [C.c_void_p, C.c_int]
        self._comm_check(self.lib.fastp_gpu_allreduce(self._ctx_array(), 1))

    def exchange_dup_prefix(self):
        self.lib.fastp_gpu_exchange_dup_prefix.argtypes = [C.c_void_p, C.c_int]
        self._comm_check(self.lib.fastp_gpu_exchange_dup_prefix(self._ctx_array(), 1))

    def counters(self):
        out = np.zeros(self.layout.total, dtype=np.int64)
        self._check(self.lib.fastp_gpu_counters(self.h, out.ctypes.data, out.size))
        return out

    def counters_device(self):
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.fastp_gpu_counters_device(self.h, C.byref(p), C.byref(n), None))
        return p.value, n.value

    def counters_export(self, dst_device_ptr: int):
        self._check(self.lib.fastp_gpu_counters_export(self.h, dst_device_ptr, self.layout.total))

    def counters_import(self, src_device_ptr: int):
        self._check(self.lib.fastp_gpu_counters_import(self.h, src_device_ptr, self.layout.total))

    def debug_phase_cycles(self):
        """cycles per phase of the fused kernel (context created with FASTP_GPU_PHASE_TIMING=1)"""
        out = (C.c_uint64 * 16)()
        fn = self.lib.fastp_gpu_debug_phase_cycles
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p]
        self._check(fn(self.h, out))
        return list(out)

    def plan(self) -> str:
        """which kernels run the worker loop for this context's options"""
        self.lib.fastp_gpu_plan.argtypes = [C.c_void_p]
        return {0: "fused", 1: "split", 2: "lane"}.get(int(self.lib.fastp_gpu_plan(self.h)), "?")

    def kernel_time(self):
        ms, k = C.c_double(), C.c_int64()
        self._check(self.lib.fastp_gpu_kernel_time(self.h, C.byref(ms), C.byref(k)))
        return ms.value, k.value
