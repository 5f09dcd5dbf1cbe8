"""ctypes binding of include/fastp_gpu_host.h - the C++ host glue (fastp_amd/csrc/fq_glue.cpp):
lets the tests run the golden cases through the C++ string side instead of fastp_amd/hostloop.py."""
import ctypes as C

import numpy as np

from fastp_amd import abi, hostloop


class Reads(C.Structure):
    _fields_ = [("n", C.c_int32),
                ("name", C.POINTER(C.c_char_p)), ("name_len", C.POINTER(C.c_int32)),
                ("seq", C.POINTER(C.c_void_p)), ("qual", C.POINTER(C.c_void_p)), ("len", C.POINTER(C.c_int32)),
                ("strand", C.POINTER(C.c_char_p)), ("strand_len", C.POINTER(C.c_int32))]


class HostOptions(C.Structure):
    _fields_ = [("want_failed", C.c_int32), ("want_unpaired1", C.c_int32), ("want_unpaired2", C.c_int32),
                ("umi_loc", C.c_int32), ("umi_len", C.c_int32), ("umi_prefix", C.c_char_p),
                ("umi_delimiter", C.c_char_p)]


EXPORTS = ["fastp_gpu_host_create", "fastp_gpu_host_destroy", "fastp_gpu_host_apply", "fastp_gpu_host_output",
           "fastp_gpu_host_clear_outputs", "fastp_gpu_host_adapter_entries", "fastp_gpu_host_adapter_entry",
           "fastp_gpu_host_add_adapter", "fastp_gpu_host_add_adapter_pair"]
UMI_LOC = {"read1": 1, "read2": 2, "per_read": 3}


def _reads(b: hostloop.FastqBatch):
    n = b.n
    keep = []
    r = Reads()
    r.n = n
    names = (C.c_char_p * max(1, n))(*b.names)
    strands = (C.c_char_p * max(1, n))(*b.strands)
    nl = (C.c_int32 * max(1, n))(*[len(x) for x in b.names])
    sl = (C.c_int32 * max(1, n))(*[len(x) for x in b.strands])
    seq = np.ascontiguousarray(b.seq)
    qual = np.ascontiguousarray(b.qual)
    lens = np.ascontiguousarray(b.lens, dtype=np.int32)
    sp = (C.c_void_p * max(1, n))(*[seq.ctypes.data + i * seq.strides[0] for i in range(n)])
    qp = (C.c_void_p * max(1, n))(*[qual.ctypes.data + i * qual.strides[0] for i in range(n)])
    r.name, r.name_len = names, nl
    r.strand, r.strand_len = strands, sl
    r.seq, r.qual = sp, qp
    r.len = lens.ctypes.data_as(C.POINTER(C.c_int32))
    keep += [names, strands, nl, sl, seq, qual, lens, sp, qp]
    return r, keep


class CppHost:
    def __init__(self, lib, params: abi.Params, want_failed=True, want_unpaired=False, umi: hostloop.UmiNameEditor | None = None):
        self.lib = lib
        lib.fastp_gpu_host_output.restype = C.c_void_p
        lib.fastp_gpu_host_adapter_entries.restype = C.c_int64
        o = HostOptions()
        o.want_failed, o.want_unpaired1, o.want_unpaired2 = int(want_failed), int(want_unpaired), int(want_unpaired)
        if umi is not None:
            o.umi_loc, o.umi_len = UMI_LOC[umi.loc], umi.umi_len
            o.umi_prefix = umi.prefix or None
            o.umi_delimiter = umi.delimiter
        self.h = C.c_void_p()
        rc = lib.fastp_gpu_host_create(C.byref(params), C.byref(o), C.byref(self.h))
        assert rc == 0, rc
        self.params = params

    def close(self):
        if self.h:
            self.lib.fastp_gpu_host_destroy(self.h)
            self.h = None

    def apply(self, b1, b2, r1, r2, pair, corr, events):
        res = abi.Results()
        r1 = np.ascontiguousarray(r1)
        res.r1 = r1.ctypes.data
        keep = [r1]
        if b2 is not None:
            r2, pair = np.ascontiguousarray(r2), np.ascontiguousarray(pair)
            res.r2, res.pair = r2.ctypes.data, pair.ctypes.data
            keep += [r2, pair]
        corr = np.ascontiguousarray(corr if corr is not None else np.zeros(0, dtype=abi.CORRECTION_DTYPE))
        nc = C.c_int32(len(corr))
        res.corrections, res.corrections_capacity, res.n_corrections = corr.ctypes.data, len(corr), C.addressof(nc)
        ev = np.ascontiguousarray(events if events is not None else np.zeros(0, dtype=abi.ADAPTER_EVENT_DTYPE))
        ne = C.c_int32(len(ev))
        res.adapter_events, res.adapter_events_capacity, res.n_adapter_events = ev.ctypes.data, len(ev), C.addressof(ne)
        v1, k1 = _reads(b1)
        if b2 is not None:
            v2, k2 = _reads(b2)
            rc = self.lib.fastp_gpu_host_apply(self.h, C.byref(v1), C.byref(v2), C.byref(res))
        else:
            rc = self.lib.fastp_gpu_host_apply(self.h, C.byref(v1), None, C.byref(res))
        assert rc == 0, rc

    def output(self, which):
        n = C.c_size_t()
        p = self.lib.fastp_gpu_host_output(self.h, which, C.byref(n))
        return None if not p else C.string_at(p, n.value)

    def outputs(self, paired):
        o = hostloop.Outputs(paired, True, True, True)
        o.out1 = self.output(0) or b""
        o.out2 = self.output(1) if paired else None
        o.failed = self.output(2)
        o.merged = self.output(3) or b""
        o.unpaired1, o.unpaired2 = self.output(4), self.output(5)
        o.overlapped = self.output(abi.OUT_OVERLAPPED) or b""
        return o

    def adapter_maps(self):
        am = hostloop.AdapterMaps()
        for is_r2, d in ((0, am.a1), (1, am.a2)):
            for i in range(self.lib.fastp_gpu_host_adapter_entries(self.h, is_r2)):
                s, ln, cnt = C.c_void_p(), C.c_int32(), C.c_int64()
                rc = self.lib.fastp_gpu_host_adapter_entry(self.h, is_r2, C.c_int64(i), C.byref(s), C.byref(ln), C.byref(cnt))
                assert rc == 0
                d[C.string_at(s, ln.value)] = cnt.value
        return am
