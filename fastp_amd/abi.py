"""ctypes mirror of include/fastp_gpu.h (the C ABI of the engine).

Only declarations live here - no computation.  Used by the Python host mirror
(fastp_amd.engine), the tests and bench.py.
"""
import ctypes as C

ABI_VERSION = 4
MAX_READ_LEN = 512
MAX_ADAPTER_LEN = 256

OK = 0
E_INVALID, E_NO_DEVICE, E_HIP, E_ALPHABET, E_TOO_LONG, E_UNSUPPORTED, E_OVERFLOW, E_NOMEM = (
    -1, -2, -3, -4, -5, -6, -7, -8)

PASS_FILTER = 0
FAIL_POLY_X = 4
FAIL_OVERLAP = 8
FAIL_N_BASE = 12
FAIL_LENGTH = 16
FAIL_TOO_LONG = 17
FAIL_QUALITY = 20
FAIL_COMPLEXITY = 24
FAIL_ADAPTER_DIMER = 28
FILTER_RESULT_TYPES = 32

BATCH_STAT_ISIZE = 1
BATCH_DEFER_OVERREP = 2

RF_NULL, RF_DUP, RF_ADAPTER, RF_ADAPTER_OV, RF_CORRECTED, RF_MERGED, RF_POLYX = (
    0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40)
PF_OVERLAPPED, PF_HAS_GAP, PF_ISIZE = 0x1, 0x2, 0x4

STATS_PRE1, STATS_POST1, STATS_PRE2, STATS_POST2 = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("paired", C.c_int32), ("max_len", C.c_int32),
        ("trim_front1", C.c_int32), ("trim_tail1", C.c_int32),
        ("trim_front2", C.c_int32), ("trim_tail2", C.c_int32),
        ("max_len1", C.c_int32), ("max_len2", C.c_int32),
        ("cut_front", C.c_int32), ("cut_tail", C.c_int32), ("cut_right", C.c_int32),
        ("cut_front_window", C.c_int32), ("cut_front_quality", C.c_int32),
        ("cut_tail_window", C.c_int32), ("cut_tail_quality", C.c_int32),
        ("cut_right_window", C.c_int32), ("cut_right_quality", C.c_int32),
        ("poly_g", C.c_int32), ("poly_g_min_len", C.c_int32),
        ("poly_x", C.c_int32), ("poly_x_min_len", C.c_int32),
        ("adapter_enabled", C.c_int32), ("allow_gap_overlap_trimming", C.c_int32),
        ("dimer_max_len", C.c_int32),
        ("adapter_seq_r1", C.c_char_p), ("adapter_seq_r2", C.c_char_p),
        ("correction", C.c_int32), ("merge", C.c_int32), ("merge_include_unmerged", C.c_int32),
        ("overlap_require", C.c_int32), ("overlap_diff_limit", C.c_int32),
        ("overlap_diff_percent_limit", C.c_int32),
        ("qual_filter", C.c_int32), ("qualified_qual", C.c_int32),
        ("unqualified_percent_limit", C.c_int32), ("n_base_limit", C.c_int32),
        ("avg_qual_req", C.c_int32),
        ("length_filter", C.c_int32), ("length_required", C.c_int32), ("length_limit", C.c_int32),
        ("complexity_filter", C.c_int32), ("complexity_threshold", C.c_double),
        ("dup_enabled", C.c_int32), ("dedup", C.c_int32), ("dup_accuracy_level", C.c_int32),
        ("insert_size_max", C.c_int32),
        ("umi_len1", C.c_int32), ("umi_len2", C.c_int32), ("umi_skip", C.c_int32),
        ("n_adapter_fasta", C.c_int32), ("adapter_fasta", C.POINTER(C.c_char_p)),
        ("overrep_enabled", C.c_int32), ("overrep_sampling", C.c_int32),
        ("eval_seq_len1", C.c_int32), ("eval_seq_len2", C.c_int32),
        ("n_overrep_seqs1", C.c_int32), ("n_overrep_seqs2", C.c_int32),
        ("overrep_seqs1", C.POINTER(C.c_char_p)), ("overrep_seqs2", C.POINTER(C.c_char_p)),
        ("overlapped_out", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


def default_params(paired, max_len):
    """Values of an un-flagged `fastp -i R1 [-I R2]` run (main.cpp:34-156 defaults;
    SURVEY.md 8b table).  polyG is OFF here: the reference only auto-enables it from
    the read-name prefix (evaluator.cpp:16-45), which is host logic."""
    p = Params()
    p.abi_version = ABI_VERSION
    p.paired = int(bool(paired))
    p.max_len = max_len
    p.cut_front_window = p.cut_tail_window = p.cut_right_window = 4
    p.cut_front_quality = p.cut_tail_quality = p.cut_right_quality = 20
    p.poly_g_min_len = 10
    p.poly_x_min_len = 10
    p.adapter_enabled = 1
    p.dimer_max_len = 2
    p.overlap_require = 30
    p.overlap_diff_limit = 5
    p.overlap_diff_percent_limit = 20
    p.qual_filter = 1
    p.qualified_qual = 15
    p.unqualified_percent_limit = 40
    p.n_base_limit = 5
    p.length_filter = 1
    p.length_required = 15
    p.complexity_threshold = 0.30
    p.dup_enabled = 1
    p.dup_accuracy_level = 1
    p.insert_size_max = 512
    return p


class Batch(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("flags", C.c_uint32),
        ("seq1", C.c_void_p), ("qual1", C.c_void_p), ("len1", C.c_void_p),
        ("seq2", C.c_void_p), ("qual2", C.c_void_p), ("len2", C.c_void_p),
        # units with letters outside ACGTN (ABI v4): see include/fastp_gpu.h
        ("n_exotic", C.c_int32), ("exotic_dense", C.c_int32), ("exotic_unit", C.c_void_p),
        ("exotic_text", C.c_void_p * 2), ("exotic_off", C.c_void_p * 2), ("exotic_text_bytes", C.c_int64 * 2),
    ]


class ReadResult(C.Structure):
    _fields_ = [
        ("front", C.c_uint16), ("len", C.c_uint16), ("code", C.c_uint8), ("flags", C.c_uint8),
        ("adapter_pos", C.c_int16), ("adapter_len", C.c_uint16), ("reserved", C.c_uint16),
    ]


class PairResult(C.Structure):
    _fields_ = [("ov_offset", C.c_int16), ("ov_len", C.c_uint16), ("ov_diff", C.c_uint16),
                ("flags", C.c_uint16)]


class Correction(C.Structure):
    _fields_ = [("read", C.c_uint32), ("pos", C.c_uint16), ("base", C.c_uint8), ("qual", C.c_uint8)]


class AdapterEvent(C.Structure):
    _fields_ = [("read", C.c_uint32), ("pos", C.c_int16), ("len", C.c_uint16), ("adapter", C.c_uint16),
                ("reserved", C.c_uint16)]


class Results(C.Structure):
    _fields_ = [
        ("r1", C.c_void_p), ("r2", C.c_void_p), ("pair", C.c_void_p),
        ("corrections", C.c_void_p), ("corrections_capacity", C.c_int32),
        ("n_corrections", C.c_void_p),
        ("adapter_events", C.c_void_p), ("adapter_events_capacity", C.c_int32),
        ("n_adapter_events", C.c_void_p),
    ]


def set_adapter_fasta(params, seqs):
    """attach the --adapter_fasta list (bytes objects) to a Params; the array is kept alive on it"""
    arr = (C.c_char_p * max(1, len(seqs)))(*seqs)
    params._fasta_keepalive = (arr, list(seqs))
    params.adapter_fasta = C.cast(arr, C.POINTER(C.c_char_p))
    params.n_adapter_fasta = len(seqs)
    return params


def set_overrep(params, seqs1, seqs2, eval_len1, eval_len2, sampling=20):
    """attach the overrepresentation-analysis seeds (Options::overRepSeqs1/2, sorted) to a Params"""
    a1 = (C.c_char_p * max(1, len(seqs1)))(*seqs1)
    a2 = (C.c_char_p * max(1, len(seqs2)))(*seqs2)
    params._overrep_keepalive = (a1, a2, list(seqs1), list(seqs2))
    params.overrep_enabled = 1
    params.overrep_sampling = sampling
    params.eval_seq_len1, params.eval_seq_len2 = eval_len1, eval_len2
    params.n_overrep_seqs1, params.n_overrep_seqs2 = len(seqs1), len(seqs2)
    params.overrep_seqs1 = C.cast(a1, C.POINTER(C.c_char_p))
    params.overrep_seqs2 = C.cast(a2, C.POINTER(C.c_char_p))
    return params


def overrep_lists(params):
    return ([params.overrep_seqs1[i] for i in range(params.n_overrep_seqs1)],
            [params.overrep_seqs2[i] for i in range(params.n_overrep_seqs2)])


def adapter_fasta_list(params):
    return [params.adapter_fasta[i] for i in range(params.n_adapter_fasta)]


class ParseInfo(C.Structure):
    _fields_ = [("n_records", C.c_int32), ("first_bad", C.c_int32), ("consumed", C.c_int64), ("n_lines", C.c_int64),
                ("bad_kind", C.c_int32), ("max_seq_len", C.c_int32), ("n_exotic", C.c_int32), ("reserved", C.c_int32)]


class InflateInfo(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("first_bad", C.c_int32), ("consumed", C.c_int64), ("out_bytes", C.c_int64)]


class FormatIn(C.Structure):
    _fields_ = [("text", C.c_void_p), ("line_off", C.c_void_p), ("line_len", C.c_void_p), ("res", C.c_void_p)]


class FormatOptions(C.Structure):
    _fields_ = [("want_failed", C.c_int32), ("want_unpaired1", C.c_int32), ("want_unpaired2", C.c_int32),
                ("umi_loc", C.c_int32), ("umi_len", C.c_int32), ("umi_prefix", C.c_char_p), ("umi_delimiter", C.c_char_p),
                ("corrections_capacity", C.c_int32)]


N_OUTPUTS = 6  # FASTP_GPU_OUT1, OUT2, FAILED, MERGED, UNPAIRED1, UNPAIRED2
OUT_OVERLAPPED = 6   # host glue only (FASTP_GPU_OVERLAPPED)
OVOUT_HIT = 0x8000   # fastp_gpu_read_result.reserved of read 1 with overlapped_out


class CounterLayout(C.Structure):
    _fields_ = [
        ("total", C.c_int64), ("cycles", C.c_int64),
        ("filter_stats", C.c_int64), ("adapter_reads", C.c_int64), ("adapter_bases", C.c_int64),
        ("polyx_reads", C.c_int64), ("polyx_bases", C.c_int64), ("correction", C.c_int64),
        ("corrected_reads", C.c_int64), ("merged_pairs", C.c_int64),
        ("dup_total", C.c_int64), ("dup_count", C.c_int64), ("isize", C.c_int64),
        ("stats", C.c_int64 * 4),
        ("st_reads", C.c_int64), ("st_length_sum", C.c_int64), ("st_qual_hist", C.c_int64),
        ("st_kmer", C.c_int64), ("st_cycle", C.c_int64), ("st_size", C.c_int64),
        ("n_overrep", C.c_int64 * 4), ("eval_len", C.c_int64 * 4),
        ("overrep_count", C.c_int64 * 4), ("overrep_dist", C.c_int64 * 4),
    ]


# numpy dtypes with the same memory layout as the result structs
import numpy as _np  # noqa: E402

READ_RESULT_DTYPE = _np.dtype([("front", "<u2"), ("len", "<u2"), ("code", "u1"), ("flags", "u1"),
                               ("adapter_pos", "<i2"), ("adapter_len", "<u2"), ("reserved", "<u2")])
PAIR_RESULT_DTYPE = _np.dtype([("ov_offset", "<i2"), ("ov_len", "<u2"), ("ov_diff", "<u2"),
                               ("flags", "<u2")])
CORRECTION_DTYPE = _np.dtype([("read", "<u4"), ("pos", "<u2"), ("base", "u1"), ("qual", "u1")])
ADAPTER_EVENT_DTYPE = _np.dtype([("read", "<u4"), ("pos", "<i2"), ("len", "<u2"), ("adapter", "<u2"),
                                 ("reserved", "<u2")])
assert ADAPTER_EVENT_DTYPE.itemsize == C.sizeof(AdapterEvent) == 12
assert READ_RESULT_DTYPE.itemsize == C.sizeof(ReadResult) == 12
assert PAIR_RESULT_DTYPE.itemsize == C.sizeof(PairResult) == 8
assert CORRECTION_DTYPE.itemsize == C.sizeof(Correction) == 8


def seq_stride(max_len):
    return ((max_len + 3) // 4 + 7) // 8 * 8


def qual_stride(max_len):
    return (max_len + 7) // 8 * 8


def cycles_for(params):
    """per-cycle capacity of the Stats slots (fastp_gpu_cycles_for)"""
    return 2 * params.max_len if params.merge else params.max_len
