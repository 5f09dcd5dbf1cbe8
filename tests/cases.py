"""Parity cases: the CLI flags given to the reference binary and the equivalent
engine parameter block (main.cpp:176-427 + options.cpp:85-446 derivations)."""
from fastp_amd import abi
import evalport


ADAPTER_R1 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
ADAPTER_R2 = "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"


def _pe(**kw):
    def f(max_len):
        p = abi.default_params(True, max_len)
        for k, v in kw.items():
            setattr(p, k, v)
        return p
    return f


def _se(**kw):
    def f(max_len):
        p = abi.default_params(False, max_len)
        p.adapter_seq_r1 = None
        for k, v in kw.items():
            setattr(p, k, v)
        return p
    return f


# name -> (paired, reference flags (besides -i/-I/-o/-O/-j/-h/-w 1), params factory, synth kwargs)
# -G pins polyG off (the reference would otherwise guess from read names); SE cases pin
# the adapter with -a or -A because SE auto-detection is Evaluator (host) logic.
CASES = {
    "pe_default": (True, ["-G"], _pe(), {}),
    "pe_cut_right": (True, ["-G", "--cut_right"], _pe(cut_right=1), {}),
    "pe_cut_front_tail": (True, ["-G", "--cut_front", "--cut_tail", "-W", "5", "-M", "18"],
                          _pe(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5,
                              cut_right_window=5, cut_front_quality=18, cut_tail_quality=18,
                              cut_right_quality=18), {}),
    "pe_polyg_polyx": (True, ["-g", "-x"], _pe(poly_g=1, poly_x=1),
                       {"polyg_frac": 0.2, "polyx_frac": 0.2}),
    "pe_adapter_seq": (True, ["-G", "-a", ADAPTER_R1, "--adapter_sequence_r2", ADAPTER_R2],
                       _pe(adapter_seq_r1=ADAPTER_R1.encode(), adapter_seq_r2=ADAPTER_R2.encode()),
                       {"insert_mean": 160.0}),
    "pe_correction": (True, ["-G", "-c"], _pe(correction=1), {"insert_mean": 200.0}),
    "pe_trim_fixed": (True, ["-G", "-f", "3", "-t", "2", "-F", "1", "-T", "4", "-b", "120", "-B", "100"],
                      _pe(trim_front1=3, trim_tail1=2, trim_front2=1, trim_tail2=4, max_len1=120, max_len2=100),
                      {}),
    "pe_filters": (True, ["-G", "-q", "20", "-u", "30", "-n", "2", "-e", "25", "-l", "40", "--length_limit",
                          "140", "-y", "-Y", "40", "--cut_tail"],
                   _pe(qualified_qual=20, unqualified_percent_limit=30, n_base_limit=2, avg_qual_req=25,
                       length_required=40, length_limit=140, complexity_filter=1, complexity_threshold=0.40,
                       cut_tail=1), {}),
    "pe_noadapter_dedup": (True, ["-G", "-A", "--dedup"],
                           _pe(adapter_enabled=0, dedup=1, dup_accuracy_level=3), {"dup_frac": 0.3}),
    "pe_nofilters": (True, ["-G", "-Q", "-L", "--dont_eval_duplication"],
                     _pe(qual_filter=0, length_filter=0, dup_enabled=0), {}),
    "pe_merge": (True, ["-G", "-m", "--merged_out", "@TMP@/merged.fq"], _pe(merge=1, correction=1),
                 {"insert_mean": 220.0}),
    "pe_merge_unmerged": (True, ["-G", "-m", "--include_unmerged", "--merged_out", "@TMP@/merged.fq"],
                          _pe(merge=1, correction=1, merge_include_unmerged=1), {"insert_mean": 260.0}),
    "pe_allow_gap": (True, ["-G", "--allow_gap_overlap_trimming", "-c"],
                     _pe(allow_gap_overlap_trimming=1, correction=1), {"insert_mean": 150.0}),
    "pe_umi_per_read": (True, ["-G", "-U", "--umi_loc", "per_read", "--umi_len", "6", "--umi_skip", "2"],
                        _pe(umi_len1=6, umi_len2=6, umi_skip=2), {}),
    "pe_overlap_knobs": (True, ["-G", "--overlap_len_require", "20", "--overlap_diff_limit", "8",
                                "--overlap_diff_percent_limit", "10", "--dimer_max_len", "40"],
                         _pe(overlap_require=20, overlap_diff_limit=8, overlap_diff_percent_limit=10,
                             dimer_max_len=40), {"insert_mean": 120.0, "insert_sd": 60.0}),
    "se_default_noadapter": (False, ["-G", "-A"], _se(adapter_enabled=0), {}),
    "se_adapter_cut": (False, ["-g", "-a", ADAPTER_R1, "--cut_right", "--cut_front"],
                       _se(adapter_seq_r1=ADAPTER_R1.encode(), poly_g=1, cut_right=1, cut_front=1),
                       {"insert_mean": 140.0, "polyg_frac": 0.15}),
    "se_umi_read1": (False, ["-G", "-A", "-U", "--umi_loc", "read1", "--umi_len", "8"],
                     _se(adapter_enabled=0, umi_len1=8), {}),
    "se_polyx_complexity": (False, ["-G", "-A", "-x", "-y", "--poly_x_min_len", "8"],
                            _se(adapter_enabled=0, poly_x=1, poly_x_min_len=8, complexity_filter=1),
                            {"polyx_frac": 0.3}),
}

# --overlapped_out (peprocessor.cpp:488-495): the third, exact analysis after adapter trimming
CASES["pe_overlapped_out"] = (True, ["-G", "--overlapped_out", "@TMP@/overlapped.fq"], _pe(overlapped_out=1),
                              {"insert_mean": 200.0})
CASES["pe_overlapped_out_trims"] = (True, ["-G", "-c", "-x", "-b", "120", "-B", "100", "--cut_front", "-f", "2",
                                           "--overlapped_out", "@TMP@/overlapped.fq"],
                                    _pe(overlapped_out=1, correction=1, poly_x=1, max_len1=120, max_len2=100, cut_front=1,
                                        trim_front1=2, trim_front2=2),
                                    {"insert_mean": 170.0, "polyx_frac": 0.2})
CASES["pe_overlapped_out_noadapter"] = (True, ["-G", "-A", "--overlapped_out", "@TMP@/overlapped.fq"],
                                        _pe(overlapped_out=1, adapter_enabled=0), {"insert_mean": 150.0, "insert_sd": 60.0})
# --overlapped_out together with --merge: the exact analysis (:488) comes before the polyX / max_len cuts, merge mode's
# own analysis (:518) after them; both want the records' reserved fields (the engine gives them to --overlapped_out
# and the merged part lengths follow from the pair record)
CASES["pe_merge_overlapped_out"] = (True, ["-G", "-m", "--merged_out", "@TMP@/merged.fq", "--overlapped_out", "@TMP@/overlapped.fq"],
                                    _pe(merge=1, correction=1, overlapped_out=1), {"insert_mean": 200.0})
CASES["pe_merge_overlapped_out_trims"] = (True, ["-G", "-m", "--include_unmerged", "--merged_out", "@TMP@/merged.fq", "-x", "-b", "120",
                                                 "-B", "100", "--cut_front", "-f", "2", "--overlapped_out", "@TMP@/overlapped.fq"],
                                          _pe(merge=1, correction=1, merge_include_unmerged=1, overlapped_out=1, poly_x=1, max_len1=120,
                                              max_len2=100, cut_front=1, trim_front1=2, trim_front2=2),
                                          {"insert_mean": 170.0, "polyx_frac": 0.2})
# letters outside ACGTN: soft-masked stretches and reads, IUPAC codes, '.' - the reference
# bins them by `base & 7`, hashes them as 13, complements a/c/g/t to T/G/C/A and the rest to N, compares raw bytes
# everywhere else; the engine runs those units through the text kernel (fq_text.h)
CASES["pe_exotic_default"] = (True, ["-G", "--cut_right", "--overlapped_out", "@TMP@/overlapped.fq"], _pe(cut_right=1, overlapped_out=1),
                              {"insert_mean": 190.0, "exotic_frac": 0.12})
CASES["pe_exotic_merge"] = (True, ["-G", "-m", "--include_unmerged", "--merged_out", "@TMP@/merged.fq", "-x", "-y", "--cut_front", "--cut_tail"],
                            _pe(merge=1, correction=1, merge_include_unmerged=1, poly_x=1, complexity_filter=1, cut_front=1, cut_tail=1),
                            {"insert_mean": 200.0, "polyx_frac": 0.15, "exotic_frac": 0.15})
CASES["pe_exotic_dedup_adapters"] = (True, ["-g", "--dedup", "-c", "--allow_gap_overlap_trimming", "--adapter_sequence", ADAPTER_R1,
                                            "--adapter_sequence_r2", ADAPTER_R2],
                                     _pe(dedup=1, dup_accuracy_level=3, correction=1, allow_gap_overlap_trimming=1, poly_g=1,
                                         adapter_seq_r1=ADAPTER_R1.encode(), adapter_seq_r2=ADAPTER_R2.encode()),
                                     {"insert_mean": 140.0, "insert_sd": 50.0, "polyg_frac": 0.1, "dup_frac": 0.3, "exotic_frac": 0.12})
CASES["se_exotic_adapter"] = (False, ["-G", "-a", ADAPTER_R1, "-x", "-y", "--cut_right", "-U", "--umi_loc", "read1", "--umi_len", "6"],
                              _se(adapter_seq_r1=ADAPTER_R1.encode(), poly_x=1, complexity_filter=1, cut_right=1, umi_len1=6),
                              {"insert_mean": 120.0, "polyx_frac": 0.2, "exotic_frac": 0.15})

# adapters longer than 64 bases (the engine's cap is FASTP_GPU_MAX_ADAPTER_LEN = 256): the read-through of the
# synthetic reads is adapter + poly-A, so these match over their whole length
LONG_R1 = ADAPTER_R1 + "A" * 70      # 103 bases
LONG_R2 = ADAPTER_R2 + "A" * 100     # 133 bases
CASES["se_adapter_long"] = (False, ["-G", "-a", LONG_R1], _se(adapter_seq_r1=LONG_R1.encode()), {"insert_mean": 90.0})
CASES["pe_adapter_long"] = (True, ["-G", "-a", LONG_R1, "--adapter_sequence_r2", LONG_R2],
                            _pe(adapter_seq_r1=LONG_R1.encode(), adapter_seq_r2=LONG_R2.encode()),
                            {"insert_mean": 100.0, "insert_sd": 50.0})
# a long adapter without the poly-A run (one-gap matches need the indel to matter), indel-bearing reads that start with it
LONG_MIXED = ADAPTER_R1 + "CTGACCTCAAGTCTGCACACGAGAAGGCTAGATCGTAGCTAGCTAGGATCCATCGATTTACGGCAAT" + ADAPTER_R2[::-1]   # 131 bases
CASES["se_adapter_long_indel"] = (False, ["-G", "-a", LONG_MIXED], _se(adapter_seq_r1=LONG_MIXED.encode()),
                                  {"gen": "adapter_indel", "adapters": (LONG_MIXED, LONG_MIXED)})

# inputs on which the reference's one-gap code ACCEPTS (synth.indel_overlap_pairs / adapter_indel_reads): the
# closed-form device versions of Matcher::diffWithOneInsertion / matchWithOneInsertion take their positive branch
CASES["pe_allow_gap_indel"] = (True, ["-G", "--allow_gap_overlap_trimming"], _pe(allow_gap_overlap_trimming=1),
                               {"gen": "indel_overlap"})
CASES["pe_allow_gap_indel_corr"] = (True, ["-G", "--allow_gap_overlap_trimming", "-c", "--overlap_len_require", "20",
                                           "--overlap_diff_limit", "7", "--overlap_diff_percent_limit", "30"],
                                    _pe(allow_gap_overlap_trimming=1, correction=1, overlap_require=20, overlap_diff_limit=7,
                                        overlap_diff_percent_limit=30), {"gen": "indel_overlap"})
CASES["se_adapter_indel"] = (False, ["-G", "-a", ADAPTER_R1], _se(adapter_seq_r1=ADAPTER_R1.encode()),
                             {"gen": "adapter_indel"})
CASES["pe_adapter_indel"] = (True, ["-G", "-a", ADAPTER_R1, "--adapter_sequence_r2", ADAPTER_R2],
                             _pe(adapter_seq_r1=ADAPTER_R1.encode(), adapter_seq_r2=ADAPTER_R2.encode()),
                             {"gen": "adapter_indel"})
# least number of one-gap ACCEPTS the oracle must report on 20 000 units of these inputs (tests assert it)
GAP_CASES = ("pe_allow_gap_indel", "pe_allow_gap_indel_corr", "se_adapter_indel", "pe_adapter_indel", "se_adapter_long_indel")

# --adapter_fasta: contigs as the FASTA file lists them; Options::loadFastaAdapters keeps them in
# contig-name order (std::map), >= 6 bp, de-duplicated (options.cpp:50-83)
FASTA_CONTIGS = [("a_truseq_r2", ADAPTER_R2), ("b_truseq_r1", ADAPTER_R1), ("c_nextera", "CTGTCTCTTATACACATCT"),
                 ("d_short", "AGATCGGA"), ("e_polya", "AAAAAAAAAAAAAAAA")]
FASTA_LIST = [seq.encode() for _, seq in sorted(FASTA_CONTIGS)]
FASTA_FILE = "".join(f">{n}\n{q}\n" for n, q in FASTA_CONTIGS).encode()


def _with_fasta(factory):
    def f(max_len):
        return abi.set_adapter_fasta(factory(max_len), FASTA_LIST)
    return f


CASES["pe_adapter_fasta"] = (True, ["-G", "--adapter_fasta", "@TMP@/adapters.fa"], _with_fasta(_pe()),
                             {"insert_mean": 150.0, "polyx_frac": 0.2})
CASES["se_adapter_fasta"] = (False, ["-G", "-a", ADAPTER_R1, "--adapter_fasta", "@TMP@/adapters.fa"],
                             _with_fasta(_se(adapter_seq_r1=ADAPTER_R1.encode())), {"insert_mean": 120.0, "polyx_frac": 0.3})
# overrepresentation analysis (-p, -P sampling): the seeds come from the Evaluator pre-pass over the
# input itself (host logic, tests/evalport.py), so the parameter block is completed
# by finalize_params() once the input is known
CASES["pe_overrep"] = (True, ["-G", "-p", "-P", "3"], _pe(), {"insert_mean": 90.0, "insert_sd": 30.0, "polyx_frac": 0.3})
CASES["se_overrep"] = (False, ["-G", "-A", "-p", "-P", "2", "--cut_right"], _se(adapter_enabled=0, cut_right=1),
                       {"insert_mean": 80.0, "insert_sd": 25.0})
CASES["pe_overrep_correction"] = (True, ["-G", "-p", "-P", "3", "-c"], _pe(correction=1),
                                  {"insert_mean": 110.0, "insert_sd": 30.0, "polyx_frac": 0.3, "lowq_site_rate": 0.08})
CASES["pe_overrep_merge"] = (True, ["-G", "-p", "-P", "2", "-m", "--merged_out", "@TMP@/merged.fq"], _pe(merge=1, correction=1),
                             {"insert_mean": 170.0, "insert_sd": 60.0, "polyx_frac": 0.3, "lowq_site_rate": 0.06})
CASES["pe_overrep_merge_unmerged"] = (True, ["-G", "-p", "-P", "2", "-m", "--include_unmerged", "--merged_out", "@TMP@/merged.fq", "--dedup"],
                                      _pe(merge=1, correction=1, merge_include_unmerged=1, dedup=1),
                                      {"insert_mean": 260.0, "insert_sd": 90.0, "polyx_frac": 0.3, "dup_frac": 0.3})
# -p on reads with letters outside ACGTN: the counting kernel takes those units' symbols from their text (a foreign byte equals
# no seed symbol; in a merged read a/c/g/t complement to real bases, util.h:16-33)
CASES["pe_exotic_overrep_merge"] = (True, ["-G", "-p", "-P", "2", "-m", "--include_unmerged", "--merged_out", "@TMP@/merged.fq"],
                                    _pe(merge=1, correction=1, merge_include_unmerged=1),
                                    {"insert_mean": 200.0, "insert_sd": 80.0, "polyx_frac": 0.3, "lowq_site_rate": 0.06, "exotic_frac": 0.2})
CASES["se_exotic_overrep"] = (False, ["-G", "-A", "-p", "-P", "2"], _se(adapter_enabled=0),
                              {"insert_mean": 110.0, "insert_sd": 30.0, "polyx_frac": 0.3, "exotic_frac": 0.2})
OVERREP = {"pe_overrep": 3, "se_overrep": 2, "pe_overrep_correction": 3, "pe_overrep_merge": 2, "pe_overrep_merge_unmerged": 2,
           "pe_exotic_overrep_merge": 2, "se_exotic_overrep": 2}
# reads LONGER than the length the reference evaluates from the first 1000 reads (Evaluator::computeSeqLen evaluator.cpp:54-76; it grows
# its buffers, Stats::extendBuffer stats.cpp:65-83, quirk ledger: a20): the first 1100 units are at most 100 bases, later ones 150
CASES["pe_late_long_reads"] = (True, ["-G", "--cut_right"], _pe(cut_right=1), {"gen": "late_long"})
CASES["se_late_long_reads"] = (False, ["-G", "-A", "-p", "-P", "2"], _se(adapter_enabled=0), {"gen": "late_long", "insert_mean": 80.0, "insert_sd": 25.0})
OVERREP["se_late_long_reads"] = 2   # the distance arrays are sized by the EVALUATED length (100): stats.cpp:279
N_PAIRS_OVERRIDE = {"pe_late_long_reads": 1700, "se_late_long_reads": 1700, "pe_overrep": 1500, "se_overrep": 1500, "pe_overrep_correction": 1500, "pe_overrep_merge": 1500,
                    "pe_overrep_merge_unmerged": 1500, "pe_exotic_overrep_merge": 1500, "se_exotic_overrep": 1500,
                    "pe_allow_gap_indel": 1500, "pe_allow_gap_indel_corr": 1500, "se_adapter_indel": 1500,
                    "pe_adapter_indel": 1500}   # golden input size (default 500)


class _ArrayBatch:
    def __init__(self, seq, lens):
        self.seq, self.lens, self.n = seq, lens, len(lens)


def finalize_params(name, p, seq1, len1, seq2=None, len2=None):
    """attach what the reference's Evaluator would have derived from this input"""
    if name not in OVERREP:
        return p
    from fastp_amd import hostloop
    b1 = _ArrayBatch(seq1, len1)
    e1 = evalport.evaluate_seq_len(b1)
    s1 = evalport.evaluate_overrep_seqs(b1, e1)
    e2, s2 = 0, []
    if seq2 is not None:
        b2 = _ArrayBatch(seq2, len2)
        e2 = evalport.evaluate_seq_len(b2)
        s2 = evalport.evaluate_overrep_seqs(b2, e2)
    return abi.set_overrep(p, s1, s2, e1, e2, OVERREP[name])


# files a case's reference run needs next to its inputs
FILES = {"pe_adapter_fasta": {"adapters.fa": FASTA_FILE}, "se_adapter_fasta": {"adapters.fa": FASTA_FILE}}

# host-side UMI name editing that goes with a case (the engine only trims the sequence)
UMI = {
    "pe_umi_per_read": ("per_read", 6),
    "se_umi_read1": ("read1", 8),
    "se_exotic_adapter": ("read1", 6),
}


# Filter::trimAndCut stress: (paired, params overrides) run on synth.noisy_reads
TRIM_STRESS = [
    (True, dict(cut_front=1, cut_tail=1, cut_front_window=1, cut_tail_window=1)),
    (True, dict(cut_front=1, cut_right=1, cut_front_window=3, cut_right_window=7, cut_front_quality=25,
                cut_right_quality=12)),
    (True, dict(cut_front=1, cut_tail=1, cut_right=1, cut_front_window=4, cut_tail_window=9, cut_right_window=2,
                trim_front1=2, trim_tail1=3, trim_front2=5, trim_tail2=1)),
    (True, dict(cut_tail=1, cut_tail_window=33, cut_tail_quality=15, umi_len1=5, umi_len2=7, umi_skip=1)),
    (True, dict(cut_front=1, cut_front_window=40, cut_front_quality=10, umi_len1=4, trim_tail1=10, trim_tail2=10)),
    (False, dict(cut_right=1, cut_right_window=16, cut_right_quality=30, adapter_enabled=0)),
    (False, dict(cut_front=1, cut_tail=1, cut_front_window=149, cut_tail_window=150, adapter_enabled=0)),
    (False, dict(cut_front=1, cut_right=1, cut_front_window=1000, cut_right_window=5, adapter_enabled=0)),
    (True, dict(trim_front1=20, trim_tail1=140, trim_front2=0, trim_tail2=151)),
]

# OverlapAnalysis::analyze stress: params overrides run on synth.overlap_pairs (paired)
OVERLAP_STRESS = [
    dict(),
    dict(correction=1),
    dict(overlap_require=5, overlap_diff_limit=20, overlap_diff_percent_limit=50, correction=1),
    dict(overlap_require=40, overlap_diff_limit=2, overlap_diff_percent_limit=5),
    dict(overlap_require=1, overlap_diff_limit=0, overlap_diff_percent_limit=0, adapter_enabled=0),
    dict(overlap_require=16, overlap_diff_limit=5, overlap_diff_percent_limit=20, cut_front=1, cut_tail=1, poly_g=1),
    dict(overlap_require=149, overlap_diff_limit=60, overlap_diff_percent_limit=100),
    dict(allow_gap_overlap_trimming=1, correction=1),
    dict(allow_gap_overlap_trimming=1, correction=1, overlap_require=10, overlap_diff_limit=10, overlap_diff_percent_limit=40),
    dict(allow_gap_overlap_trimming=1, overlap_require=5, overlap_diff_limit=20, overlap_diff_percent_limit=50),
]

# merge mode stress: params overrides run on synth.overlap_pairs (paired)
MERGE_STRESS = [
    dict(merge=1, correction=1),
    dict(merge=1, correction=1, merge_include_unmerged=1, dedup=1, dup_accuracy_level=3),
    dict(merge=1, correction=1, complexity_filter=1, complexity_threshold=0.70, n_base_limit=1, length_required=100),
    dict(merge=1, correction=1, adapter_enabled=0, cut_front=1, cut_tail=1, overlap_require=10, max_len1=120, max_len2=90),
    dict(merge=1, correction=1, merge_include_unmerged=1, poly_x=1, trim_front1=3, trim_front2=7, avg_qual_req=30),
]


# --merge on the lane plan (DevParams::merge_lane): params overrides on top of --merge -c --cut_right
MERGE_LANE = [
    dict(),
    dict(merge_include_unmerged=1, dedup=1, dup_accuracy_level=2),
    dict(complexity_filter=1, complexity_threshold=0.45, n_base_limit=2, length_required=60, poly_x=1),
    dict(merge_include_unmerged=1, adapter_seq_r1=ADAPTER_R1.encode(), adapter_seq_r2=ADAPTER_R2.encode(), max_len1=140, max_len2=120,
         trim_tail1=3, cut_tail=1),
    dict(adapter_enabled=0, overlap_require=12, overlap_diff_limit=9, overlap_diff_percent_limit=35, avg_qual_req=20, dup_enabled=0),
]


def merge_lane_case(k, L=150, n=900):
    """--merge on the lane plan (DevParams::merge_lane): option set k, and the two kinds of input - inserts around the read length
    and the overlap stress pairs (every insert size, mismatch bursts: the two overlap analyses of a pair can disagree there)"""
    import synth
    p = abi.default_params(True, L)
    p.cut_right = 1
    p.merge, p.correction = 1, 1
    for key, v in MERGE_LANE[k].items():
        setattr(p, key, v)
    a = synth.synth_pairs(n, L=L, seed=900 + k, insert_mean=L * 1.2, insert_sd=L * 0.5, polyg_frac=0.05, polyx_frac=0.1, dup_frac=0.2,
                          ragged_frac=0.1, lowq_site_rate=0.05)
    b = synth.overlap_pairs(n, L=L, seed=950 + k, err=0.04, n_rate=0.01)
    return p, [a, b]
