"""FASTQ files in -> trimmed / filtered FASTQ files out, every per-read step on the device.

The host only moves bytes: it reads raw text chunks (the reference's FastqReader fills 8 MiB blocks,
fastqreader.cpp:31,88-149), ships them to HBM, and writes the text that comes back (WriterThread,
writerthread.cpp:118-168).  Line splitting + packing (`fastp_gpu_parse_fastq`), the worker loop
(`fastp_gpu_submit_device`) and record formatting (`fastp_gpu_format_fastq`) run on the GPU; what the
reference's processSingleEnd / processPairEnd would have written to out1 / out2 comes out byte for
byte.  Input files may be plain FASTQ or BGZF (.gz written by bgzip: the reference's BgzfMtReader path,
src/bgzf.h) - those are shipped compressed and inflated on the device (`fastp_gpu_inflate_bgzf`).
Every output stream of the worker loop is assembled on the device (`fastp_gpu_format_streams`): out1 / out2,
--failed_out, --unpaired1/2, the merged stream, UMI-renamed records; an output path ending in ".gz" is compressed
on the device too (`fastp_gpu_deflate_bgzf`: BGZF-framed gzip members, what the reference's WriterThread does per
pack with libdeflate).  Reader and writer run on their own threads so file I/O overlaps the device work of the
neighbouring chunks.
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
import time

import numpy as np

from . import abi, engine


class PipelineError(RuntimeError):
    pass


class _Mate:
    """per-mate buffers of one in-flight chunk"""

    def __init__(self, torch, dev, text, max_records, max_len):
        ss, qs = abi.seq_stride(max_len), abi.qual_stride(max_len)
        self.cap = text.numel()                   # [text carried from the previous chunk | this chunk's text]
        self.text = text
        self.seq = torch.empty(max_records * ss, dtype=torch.uint8, device=dev)
        self.qual = torch.empty(max_records * qs, dtype=torch.uint8, device=dev)
        self.lens = torch.empty(max_records, dtype=torch.int16, device=dev)
        self.loff = torch.empty(4 * max_records, dtype=torch.int32, device=dev)
        self.llen = torch.empty(4 * max_records, dtype=torch.int32, device=dev)
        self.res = torch.zeros(max_records * 12, dtype=torch.uint8, device=dev)


class FastqPipeline:
    def __init__(self, params: abi.Params, device: int = 0, chunk_bytes: int = 256 << 20, max_records: int | None = None,
                 corr_capacity: int = 1 << 22):
        import torch
        self.torch = torch
        self.params = params
        self.paired = bool(params.paired)
        self.dev = torch.device("cuda", device)
        self.eng = engine.GpuEngine(params, device=device)
        self.chunk = int(chunk_bytes)
        self.max_records = int(max_records or max(1024, (2 * self.chunk) // 48))
        nm = 2 if self.paired else 1
        # one allocation for both mates' text (and, for BGZF input, one for their compressed chunks): a single
        # inflate launch then covers the blocks of both files
        self.text_cap = (2 * self.chunk + 64 + 255) // 256 * 256
        self.text_all = torch.zeros(nm * self.text_cap, dtype=torch.uint8, device=self.dev)
        self.comp_cap = (self.chunk + 64 + 255) // 256 * 256
        self.comp_all = None
        self.idx_all = None
        self.mates = [_Mate(torch, self.dev, self.text_all[m * self.text_cap:(m + 1) * self.text_cap], self.max_records,
                            params.max_len) for m in range(nm)]
        self.pair = torch.zeros(self.max_records * 8, dtype=torch.uint8, device=self.dev)
        self.corr_cap = corr_capacity if params.correction else 0
        self.corr = torch.zeros(max(1, self.corr_cap) * 8, dtype=torch.uint8, device=self.dev)
        self.nc = torch.zeros(4, dtype=torch.int32, device=self.dev)
        nf = int(params.n_adapter_fasta)
        self.ev_cap = self.max_records * 2 * min(nf, 8) if nf else 0
        self.ev = torch.zeros(max(1, self.ev_cap) * 12, dtype=torch.uint8, device=self.dev)
        self.nev = torch.zeros(4, dtype=torch.int32, device=self.dev)
        # two pinned staging sets per direction: the reader fills one while the device works on the other
        self.stage_in = [[torch.empty(self.chunk + 64, dtype=torch.uint8).pin_memory() for _ in range(nm)] for _ in range(2)]
        self.stage_out = None      # [slot][stream]: pinned, sized in run() for the streams that are asked for
        self.outs = [None] * abi.N_OUTPUTS       # device text per stream
        self.gzbuf = [None] * abi.N_OUTPUTS      # device gzip members per compressed stream
        self.max_blocks = (2 * self.chunk) // 8192 + 64     # BGZF members per chunk (bgzip: ~64 KiB of text each)
        self.check_crc = True
        self.stats = dict(units=0, chunks=0, bytes_in=0, bytes_out=0, t_parse=0.0, t_engine=0.0, t_format=0.0, t_h2d=0.0,
                          t_d2h=0.0, t_wait_read=0.0, t_wait_write=0.0)

    def close(self):
        self.eng.close()

    # -- reader thread: (leftover of the previous chunk | fresh bytes) into a pinned staging buffer ------------
    IO_THREADS = int(os.environ.get("FASTP_PIPELINE_IO_THREADS", "8"))   # positional reads / writes in flight per direction
    IO_PIECE = 32 << 20

    def _reader(self, files, q_free, q_full, tails):
        """fills a staging set: [file bytes the last trip left over | fresh bytes], all mates concurrently;
        tails[m] = bytes per trip for file m"""
        import os
        from concurrent.futures import ThreadPoolExecutor
        fds = [f.fileno() for f in files]
        sizes = [os.fstat(fd).st_size for fd in fds]
        pos = [0] * len(files)

        def piece(fd, mv, off):
            got = 0
            while got < len(mv):
                r = os.preadv(fd, [mv[got:]], off + got)
                if r <= 0:
                    raise PipelineError("short read")
                got += r

        try:
            with ThreadPoolExecutor(self.IO_THREADS) as pool:
                while True:
                    item = q_free.get()
                    if item is None:
                        return
                    slot, carry, budget = item
                    fills, futs = [], []
                    for m in range(len(files)):
                        buf = self.stage_in[slot][m].numpy()
                        k = len(carry[m])
                        if k:
                            buf[:k] = np.frombuffer(carry[m], dtype=np.uint8)
                        want = max(0, min(budget[m] - k, sizes[m] - pos[m]))
                        mv = memoryview(buf)[k:k + want]
                        for a in range(0, want, self.IO_PIECE):
                            e = min(want, a + self.IO_PIECE)
                            futs.append(pool.submit(piece, fds[m], mv[a:e], pos[m] + a))
                        pos[m] += want
                        fills.append((k + want, pos[m] >= sizes[m]))
                    for fu in futs:
                        fu.result()
                    q_full.put((slot, fills))
        except Exception as e:  # surface in the main thread
            q_full.put(e)

    def _writer(self, files, q_out, q_done):
        import os
        from concurrent.futures import ThreadPoolExecutor
        fds = [f.fileno() if f is not None else -1 for f in files]
        pos = [0] * len(files)

        def piece(fd, mv, off):
            done = 0
            while done < len(mv):
                done += os.pwritev(fd, [mv[done:]], off + done)

        try:
            with ThreadPoolExecutor(self.IO_THREADS) as pool:
                while True:
                    item = q_out.get()
                    if item is None:
                        return
                    slot, lens = item
                    futs = []
                    for m in range(len(files)):
                        if files[m] is None or not lens[m]:
                            continue
                        mv = memoryview(self.stage_out[slot][m].numpy())[:lens[m]]
                        for a in range(0, lens[m], self.IO_PIECE):
                            e = min(lens[m], a + self.IO_PIECE)
                            futs.append(pool.submit(piece, fds[m], mv[a:e], pos[m] + a))
                        pos[m] += lens[m]
                    for fu in futs:
                        fu.result()
                    q_done.put(slot)
        except Exception as e:
            q_done.put(e)

    @staticmethod
    def is_bgzf(path: str) -> bool:
        """isBgzf (src/bgzf.h:17-27): gzip magic, FEXTRA, BC subfield of length 2 first"""
        with open(path, "rb") as f:
            h = f.read(18)
        return (len(h) == 18 and h[0] == 0x1f and h[1] == 0x8b and h[2] == 8 and (h[3] & 4) and h[12] == 0x42 and h[13] == 0x43
                and (h[14] | (h[15] << 8)) == 2)

    def _inflate_all(self, slot, fills, tcarry, gz):
        """BGZF bytes in the staging buffers -> text on the device behind each mate's carried text, ONE launch for
        the blocks of all compressed mates.  Returns per mate (text bytes added, compressed bytes consumed)."""
        torch = self.torch
        nm = len(self.mates)
        if self.comp_all is None:
            self.comp_all = torch.zeros(nm * self.comp_cap, dtype=torch.uint8, device=self.dev)
            self.idx_all = torch.zeros(nm * self.max_blocks * 24 + 64, dtype=torch.uint8, device=self.dev)
        res = [(0, 0)] * nm
        parts = [[] for _ in range(5)]
        for m in range(nm):
            if not gz[m]:
                continue
            nb, eof_file = fills[m]
            host = self.stage_in[slot][m].numpy()[:nb]
            room = self.mates[m].cap - 64 - tcarry[m]
            info, poff, plen, isz, crc, ooff = self.eng.bgzf_index(host, self.max_blocks, room)
            nblk, used = int(info.n_blocks), int(info.consumed)
            if nblk == 0:
                if eof_file and nb > 0:
                    raise PipelineError(f"truncated BGZF member at the end of mate {m + 1}'s file")
                continue
            c0 = m * self.comp_cap
            self.comp_all[c0:c0 + used].copy_(self.stage_in[slot][m][:used], non_blocking=True)
            self.comp_all[c0 + used:c0 + used + 32].zero_()
            parts[0].append(poff + np.uint32(c0))
            parts[1].append(plen)
            parts[2].append(isz)
            parts[3].append(crc)
            parts[4].append(ooff + np.uint64(m * self.text_cap + tcarry[m]))
            res[m] = (int(info.out_bytes), used)
        n = sum(len(a) for a in parts[0])
        if n == 0:
            return res
        packed = np.concatenate([np.concatenate(p).view(np.uint8) for p in parts])   # 4+4+4+4+8 bytes per block
        self.idx_all[:packed.size].copy_(torch.from_numpy(packed), non_blocking=False)
        base = self.idx_all.data_ptr()
        torch.cuda.synchronize(self.dev)
        self.eng.inflate_bgzf(self.comp_all.data_ptr(), n, base, base + 4 * n, base + 8 * n, base + 12 * n, base + 16 * n,
                              self.text_all.data_ptr(), self.text_all.numel(), self.check_crc)
        return res

    EOF_MEMBER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")   # bgzip's empty last member

    def _setup_outputs(self, paths, umi=None):
        """device + pinned buffers for the streams that are written; paths[q] per abi stream index"""
        torch = self.torch
        nm = len(self.mates)
        # what a record can grow by over its input text: the UMI name tag (addUmiToName: delimiter + prefix + '_' + the
        # UMI of one or both mates joined by '_'), the failed / merged tags on the name and the strand line
        grow = 0
        if umi:
            grow = len(umi[3] if len(umi) > 3 and umi[3] else b":") + (len(umi[2]) + 1 if len(umi) > 2 and umi[2] else 0) + \
                2 * int(umi[1]) + 1
        both = nm * self.text_cap + self.max_records * (96 + 2 * grow)          # every record of both mates + tags
        one = self.text_cap + self.max_records * (64 + grow)
        caps = [one, one, both, both, both, both]
        self.out_cap = [0] * abi.N_OUTPUTS
        self.gz_out = [bool(p) and p.endswith(".gz") for p in paths]
        for q in range(abi.N_OUTPUTS):
            if not paths[q]:
                continue
            self.out_cap[q] = caps[q]
            if self.outs[q] is None or self.outs[q].numel() < caps[q]:
                self.outs[q] = torch.empty(caps[q], dtype=torch.uint8, device=self.dev)
            if self.gz_out[q]:
                gcap = caps[q] + 31 * (caps[q] // 65280 + 1) + 64
                if self.gzbuf[q] is None or self.gzbuf[q].numel() < gcap:
                    self.gzbuf[q] = torch.empty(gcap, dtype=torch.uint8, device=self.dev)
        need = [(self.gzbuf[q].numel() if self.gz_out[q] else caps[q]) if paths[q] else 0 for q in range(abi.N_OUTPUTS)]
        if self.stage_out is None:
            self.stage_out = [[None] * abi.N_OUTPUTS for _ in range(2)]
        for sl in range(2):
            for q in range(abi.N_OUTPUTS):
                have = self.stage_out[sl][q]
                if need[q] and (have is None or have.numel() < need[q]):
                    self.stage_out[sl][q] = torch.empty(need[q], dtype=torch.uint8).pin_memory()

    def run(self, in1: str, in2: str | None, out1: str, out2: str | None, failed_out: str | None = None,
            merged_out: str | None = None, unpaired1: str | None = None, unpaired2: str | None = None,
            umi: tuple | None = None) -> dict:
        """umi = (location "read1" | "read2" | "per_read", length[, prefix bytes[, delimiter bytes]]): the name edit that goes
        with params.umi_len1/2 (UmiProcessor::addUmiToName)"""
        torch = self.torch
        if self.paired != (in2 is not None) or self.paired != (out2 is not None):
            raise PipelineError("paired engine needs in2/out2, single-end engine must not get them")
        if self.params.merge and not merged_out:
            raise PipelineError("merge mode needs merged_out")
        if (self.params.umi_len1 or self.params.umi_len2) and umi is None:
            raise PipelineError("params trim a UMI off the reads: pass umi=(location, length) for the name edit")
        out_paths = [out1, out2, failed_out, merged_out if self.params.merge else None, unpaired1 if self.paired else None,
                     unpaired2 if self.paired else None]
        self._setup_outputs(out_paths, umi)
        self.fmt_opts = abi.FormatOptions()
        self.fmt_opts.want_failed = int(bool(failed_out))
        self.fmt_opts.want_unpaired1 = int(bool(out_paths[4]))
        self.fmt_opts.want_unpaired2 = int(bool(out_paths[5]))
        if umi is not None:
            self.fmt_opts.umi_loc = {"read1": 1, "read2": 2, "per_read": 3}[umi[0]]
            self.fmt_opts.umi_len = int(umi[1])
            self.fmt_opts.umi_prefix = umi[2] if len(umi) > 2 and umi[2] else None
            self.fmt_opts.umi_delimiter = umi[3] if len(umi) > 3 else None
        nm = len(self.mates)
        paths = (in1, in2) if self.paired else (in1,)
        gz = [self.is_bgzf(p) for p in paths]
        for p, g in zip(paths, gz):
            if not g:
                with open(p, "rb") as f:
                    if f.read(2) == b"\x1f\x8b":
                        raise PipelineError(f"{p}: gzip but not BGZF - a single deflate stream has no independent blocks; "
                                            "decompress on the host (as the reference does) or recompress with bgzip")
        fin = [open(p, "rb", buffering=0) for p in paths]
        fout = [open(p, "wb", buffering=0) if p else None for p in out_paths]
        q_free, q_full, q_out, q_done = queue.Queue(), queue.Queue(), queue.Queue(), queue.Queue()
        # a compressed chunk expands ~4-5x: read a quarter of the text budget per trip
        want = [self.chunk // 4 if g else self.chunk for g in gz]
        rd = threading.Thread(target=self._reader, args=(fin, q_free, q_full, want), daemon=True)
        wr = threading.Thread(target=self._writer, args=(fout, q_out, q_done), daemon=True)
        rd.start()
        wr.start()
        st = self.stats
        st.setdefault("t_inflate", 0.0)
        t_start = time.perf_counter()
        tcarry = [0] * nm          # text bytes already at the front of mates[m].text (device-side carry)
        try:
            q_free.put((0, [b""] * nm, list(want)))
            out_free = [0, 1]
            done = False
            drain = False              # every file is at EOF but the device still holds complete records
            slot = 0
            while not done:
                if drain:
                    fills = [(0, True)] * nm   # nothing new from the files: parse what was carried over
                else:
                    t0 = time.perf_counter()
                    item = q_full.get()
                    if isinstance(item, Exception):
                        raise item
                    slot, fills = item
                    st["t_wait_read"] += time.perf_counter() - t0
                st["bytes_in"] += sum(f[0] for f in fills)
                total, fcarry, eof = [0] * nm, [b""] * nm, [False] * nm
                if any(gz):
                    t0 = time.perf_counter()
                    inflated = self._inflate_all(slot, fills, tcarry, gz)
                    st["t_inflate"] += time.perf_counter() - t0
                for m in range(nm):
                    nb, eof_file = fills[m]
                    M = self.mates[m]
                    if gz[m]:
                        added, used = inflated[m]
                        if nb > used:
                            fcarry[m] = bytes(memoryview(self.stage_in[slot][m].numpy())[used:nb])
                        if added == 0 and not eof_file and len(fcarry[m]) >= want[m]:
                            raise PipelineError("the text buffer cannot take another BGZF block: raise chunk_bytes")
                        total[m] = tcarry[m] + added
                        eof[m] = eof_file and not fcarry[m]
                    else:
                        t0 = time.perf_counter()
                        if tcarry[m] + nb > M.cap - 64:
                            raise PipelineError("a record does not fit the chunk size")
                        M.text[tcarry[m]:tcarry[m] + nb].copy_(self.stage_in[slot][m][:nb], non_blocking=True)
                        torch.cuda.synchronize(self.dev)
                        st["t_h2d"] += time.perf_counter() - t0
                        total[m] = tcarry[m] + nb
                        eof[m] = eof_file
                    M.text[total[m]:total[m] + 32].zero_()
                all_eof = all(eof)
                # ---- parse (both mates to the same record count) ----
                t0 = time.perf_counter()
                infos = [self._parse(m, total[m], eof[m], self.max_records) for m in range(nm)]
                n = min(i.n_records for i in infos)
                for m in range(nm):
                    if infos[m].n_records != n:
                        infos[m] = self._parse(m, total[m], eof[m], n)
                st["t_parse"] += time.perf_counter() - t0
                left = [total[m] - int(infos[m].consumed) for m in range(nm)]
                # the reader refills the other staging set while the device works on this chunk; a plain-text mate
                # whose tail stays on the device gets that much less this trip
                if not all_eof:
                    q_free.put((1 - slot, fcarry, [want[m] if gz[m] else max(0, want[m] - left[m]) for m in range(nm)]))
                if all_eof:
                    # a trip takes at most max_records records: when the cap was hit, complete records may remain in
                    # the carried text - keep parsing without refilling until a trip comes back short.  What is left
                    # then is a trailing partial record / the longer mate's surplus: the reference stops too
                    drain = n > 0 and n >= self.max_records and any(left)
                    done = not drain
                elif n == 0 and any(left[m] >= self.chunk for m in range(nm)):
                    raise PipelineError("a record does not fit the chunk size")
                if n > 0:
                    lens = self._process_and_format(n, st)
                    # ---- D2H + hand to the writer ----
                    t0 = time.perf_counter()
                    while not out_free:
                        d = q_done.get()
                        if isinstance(d, Exception):
                            raise d
                        out_free.append(d)
                    st["t_wait_write"] += time.perf_counter() - t0
                    oslot = out_free.pop(0)
                    t0 = time.perf_counter()
                    for q in range(abi.N_OUTPUTS):
                        if fout[q] is not None and lens[q]:
                            src = self.gzbuf[q] if self.gz_out[q] else self.outs[q]
                            self.stage_out[oslot][q][:lens[q]].copy_(src[:lens[q]], non_blocking=True)
                    torch.cuda.synchronize(self.dev)
                    st["t_d2h"] += time.perf_counter() - t0
                    q_out.put((oslot, lens))
                    st["units"] += n
                    st["chunks"] += 1
                    st["bytes_out"] += sum(lens)
                # ---- the unconsumed text tail moves to the front (after formatting: the records point into the text) ----
                for m in range(nm):
                    a = int(infos[m].consumed)
                    if left[m] and a:
                        tail = self.mates[m].text[a:total[m]].clone()
                        self.mates[m].text[:left[m]].copy_(tail)
                    tcarry[m] = left[m]
        finally:
            q_free.put(None)
            q_out.put(None)
            wr.join()
            rd.join(timeout=5)
            for q, f in enumerate(fout):
                if f is not None and self.gz_out[q]:   # the writer thread has joined: append after its last positional write
                    os.pwrite(f.fileno(), self.EOF_MEMBER, os.fstat(f.fileno()).st_size)
            for f in fin + fout:
                if f is not None:
                    f.close()
        while not q_done.empty():
            d = q_done.get()
            if isinstance(d, Exception):
                raise d
        st["wall"] = time.perf_counter() - t_start
        return dict(st)

    def _process_and_format(self, n, st):
        """worker loop + record formatting of the parsed chunk; returns the output byte counts per stream"""
        nm = len(self.mates)
        t0 = time.perf_counter()
        b = abi.Batch()
        b.n, b.flags = n, abi.BATCH_STAT_ISIZE
        M = self.mates
        b.seq1, b.qual1, b.len1 = M[0].seq.data_ptr(), M[0].qual.data_ptr(), M[0].lens.data_ptr()
        if self.paired:
            b.seq2, b.qual2, b.len2 = M[1].seq.data_ptr(), M[1].qual.data_ptr(), M[1].lens.data_ptr()
        xu = np.unique(np.concatenate([getattr(M[m], "exotic", np.zeros(0, dtype=np.int32)) for m in range(nm)])).astype(np.int32)
        xu = np.ascontiguousarray(xu[xu < n])
        if len(xu):
            b.n_exotic, b.exotic_dense, b.exotic_unit = len(xu), 1, xu.ctypes.data
            for m in range(nm):
                b.exotic_text[m], b.exotic_off[m] = M[m].text.data_ptr(), M[m].loff.data_ptr()
        r = abi.Results()
        r.r1 = M[0].res.data_ptr()
        if self.paired:
            r.r2, r.pair = M[1].res.data_ptr(), self.pair.data_ptr()
        if self.corr_cap:
            r.corrections, r.corrections_capacity = self.corr.data_ptr(), self.corr_cap
        r.n_corrections = self.nc.data_ptr()
        if self.ev_cap:
            r.adapter_events, r.adapter_events_capacity, r.n_adapter_events = self.ev.data_ptr(), self.ev_cap, self.nev.data_ptr()
        self.eng.submit_device(b, r)
        self.eng.synchronize()
        if self.corr_cap and int(self.nc[0].item()) > self.corr_cap:   # before anything reads the list
            raise PipelineError("correction list overflow: raise corr_capacity")
        st["t_engine"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        fi = []
        for m in range(nm):
            f = abi.FormatIn()
            f.text, f.line_off, f.line_len, f.res = M[m].text.data_ptr(), M[m].loff.data_ptr(), M[m].llen.data_ptr(), M[m].res.data_ptr()
            fi.append(f)
        rc, lens = self.eng.format_streams(n, fi[0], fi[1] if self.paired else None, self.pair.data_ptr() if self.paired else None,
                                           self.corr.data_ptr() if self.corr_cap else None,
                                           self.nc.data_ptr() if self.corr_cap else None, self.fmt_opts,
                                           [t.data_ptr() if t is not None and self.out_cap[q] else None for q, t in enumerate(self.outs)],
                                           self.out_cap)
        if self.corr_cap and int(self.nc[0].item()) > self.corr_cap:
            raise PipelineError("correction list overflow: raise corr_capacity")
        st["t_format"] += time.perf_counter() - t0
        st["bytes_text"] = st.get("bytes_text", 0) + sum(lens)
        t0 = time.perf_counter()
        for q in range(abi.N_OUTPUTS):
            if self.out_cap[q] and self.gz_out[q]:
                rc, lens[q] = self.eng.deflate_bgzf(self.outs[q].data_ptr(), lens[q], self.gzbuf[q].data_ptr(), self.gzbuf[q].numel())
        st["t_deflate"] = st.get("t_deflate", 0.0) + time.perf_counter() - t0
        return lens

    def _parse(self, m, nbytes, is_last, max_records):
        M = self.mates[m]
        info = self.eng.parse_fastq(M.text.data_ptr(), nbytes, is_last, max_records, M.seq.data_ptr(), M.qual.data_ptr(),
                                    M.lens.data_ptr(), M.loff.data_ptr(), M.llen.data_ptr())
        if info.first_bad >= 0:
            raise PipelineError(f"malformed FASTQ record {info.first_bad} of a chunk (mate {m + 1}): the device parser does not repair input")
        # records with letters outside ACGTN: the engine's text kernel takes those units from the chunk's own text
        M.exotic = self.eng.parse_exotic() if info.n_exotic else np.zeros(0, dtype=np.int32)
        return info

    def counters(self):
        return self.eng.counters()
