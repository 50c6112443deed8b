"""Host-side mirror of the reference's `scanfastq` worker for one chunk of FASTQ text (WorkerReadscanner.scan ->
Parser.call, FJ!nanoporereadscanner/WorkerReadscanner.java:L186-273, FJ!nanoporereadscanner/analyzers/Parser.java:L132-185):
everything per read runs in the HIP library behind include/sicelore_mi.h; this module only strings the entry points
together in the reference's order and formats the output records.  No CPU fallback: it needs a Context.

pass 1 (UsedCellBCListGenerator): FASTQ text -> K-FQ -> K-PACK (+ qualities) -> K-SCAN (complete adapter, quality
filter) -> histogram of whitelist members.
pass 2 (Parser): FASTQ text -> K-FQ -> [K-PACKR -> K-CHIM -> fragment offsets] -> K-PACK -> K-SCAN -> K-BC1/K-BC2 ->
K-WRITE (`pass2_write_chunk`: the `passed` / `failed` FASTQ text assembled on the device), or per-record names on the
host (`pass2_chunk`: smi_format_read_name / smi_chimera_fragment_name).
"""
import numpy as np
import torch

from . import lib as _lib

READS_AFTER_SPLIT = 1 << 3            # ReadFlags$Flags.getValue() (ReadFlags.java:L72-109; sicelore_mi.h SMI_F_*)
MULTI_CHIMERIC_READS_DISCARDED = 1 << 2
FAILED = 1 << 5
PASSED_ANY = (1 << _lib.FLAG_BITS["PASSED_FWD"]) | (1 << _lib.FLAG_BITS["PASSED_REV"])


def read_fastq_file(path):
    """bytes of a FASTQ file; *.gz (plain or block gzip) is inflated by the library (smi_gz_inflate), as FastqFileReader
    does through GZIPInputStream (FastqFileReader.java:L138-150)"""
    raw = np.fromfile(path, dtype=np.uint8)
    if str(path).lower().endswith(".gz"):
        return _lib.gz_inflate(raw).tobytes()
    return raw.tobytes()


class ReadScanner:
    def __init__(self, ctx, max_ed=1, five_prime=False, dont_search_polya=False, split_chimeras=True):
        self.ctx, self.max_ed, self.five_prime = ctx, int(max_ed), bool(five_prime)
        self.dont_search_polya = bool(dont_search_polya)
        # Parser.java:L176: chimeras are split in pass 2 unless --noPolyARequired
        self.split_chimeras = bool(split_chimeras) and not (five_prime and dont_search_polya)
        self.dev = torch.device("cuda", ctx.device)

    # ---- FASTQ text -> contiguous reads / qualities -------------------------------------------------------------
    def _ingest(self, text, want_quals, want_names=True):
        t = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).to(self.dev)
        cap = text.count(b"\n") // 4 + 2
        i64 = lambda n: torch.zeros(n, dtype=torch.int64, device=self.dev)  # noqa: E731
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=self.dev)  # noqa: E731
        line, ns, ss, qs, offs, nl, sl = i64(4 * cap + 8), i64(cap), i64(cap), i64(cap), i64(cap + 1), i32(cap), i32(cap)
        n, err = self.ctx.fastq_index_device(t, len(text), line, ns, nl, ss, sl, qs, offs, cap)
        if err:
            raise _lib.SmiError(f"malformed FASTQ (smi_fastq_index_device error bits {err})")  # the reference throws too
        total = int(offs[n].item()) if n else 0
        reads = torch.zeros(max(total, 1), dtype=torch.uint8, device=self.dev)
        self.ctx.fastq_gather_device(t, ss, offs, n, reads)
        quals = None
        if want_quals:
            quals = torch.zeros(max(total, 1), dtype=torch.uint8, device=self.dev)
            self.ctx.fastq_gather_device(t, qs, offs, n, quals)
        self._text, self._lines = t, line  # the record writer takes the names and '+' lines from the chunk itself
        if not want_names:
            return n, total, reads, quals, offs[:n + 1].contiguous(), None
        ns_h, nl_h = ns[:n].cpu().numpy(), nl[:n].cpu().numpy()
        names = [text[int(a):int(a) + int(b)].decode() for a, b in zip(ns_h, nl_h)]
        return n, total, reads, quals, offs[:n + 1].contiguous(), names

    def _scan(self, reads, quals, offs, n, pass_no):
        ends = torch.zeros((28, 2 * max(n, 1)), dtype=torch.int32, device=self.dev)
        lens = torch.zeros(max(n, 1), dtype=torch.int32, device=self.dev)
        qt = qsum = None
        if quals is not None:
            qt = torch.zeros((max(n, 1), 224), dtype=torch.uint8, device=self.dev)
            qsum = torch.zeros(max(n, 1), dtype=torch.int32, device=self.dev)
        self.ctx.pack_ends_device(reads, quals, offs, n, ends, lens, qt, qsum, five_prime=self.five_prime)
        cfg = self.ctx.scan_config_5p(pass_no, self.dont_search_polya) if self.five_prime else self.ctx.scan_config(pass_no)
        scan = torch.zeros((max(n, 1), 8), dtype=torch.int32, device=self.dev)
        win = torch.zeros((max(n, 1), 2), dtype=torch.int64, device=self.dev)
        self.ctx.scan_device(ends, lens, n, cfg, scan, win, qt, qsum)
        return scan, win

    # ---- pass 1 ----------------------------------------------------------------------------------------------------
    def pass1_chunk(self, text, hist):
        """adds this chunk's whitelist hits to `hist` (int32 device tensor, one counter per loaded barcode); -> n reads"""
        n, _total, reads, quals, offs, _names = self._ingest(text, want_quals=True, want_names=False)
        if n:
            scan, win = self._scan(reads, quals, offs, n, pass_no=1)
            self.ctx.hist_windows_device(win, scan, n, hist)
        return n

    # ---- pass 2 ----------------------------------------------------------------------------------------------------
    def _pass2_device(self, reads, quals, offs, n, total):
        """splitter -> scan -> barcode match, everything left on the device"""
        d_chim = fsrc = None
        n_out = n
        if self.split_chimeras:
            planes = torch.zeros(self.ctx.read_planes_words(total, n), dtype=torch.int32, device=self.dev)
            self.ctx.pack_reads_device(reads, offs, n, total, planes)
            d_chim = torch.zeros((n, 4), dtype=torch.int32, device=self.dev)
            self.ctx.chimera_device(planes, offs, n, total, self.ctx.chimera_config(self.five_prime), d_chim)
            scratch = torch.zeros((n + 1023) // 1024 + 1, dtype=torch.int32, device=self.dev)
            nfrag = torch.zeros(1, dtype=torch.int64, device=self.dev)
            foffs = torch.zeros(3 * n + 1, dtype=torch.int64, device=self.dev)
            fsrc = torch.zeros(3 * n, dtype=torch.int32, device=self.dev)
            self.ctx.split_offsets_device(d_chim, offs, n, scratch, nfrag, foffs, fsrc)
            n_out = int(nfrag.item())
            flags = (d_chim.view(torch.uint8).view(n, 16)[:, 11]).to(torch.int32)  # smi_chimera_result.flags
            if int((flags & (_lib.CHIM_RANGE | _lib.CHIM_OVERFLOW)).max().item()):
                raise _lib.SmiError("chimera splitter: a read outside the supported range (SMI_CHIM_RANGE / SMI_CHIM_OVERFLOW)")
            offs = foffs[:n_out + 1].contiguous()
            fsrc = fsrc[:n_out].contiguous()
        scan_d, win = self._scan(reads, None, offs, n_out, pass_no=2)  # the quality filter belongs to pass 1 (UsedCellBCListGenerator L198-202)
        res_d = torch.zeros((max(n_out, 1), 4), dtype=torch.int32, device=self.dev)
        self.ctx.bc_match_device(win, res_d, n_out, max_ed=self.max_ed, five_prime=self.five_prime)
        if fsrc is not None and n_out:
            # a read the splitter discarded whole is never scanned by the reference (Parser.java:L92): no barcode for its record, as the
            # chunk workers' k_drop_discarded has it
            multi = (flags[(fsrc >> 2).to(torch.int64)] & _lib.CHIM_MULTI) != 0
            res_d.view(torch.uint8).view(-1, 16)[:n_out, 8][multi] = 0
        return n_out, offs, d_chim, fsrc, scan_d, res_d

    def pass2_write_chunk(self, text, rank_keys=None, rank_values=None, first_read_id=1, trim_fastq=False):
        """-> (passed FASTQ bytes, failed FASTQ bytes, info): the records of the chunk as the reference writes them
        (FastqWriterThreadPool$FastQoneFileThread.run), assembled by K-WRITE.  rank_keys (sorted uint64) / rank_values:
        the used list of pass 1, for the rk= field.  info: n_records, n_passed, is_passed, rec_off, scan, bc (numpy)."""
        n, total, reads, quals, offs, _ = self._ingest(text, want_quals=True, want_names=False)
        if n == 0:
            return b"", b"", dict(n_records=0, n_passed=0)
        n_out, offs, d_chim, fsrc, scan_d, res_d = self._pass2_device(reads, quals, offs, n, total)
        d_rank = None
        bc = res_d.cpu().numpy().view(_lib.BC_RESULT_DTYPE).reshape(-1)[:n_out]
        if rank_keys is not None and len(rank_keys):
            keys = bc["bc"].astype(np.uint64)
            pos = np.minimum(np.searchsorted(rank_keys, keys), len(rank_keys) - 1)
            hit = (bc["found"] == 1) & (np.asarray(rank_keys)[pos] == keys)
            d_rank = torch.from_numpy(np.where(hit, np.asarray(rank_values)[pos], 0).astype(np.int32)).to(self.dev)
        total_out = int(offs[n_out].item())
        # a record is its bases and qualities plus name, '+' line and separators; the name suffix is < 256 bytes
        cap = 2 * total_out + len(text) + 320 * n_out + 64
        out_p = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        out_f = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        rec_off = torch.zeros(n_out + 1, dtype=torch.int64, device=self.dev)
        is_p = torch.zeros(n_out, dtype=torch.uint8, device=self.dev)
        totals = self.ctx.fastq_write_device(self._text, self._lines, reads, quals, offs, fsrc, d_chim, scan_d, res_d, d_rank, n_out,
                                             first_read_id, out_p, out_f, rec_off, is_p, five_prime=self.five_prime,
                                             trim_fastq=trim_fastq)
        info = dict(n_records=n_out, n_passed=int(totals[2]), is_passed=is_p.cpu().numpy().astype(bool),
                    rec_off=rec_off[:n_out].cpu().numpy(), bc=bc,
                    scan=scan_d.cpu().numpy().view(_lib.SCAN_RESULT_DTYPE).reshape(-1)[:n_out])
        return out_p[:int(totals[0])].cpu().numpy().tobytes(), out_f[:int(totals[1])].cpu().numpy().tobytes(), info

    def pass2_chunk(self, text, rank_of=None, first_read_id=0):
        """-> list of dicts per output record (after the chimera split): name (as written by the reference), passed,
        reverse, flags, source (index of the input record), fragment (0..2) and length of the record"""
        n, total, reads, quals, offs, names = self._ingest(text, want_quals=True)
        if n == 0:
            return []
        n_out, offs, d_chim, fsrc, scan_d, res_d = self._pass2_device(reads, quals, offs, n, total)
        src = np.arange(n)
        frag = np.zeros(n, dtype=np.int64)
        chim = None
        if d_chim is not None:
            chim = d_chim.cpu().numpy().view(_lib.CHIMERA_RESULT_DTYPE).reshape(-1)
            fs = fsrc.cpu().numpy()
            src, frag = fs >> 2, fs & 3
        torch.cuda.synchronize()
        scan = scan_d.cpu().numpy().view(_lib.SCAN_RESULT_DTYPE).reshape(-1)[:n_out]
        bc = res_d.cpu().numpy().view(_lib.BC_RESULT_DTYPE).reshape(-1)[:n_out]
        o = offs.cpu().numpy()
        rb, qb = reads.cpu().numpy(), quals.cpu().numpy()
        out = []
        for i in range(n_out):
            r = int(src[i])
            name, flags = names[r], int(scan["flags"][i])
            if chim is not None:
                if chim["n_split"][r]:
                    name = _lib.chimera_fragment_name(name, chim[r], int(frag[i]))
                    flags |= READS_AFTER_SPLIT
                if chim["flags"][r] & _lib.CHIM_MULTI:
                    flags |= MULTI_CHIMERIC_READS_DISCARDED | FAILED
            seq = rb[int(o[i]):int(o[i + 1])].tobytes().decode()
            qual = qb[int(o[i]):int(o[i + 1])].tobytes().decode()
            b = bc[i] if bc["found"][i] == 1 else None
            rk = 0
            if b is not None and rank_of is not None:
                rk = int(rank_of.get(int(b["bc"]), 0))
            failed_multi = chim is not None and bool(chim["flags"][r] & _lib.CHIM_MULTI)
            if failed_multi:  # Parser.processOneRecord L92: FAILED records are not scanned
                full = name.split(" ")[0] + "_FAILED "
                sc = np.zeros(1, dtype=_lib.SCAN_RESULT_DTYPE)[0]
            else:
                sc = scan[i]
                full = _lib.format_read_name(name, seq, qual, sc, b, rank=rk, read_id=first_read_id + i,
                                             five_prime=self.five_prime)
            out.append(dict(name=full, passed=bool(int(sc["flags"]) & PASSED_ANY) and not failed_multi,
                            reverse=bool(sc["reverse"]), flags=flags, source=r, fragment=int(frag[i]),
                            length=int(o[i + 1]) - int(o[i])))
        return out
