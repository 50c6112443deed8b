// smi_name.h -- the name a scanned read is written with, shared by the host entry point (smi_format_read_name) and the
// device record writer (K-WRITE): FastqRecordExt.getRecordForWriting
// (FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L209-311); prefixes from Jar/config.xml:41-52 (PS= PE= AE= T=
// X= Q=) and ReadScannerParameters.java:L139-159 (bc= ed= ed_sec= bcStart= bcEnd= rk=).  Plain character appends, no
// library calls, so the same code runs in a kernel.
#pragma once
#include <cstdint>

#include "sicelore_mi.h"

#if defined(__HIPCC__)
#define SMI_HD __host__ __device__ __forceinline__
#else
#define SMI_HD inline
#endif

namespace smi {

struct NameSink {
    char *p;
    int n, cap;
    SMI_HD void put(char c) {
        if (n < cap) p[n] = c;
        n++;  // n > cap afterwards: overflow
    }
    SMI_HD void puts(const char *s) {
        while (*s) put(*s++);
    }
    SMI_HD void put_u64(unsigned long long v) {
        char t[20];
        int k = 0;
        do {
            t[k++] = (char)('0' + (int)(v % 10));
            v /= 10;
        } while (v);
        while (k) put(t[--k]);
    }
    SMI_HD void put_int(long long v) {
        if (v < 0) {
            put('-');
            put_u64((unsigned long long)(-v));
        } else
            put_u64((unsigned long long)v);
    }
};

// FastqRecordExt.REVERSE_COMPLEMENT (L72-104): a char[254] that is zero except for these letters
SMI_HD char rc_char(unsigned char c) {
    switch (c) {
    case 'A': case 'a': return 'T';
    case 'G': case 'g': return 'C';
    case 'C': case 'c': return 'G';
    case 'T': case 't': return 'A';
    case 'N': case 'n': return 'N';
    case 'H': case 'h': return 'D';
    case 'R': case 'r': return 'Y';
    case 'Y': case 'y': return 'R';
    case 'M': case 'm': return 'K';
    case 'K': case 'k': return 'M';
    case 'S': case 's': return 'S';
    case 'W': case 'w': return 'W';
    case 'B': case 'b': return 'V';
    case 'V': case 'v': return 'B';
    case 'D': case 'd': return 'H';
    default: return 0;
    }
}

// new DecimalFormat("##.#").format((double) f) (L36, L270): HALF_EVEN on the exact decimal value, at most one fraction
// digit, no integer digit in front of a fraction when it is zero
SMI_HD void put_dec1(NameSink &s, float f) {
    double t = (double)f * 10.0;  // exact: a 24-bit mantissa times 10 fits a double
    const bool neg = t < 0;
    if (neg) t = -t;
    long long q = (long long)t;  // floor, t >= 0
    const double frac = t - (double)q;
    if (frac > 0.5 || (frac == 0.5 && (q & 1))) q++;
    const long long ip = q / 10, tenth = q % 10;
    if (neg && q != 0) s.put('-');
    if (tenth == 0)
        s.put_u64((unsigned long long)ip);
    else {
        if (ip != 0) s.put_u64((unsigned long long)ip);
        s.put('.');
        s.put((char)('0' + (int)tenth));
    }
}

SMI_HD void put_base36(NameSink &s, uint32_t v) {  // FastqRecordExt$NumberToAndFromAscii.convertInt = Integer.toString(id, 36), L524
    char t[8];
    int k = 0;
    do {
        const uint32_t d = v % 36u;
        t[k++] = (char)(d < 10 ? '0' + d : 'a' + (d - 10));
        v /= 36u;
    } while (v);
    while (k) s.put(t[--k]);
}

SMI_HD void put_kmer16(NameSink &s, uint32_t key) {  // TWOBIT_TO_BASE_ARRAY: A G C T
    for (int i = 15; i >= 0; i--) {
        const uint32_t b = (key >> (2 * i)) & 3u;
        s.put(b == 0 ? 'A' : (b == 1 ? 'G' : (b == 2 ? 'C' : 'T')));
    }
}

enum { NAME_OK = 0, NAME_RANGE = 1 };  // NAME_RANGE: the X= / Q= range leaves the read (the reference throws from substring / skip)

// Appends what getRecordForWriting puts behind `readName.split(" ")[0]` (L220; the caller has written that token).
// seq_at(i) / qual_at(i): base / quality character i (0-based) of the RAW read of length len.
// Returns NAME_OK or NAME_RANGE; *stranded_ok = false in the "Beginrange inconsistent" case (L257-259: the name keeps no
// suffix and the record is written with the stranded sequence and a null quality string).
template <class SeqAt, class QualAt>
SMI_HD int append_name_suffix(NameSink &s, const smi_scan_result &scan, const smi_bc_result *bc, int rank, uint32_t read_id,
                              bool five_prime, int len, SeqAt seq_at, QualAt qual_at, bool *quals_set) {
    *quals_set = true;
    const bool fwd = scan.flags & SMI_F_PASSED_FWD, rev = scan.flags & SMI_F_PASSED_REV;
    if (!fwd && !rev) {
        s.puts("_FAILED ");  // L309
        return NAME_OK;
    }
    *quals_set = false;
    if (!scan.found) return NAME_OK;  // the suffix is only attached inside `if (adapterFound())` (L247-298)
    // 3': stranded[AE-40 .. AE+2] (L253-254); 5': stranded[AE-2 .. AE+39] (L250-251)
    const int begin = five_prime ? scan.adapter_end - 3 : scan.adapter_end - 40 - 1;
    const int end = five_prime ? scan.adapter_end + 39 : scan.adapter_end + 2;
    if (begin < 0) return NAME_OK;  // L257-259
    if (end > len || begin - 1 < 0) return NAME_RANGE;
    *quals_set = true;
    s.puts(rev ? "_REV_" : "_FWD_");
    if (scan.polya_end != 0) {
        s.puts("PS=");
        s.put_int(scan.polya_start);
        s.puts("_PE=");
        s.put_int(scan.polya_end);
        s.put('_');
    }
    s.puts("AE=");
    s.put_int(scan.adapter_end);
    s.put('_');
    if (scan.tso_end != 0) {
        s.puts("T=");
        s.put_int(scan.tso_end);
        s.put('_');
    }
    const bool has_bc = bc && bc->found == 1;
    if (has_bc) {
        // Parser.java:L274-279: 3' barcodes end at the adapter, 5' barcodes start behind it
        const int bc_start = five_prime ? scan.adapter_end + 1 + bc->offset : scan.adapter_end - 1 + bc->offset;
        const int bc_end = five_prime ? bc_start + 15 + bc->ins_minus_del : bc_start - 15 - bc->ins_minus_del;
        s.puts("bc=");
        put_kmer16(s, bc->bc);
        s.puts("_ed=");
        s.put_int(bc->ed);
        s.puts("_ed_sec=");
        s.put_int(bc->ed_sec);
        s.puts("_bcStart=");
        s.put_int(bc_start);
        s.puts("_bcEnd=");
        s.put_int(bc_end);
        s.put('_');
        if (rank > 0) {
            s.puts("rk=");
            s.put_int(rank);
            s.put('_');
        }
    }
    s.puts("X=");
    for (int i = begin; i < end; i++) s.put(rev ? rc_char((unsigned char)seq_at(len - 1 - i)) : (char)seq_at(i));
    s.puts("_Q=");
    long long sum = 0;
    int cnt = 0;
    for (int i = begin - 1; i <= end - 1 && i < len; i++, cnt++) sum += (int)(unsigned char)(rev ? qual_at(len - 1 - i) : qual_at(i)) - 33;
    put_dec1(s, (float)((double)sum / (double)cnt));
    s.put('_');
    put_base36(s, read_id);
    if (has_bc) {
        s.puts(" cellBC=");
        put_kmer16(s, bc->bc);
    }
    return NAME_OK;
}

}  // namespace smi
