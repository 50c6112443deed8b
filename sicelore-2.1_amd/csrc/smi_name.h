// smi_name.h -- the name a scanned read is written with, shared by the host entry point (smi_format_read_name) and the
// device record writer (K-WRITE): FastqRecordExt.getRecordForWriting
// (FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L209-311); prefixes from Jar/config.xml:41-52 (PS= PE= AE= T=
// X= Q=) and ReadScannerParameters.java:L139-159 (bc= ed= ed_sec= bcStart= bcEnd= rk=).  Plain character appends, no
// library calls, so the same code runs in a kernel.
#pragma once
#include <cstdint>
#include <type_traits>

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
    // the same digits for a value that fits 32 bits (coordinates, counts): no 64-bit division, which a GPU lane emulates
    SMI_HD void put_u32(uint32_t v) {
#if !defined(__HIP_DEVICE_COMPILE__)
        {   // host: as many steps as the number has digits (the constant-divisor form below always takes ten)
            char t[10];
            int k = 0;
            do {
                t[k++] = (char)('0' + (int)(v % 10u));
                v /= 10u;
            } while (v);
            while (k) put(t[--k]);
            return;
        }
#endif
        // digits from the top by constant divisors: no digit buffer (an indexed local array is scratch memory on the device)
        bool started = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t div = 1000000000u; div >= 1u; div /= 10u) {
            const uint32_t d = v / div;
            v -= d * div;
            if (d || started || div == 1u) {
                put((char)('0' + (int)d));
                started = true;
            }
        }
    }
    SMI_HD void put_i32(int v) {
        if (v < 0) {
            put('-');
            put_u32(0u - (uint32_t)v);
        } else
            put_u32((uint32_t)v);
    }
};

// FastqRecordExt.REVERSE_COMPLEMENT (L72-104): a char[254] that is zero except for these letters
SMI_HD char rc_char_switch(unsigned char c) {
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

#if !defined(__HIP_DEVICE_COMPILE__)
struct RcHostTable {  // host: one table look-up per character
    char t[256];
    RcHostTable() {
        for (int i = 0; i < 256; i++) t[i] = rc_char_switch((unsigned char)i);
    }
};
static const RcHostTable g_rc_host_table;
#endif
SMI_HD char rc_char(unsigned char c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rc_char_switch(c);
#else
    return g_rc_host_table.t[c];
#endif
}

// Optional bulk hooks a host sink / window may offer (the device formats character by character out of registers): a sink that
// declares `kFastKmer` has put_kmer16(key); a window functor that declares `kBulk` has bulk_x(sink, window) / bulk_sum(window)
template <class T, class = void>
struct has_fast_kmer : std::false_type {};
template <class T>
struct has_fast_kmer<T, std::void_t<decltype(T::kFastKmer)>> : std::true_type {};
template <class T, class = void>
struct has_bulk : std::false_type {};
template <class T>
struct has_bulk<T, std::void_t<decltype(T::kBulk)>> : std::true_type {};

// new DecimalFormat("##.#").format((double) f) (L36, L270): HALF_EVEN on the exact decimal value, at most one fraction
// digit, no integer digit in front of a fraction when it is zero
template <class Sink>
SMI_HD void put_dec1(Sink &s, float f) {
    double t = (double)f * 10.0;  // exact: a 24-bit mantissa times 10 fits a double
    const bool neg = t < 0;
    if (neg) t = -t;
    long long q = (long long)t;  // floor, t >= 0
    const double frac = t - (double)q;
    if (frac > 0.5 || (frac == 0.5 && (q & 1))) q++;
    const long long ip = q / 10, tenth = q % 10;
    if (neg && q != 0) s.put('-');
    auto put_ip = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        s.put_u32((uint32_t)ip);  // the device formats mean qualities only (< 256)
#else
        if (ip <= 0xFFFFFFFFll)
            s.put_u32((uint32_t)ip);
        else
            s.put_u64((unsigned long long)ip);
#endif
    };
    if (tenth == 0)
        put_ip();
    else {
        if (ip != 0) put_ip();
        s.put('.');
        s.put((char)('0' + (int)tenth));
    }
}

template <class Sink>
SMI_HD void put_base36(Sink &s, uint32_t v) {  // FastqRecordExt$NumberToAndFromAscii.convertInt = Integer.toString(id, 36), L524
#if !defined(__HIP_DEVICE_COMPILE__)
    {
        char t[8];
        int k = 0;
        do {
            const uint32_t d = v % 36u;
            t[k++] = (char)(d < 10 ? '0' + d : 'a' + (d - 10));
            v /= 36u;
        } while (v);
        while (k) s.put(t[--k]);
        return;
    }
#endif
    bool started = false;  // digits from the top, as in put_u32: 36^6 > 2^31, seven digits at most
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t div = 2176782336u; div >= 1u; div /= 36u) {
        const uint32_t d = v / div;
        v -= d * div;
        if (d || started || div == 1u) {
            s.put((char)(d < 10 ? '0' + d : 'a' + (d - 10)));
            started = true;
        }
    }
}

template <class Sink>
SMI_HD void put_kmer16(Sink &s, uint32_t key) {  // TWOBIT_TO_BASE_ARRAY: A G C T
    if constexpr (has_fast_kmer<Sink>::value) {
        s.put_kmer16(key);
        return;
    }
    for (int i = 15; i >= 0; i--) {
        const uint32_t b = (key >> (2 * i)) & 3u;
        s.put(b == 0 ? 'A' : (b == 1 ? 'G' : (b == 2 ? 'C' : 'T')));
    }
}

enum { NAME_OK = 0, NAME_RANGE = 1 };  // NAME_RANGE: the X= / Q= range leaves the read (the reference throws from substring / skip)

// The X= and Q= fields read one window of the STRANDED read: Q= averages stranded positions begin-1 .. end-1 (0-based), X= prints
// begin .. end-1, i.e. the window without its first character.  3': stranded[AE-40 .. AE+2] (L253-254), 5': stranded[AE-2 .. AE+39]
// (L250-251), so the window has 44 resp. 43 characters.  In the RAW read it is the contiguous range raw[lo .. lo + n_chars), read
// backwards (and complemented, for the bases) when the read passed on the reverse strand.
struct NameWindow {
    bool has;     // the name carries X= / Q= (passed, adapter found, window inside the read)
    int status;   // NAME_RANGE: the reference throws
    bool rev;
    int lo, n_chars;
};
constexpr int kNameWindowMax = 44;
SMI_HD NameWindow name_window(const smi_scan_result &scan, bool five_prime, int len) {
    NameWindow w{false, NAME_OK, false, 0, 0};
    const bool fwd = scan.flags & SMI_F_PASSED_FWD, rev = scan.flags & SMI_F_PASSED_REV;
    if ((!fwd && !rev) || !scan.found) return w;
    const int begin = five_prime ? scan.adapter_end - 3 : scan.adapter_end - 40 - 1;
    const int end = five_prime ? scan.adapter_end + 39 : scan.adapter_end + 2;
    if (begin < 0) return w;  // L257-259
    if (end > len || begin - 1 < 0) {
        w.status = NAME_RANGE;
        return w;
    }
    w.has = true;
    w.rev = rev;
    w.n_chars = end - begin + 1;
    w.lo = rev ? len - end : begin - 1;
    return w;
}

// Appends what getRecordForWriting puts behind `readName.split(" ")[0]` (L220; the caller has written that token).
// seq_w(k) / qual_w(k): character k (0 <= k < n_chars) of the window of name_window() in STRANDED order, the base not yet complemented
// (host: raw[rev ? lo + n_chars - 1 - k : lo + k]; the device writer loads the window with a few wide loads before it formats).
// Returns NAME_OK or NAME_RANGE; *stranded_ok = false in the "Beginrange inconsistent" case (L257-259: the name keeps no
// suffix and the record is written with the stranded sequence and a null quality string).
template <class Sink, class SeqAt, class QualAt>
SMI_HD int append_name_suffix(Sink &s, const smi_scan_result &scan, const smi_bc_result *bc, int rank, uint32_t read_id,
                              bool five_prime, int len, SeqAt seq_w, QualAt qual_w, bool *quals_set) {
    *quals_set = true;
    const bool fwd = scan.flags & SMI_F_PASSED_FWD, rev = scan.flags & SMI_F_PASSED_REV;
    if (!fwd && !rev) {
        s.puts("_FAILED ");  // L309
        return NAME_OK;
    }
    *quals_set = false;
    if (!scan.found) return NAME_OK;  // the suffix is only attached inside `if (adapterFound())` (L247-298)
    const NameWindow nw = name_window(scan, five_prime, len);
    if (nw.status == NAME_RANGE) return NAME_RANGE;
    if (!nw.has) return NAME_OK;  // L257-259
    *quals_set = true;
    s.puts(rev ? "_REV_" : "_FWD_");
    if (scan.polya_end != 0) {
        s.puts("PS=");
        s.put_i32(scan.polya_start);
        s.puts("_PE=");
        s.put_i32(scan.polya_end);
        s.put('_');
    }
    s.puts("AE=");
    s.put_i32(scan.adapter_end);
    s.put('_');
    if (scan.tso_end != 0) {
        s.puts("T=");
        s.put_i32(scan.tso_end);
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
        s.put_i32(bc->ed);
        s.puts("_ed_sec=");
        s.put_i32(bc->ed_sec);
        s.puts("_bcStart=");
        s.put_i32(bc_start);
        s.puts("_bcEnd=");
        s.put_i32(bc_end);
        s.put('_');
        if (rank > 0) {
            s.puts("rk=");
            s.put_i32(rank);
            s.put('_');
        }
    }
    s.puts("X=");
    // constant trip counts with a guard: unrolled on the device, where the window sits in registers
    if constexpr (has_bulk<SeqAt>::value)
        seq_w.bulk_x(s, nw);
    else {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 1; k < kNameWindowMax; k++)
            if (k < nw.n_chars) s.put(rev ? rc_char((unsigned char)seq_w(k)) : (char)seq_w(k));
    }
    s.puts("_Q=");
    int sum = 0;
    if constexpr (has_bulk<QualAt>::value)
        sum = qual_w.bulk_sum(nw);
    else {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < kNameWindowMax; k++)
            if (k < nw.n_chars) sum += (int)(unsigned char)qual_w(k) - 33;
    }
    put_dec1(s, (float)((double)sum / (double)nw.n_chars));
    s.put('_');
    put_base36(s, read_id);
    if (has_bc) {
        s.puts(" cellBC=");
        put_kmer16(s, bc->bc);
    }
    return NAME_OK;
}

}  // namespace smi
