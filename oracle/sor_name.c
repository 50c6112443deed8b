/*
 * sor_name.c -- ORACLE (test infrastructure; rules in sor_bc.c).
 *
 * Read-name suffix of passed reads = FastqRecordExt.getRecordForWriting
 * (FJ!nanoporereadscanner/readerwriter/FastqRecordExt.java:L209-311), prefixes from Jar/config.xml:41-52
 * (PS= PE= AE= T= X= Q=) and ReadScannerParameters.java:L139-159 (bc= ed= ed_sec= bcStart= bcEnd= rk=),
 * DEC_FORMATTER = new DecimalFormat("##.#") (L36), read id = Integer.toString(id, 36) (L524).
 */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "sor.h"

/* FastqRecordExt.REVERSE_COMPLEMENT (L72-104): zero except for these letters */
static const char COMP[256] = {['A'] = 'T', ['a'] = 'T', ['G'] = 'C', ['g'] = 'C', ['C'] = 'G', ['c'] = 'G', ['T'] = 'A', ['t'] = 'A',
                               ['N'] = 'N', ['n'] = 'N', ['H'] = 'D', ['h'] = 'D', ['R'] = 'Y', ['r'] = 'Y', ['Y'] = 'R', ['y'] = 'R',
                               ['M'] = 'K', ['m'] = 'K', ['K'] = 'M', ['k'] = 'M', ['S'] = 'S', ['s'] = 'S', ['W'] = 'W', ['w'] = 'W',
                               ['B'] = 'V', ['b'] = 'V', ['V'] = 'B', ['v'] = 'B', ['D'] = 'H', ['d'] = 'H'};

/* DecimalFormat("##.#").format((double)f): HALF_EVEN on the exact value, no integer digit when |v| < 1 and a fraction
 * digit is printed, "0" when everything rounds away */
static int fmt_dec1(float f, char *out) {
    double t = (double)f * 10.0; /* exact: 24-bit mantissa times 10 fits a double */
    int neg = t < 0;
    if (neg) t = -t;
    double r = floor(t);
    double fr = t - r;
    long long q = (long long)r;
    if (fr > 0.5 || (fr == 0.5 && (q & 1))) q++;
    long long ip = q / 10;
    int tenth = (int)(q % 10);
    char *p = out;
    if (neg && q != 0) *p++ = '-';
    if (tenth == 0)
        p += sprintf(p, "%lld", ip);
    else if (ip == 0)
        p += sprintf(p, ".%d", tenth);
    else
        p += sprintf(p, "%lld.%d", ip, tenth);
    return (int)(p - out);
}

static void to_base36(uint32_t v, char *out) {
    char tmp[16];
    int n = 0;
    if (v == 0) tmp[n++] = '0';
    while (v) {
        int d = (int)(v % 36);
        tmp[n++] = (char)(d < 10 ? '0' + d : 'a' + d - 10);
        v /= 36;
    }
    for (int i = 0; i < n; i++) out[i] = tmp[n - 1 - i];
    out[n] = 0;
}

/* returns the length written, or -1 where the reference would throw (substring / skip out of range), -2 on overflow.
 * scan: oracle scan record; bc: oracle assign record (found==1) or NULL; rank <= 0: no rk= field. */
int sor_format_read_name(const char *read_name, const char *raw_seq, const char *raw_qual, int len,
                         const sor_scan_result *scan, const sor_assign_t *bc, int rank, uint32_t read_id,
                         int five_prime, char *out, size_t cap) {
    char buf[1024];
    char *p = buf;
    /* readName = getReadName().split(" ")[0]  (L220) */
    size_t nl = strcspn(read_name, " ");
    if (nl > 400) return -2;
    memcpy(p, read_name, nl);
    p += nl;
    const int passed = (scan->flags & (SOR_F_PASSED_FWD | SOR_F_PASSED_REV)) != 0;
    if (!passed) {
        p += sprintf(p, "_FAILED "); /* L309 */
    } else {
        const int rev = (scan->flags & SOR_F_PASSED_REV) != 0;
        char add[768];
        char *a = add;
        a += sprintf(a, "%s_", rev ? "_REV" : "_FWD");
        if (scan->polya_end) { /* polyAFound(): polyA_Result.end != null */
            a += sprintf(a, "PS=%d_", scan->polya_start);
            a += sprintf(a, "PE=%d_", scan->polya_end);
        }
        if (scan->adapter_found) a += sprintf(a, "AE=%d_", scan->adapter_end);
        if (scan->tso_end) a += sprintf(a, "T=%d_", scan->tso_end); /* tSOFound(): tSOresult.end != null */
        if (bc && bc->found == 1) {
            char s[17];
            sor_twobit_decode(bc->bc, 16, s);
            a += sprintf(a, "bc=%s_ed=%d_ed_sec=%d_bcStart=%d_bcEnd=%d_", s, bc->ed, bc->ed_sec, bc->bc_start, bc->bc_end);
            if (rank > 0) a += sprintf(a, "rk=%d_", rank);
        }
        int append = 0;
        if (scan->adapter_found) {
            /* 3': stranded[AE-40 .. AE+2] (L253-254); 5': stranded[AE-2 .. AE+39] (L250-251), nbasesOfAdapterSeqInReadname = 3 */
            const int begin = five_prime ? scan->adapter_end - 3 : scan->adapter_end - 40 - 1;
            const int end = five_prime ? scan->adapter_end + 40 - 1 : scan->adapter_end + 2;
            if (begin >= 0) { /* else: passed = false and the name keeps no suffix (L257-259) */
                if (end > len) return -1;     /* String.substring */
                if (begin - 1 < 0) return -1; /* IntStream.skip(negative) in getMeanQV */
                a += sprintf(a, "X=");
                for (int i = begin; i < end; i++) /* stranded = reverse complement for PASSED_REV (L62-67) */
                    *a++ = rev ? COMP[(unsigned char)raw_seq[len - 1 - i]] : raw_seq[i];
                *a++ = '_';
                /* Q= : getMeanQV(quals, beginRange, endRange), 1-based inclusive = 0-based [begin-1, end-1] (L270) */
                long long sum = 0;
                int cnt = 0;
                for (int i = begin - 1; i <= end - 1 && i < len; i++) {
                    sum += (unsigned char)(rev ? raw_qual[len - 1 - i] : raw_qual[i]) - 33;
                    cnt++;
                }
                float q = (float)((double)sum / (double)cnt);
                a += sprintf(a, "Q=");
                a += fmt_dec1(q, a);
                *a++ = '_';
                char id[16];
                to_base36(read_id, id);
                a += sprintf(a, "%s", id);
                if (bc && bc->found == 1) {
                    char s[17];
                    sor_twobit_decode(bc->bc, 16, s);
                    a += sprintf(a, " cellBC=%s", s);
                }
                append = 1;
            }
        }
        if (append) {
            memcpy(p, add, (size_t)(a - add));
            p += a - add;
        }
    }
    size_t n = (size_t)(p - buf);
    if (n + 1 > cap) return -2;
    memcpy(out, buf, n);
    out[n] = 0;
    return (int)n;
}

int sor_fmt_dec1(float f, char *out) {
    int n = fmt_dec1(f, out);
    out[n] = 0;
    return n;
}

/* The whole record as the pass-2 writer emits it: getRecordForWriting (L209-311) + htsjdk BasicFastqWriter.write
 * ('@' name LF bases LF '+' quality header LF qualities LF; a null quality string prints as "null").
 * read_name: the full name line (fragments: after ChimeraFindernew's replaceFirst); qual_header: text behind '+';
 * force_failed: MULTI_CHIMERIC_READS_DISCARDED | FAILED records are never scanned (Parser.java:L92);
 * trim_fastq: -u (L210-217, L301-304).  Returns the record length, -1 where the reference throws, -2 on overflow;
 * *passed = 1 when the record goes to the `passed` file (FastqWriterThreadPool.java:L301). */
int sor_fastq_record(const char *read_name, const char *qual_header, const char *raw_seq, const char *raw_qual, int len,
                     const sor_scan_result *scan, const sor_assign_t *bc, int rank, uint32_t read_id, int five_prime,
                     int trim_fastq, int force_failed, char *out, size_t cap, int *passed) {
    sor_scan_result sc = *scan;
    if (force_failed) sc.flags &= ~(uint64_t)(SOR_F_PASSED_FWD | SOR_F_PASSED_REV);
    const int is_passed = (sc.flags & (SOR_F_PASSED_FWD | SOR_F_PASSED_REV)) != 0;
    const int rev = (sc.flags & SOR_F_PASSED_REV) != 0;
    *passed = is_passed;
    char name[1200];
    const int nl = sor_format_read_name(read_name, raw_seq, raw_qual, len, &sc, is_passed ? bc : NULL, rank, read_id, five_prime,
                                        name, sizeof name);
    if (nl < 0) return nl;
    /* partOfSeqToWrite (L209-217) */
    int cut_beg = 0, cut_end = len;
    if (is_passed && trim_fastq && bc && bc->found == 1) {
        const int begin = five_prime ? bc->bc_start + 30 : (sc.tso_end ? sc.tso_end : 1);
        const int end = sc.polya_end ? sc.polya_start : len;
        if (begin < end) {
            if (begin - 1 < 0 || end > len) return -1; /* String.substring */
            cut_beg = begin - 1;
            cut_end = end;
        }
    }
    /* quals stays null for a passed read whose name got no suffix (L221, L247-268) */
    int quals_set = 1;
    if (is_passed) {
        const int b = five_prime ? sc.adapter_end - 3 : sc.adapter_end - 41;
        quals_set = sc.adapter_found && b >= 0;
    }
    const size_t n_seq = (size_t)(cut_end - cut_beg), n_q = quals_set ? n_seq : 4;
    const size_t hl = strlen(qual_header);
    const size_t total = 1 + (size_t)nl + 1 + n_seq + 1 + 1 + hl + 1 + n_q + 1;
    if (total + 1 > cap) return -2;
    char *p = out;
    *p++ = '@';
    memcpy(p, name, (size_t)nl);
    p += nl;
    *p++ = '\n';
    for (int i = cut_beg; i < cut_end; i++) *p++ = (is_passed && rev) ? COMP[(unsigned char)raw_seq[len - 1 - i]] : raw_seq[i];
    *p++ = '\n';
    *p++ = '+';
    memcpy(p, qual_header, hl);
    p += hl;
    *p++ = '\n';
    if (!quals_set) {
        memcpy(p, "null", 4);
        p += 4;
    } else
        for (int i = cut_beg; i < cut_end; i++) *p++ = (is_passed && rev) ? raw_qual[len - 1 - i] : raw_qual[i];
    *p++ = '\n';
    *p = 0;
    return (int)(p - out);
}
