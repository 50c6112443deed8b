/*
 * sor.h -- ORACLE interface (test infrastructure; see the header of sor_bc.c).
 * "sor" = sicelore oracle.  Plain C so that tests can bind it with ctypes.
 */
#ifndef SOR_H
#define SOR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BarcodeMatchTester$Matches$OneMatch */
typedef struct {
    int64_t read_seq;    /* un-mutated window (2-bit long, may be N-poisoned) */
    int64_t matching_bc; /* barcode that was hit */
    int32_t ed, subs, ins, dels, offset, length;
} sor_match_t;

/* what Parser.assignBarcode leaves in ReadScanResult$BarcodeResult */
typedef struct {
    int64_t bc;
    int32_t found, ed, ed_sec, offset, ins_minus_del, bc_start, bc_end, n_matches;
    uint64_t n_probes;
} sor_assign_t;

typedef struct sor_set sor_set;

int64_t sor_twobit_encode(const char *s, int n);
void sor_twobit_decode(int64_t seq, int len, char *out);
int64_t sor_twobit_revcomp(int64_t seq, int len);
void sor_replace_deg(int64_t seq, int pos, int len, int64_t out[4]);
void sor_insert_deg(int64_t seq, int pos, int len, int64_t out[4]);
int64_t sor_delete_byte(int64_t seq, int base_add_at_end4, int pos, int len);
int sor_fourbit_encode_char(unsigned char c);
int sor_fourbit_complement(int b);

sor_set *sor_set_new(const int64_t *keys, size_t n);
void sor_set_free(sor_set *s);
size_t sor_set_size(const sor_set *s);
int sor_set_contains(const sor_set *s, int64_t key);

int sor_bc_match(const sor_set *search, int64_t seq, int len, int ed, int skip_full_matches, int allow_indels,
                 const uint8_t *post4, int post_len, int offset, int do_next_level_if_match_found, sor_match_t *out,
                 int max_out, uint64_t *n_probes);

int sor_assign_barcode(const sor_set *search, const char *stranded, int read_len, int adapterpos, int max_ed,
                       int test_plus_minus, int five_prime, int bc_len, sor_assign_t *res);

#ifdef __cplusplus
}
#endif
#endif
