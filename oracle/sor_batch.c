/*
 * sor_batch.c -- ORACLE (test infrastructure): run sor_assign_barcode over a batch of reads, optionally on
 * several threads (OpenMP), so that parity tests at 10^5..10^6 reads finish in seconds and bench.py has a
 * CPU baseline ("port": restatement, not the Java reference -- no JVM exists in the image).
 */
#include <string.h>

#include "sor.h"

/* codes: n rows of `width` 2-bit codes (0..3 = A,G,C,T, anything else = N), stranded orientation */
int sor_assign_batch_codes(const sor_set *set, const uint8_t *codes, int width, const int32_t *ae, size_t n,
                           int max_ed, int test_plus_minus, int five_prime, sor_assign_t *out, int32_t *status,
                           int n_threads) {
    static const char LUT[5] = {'A', 'G', 'C', 'T', 'N'};
    if (width > 4096) return -1;
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
    for (long long i = 0; i < (long long)n; i++) {
        char buf[4097];
        const uint8_t *row = codes + (size_t)i * width;
        for (int j = 0; j < width; j++) buf[j] = LUT[row[j] > 3 ? 4 : row[j]];
        buf[width] = 0;
        status[i] = sor_assign_barcode(set, buf, width, ae[i], max_ed, test_plus_minus, five_prime, 16, &out[i]);
    }
    return 0;
}
