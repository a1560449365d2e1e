/*
 * mm_glue.c -- the per-benchmark glue TU for matrix_multiply: its size is not part of its ABI (`side` is a macro and
 * the parameters decay to mm_t (*)[side], tests/mm_common/mm_common_tmr.c:3), so the strong replacement symbol is
 * compiled once per benchmark with the benchmark's own side:   cc -Dside=<n> -c mm_glue.c
 * and linked in front of the weakened matrix_multiply of the driver object.
 */
#include <stdint.h>
#ifndef side
#error "compile with -Dside=<n> (take it from the benchmark: cc -dM -E driver.c | grep 'define side')"
#endif
void coast_dropin_matrix_multiply(const void *f, const void *s, void *r, int n);

void matrix_multiply(uint32_t f_matrix[][side], uint32_t s_matrix[][side], uint32_t r_matrix[][side])
{
    coast_dropin_matrix_multiply(f_matrix, s_matrix, r_matrix, side);
}
