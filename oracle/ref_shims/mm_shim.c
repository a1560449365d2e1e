/*
 * mm_shim.c -- builds the REFERENCE matrix_multiply/checkGolden (tests/mm_common/mm_common_tmr.c) for one
 * compile-time `side`, from the source where it lies under /root/reference (never copied into this repo).
 * Compiled once per size with -DSIDE=<n>; exports ref_mm_run_<n>().  Test infrastructure only.
 */
#include <stdint.h>
#include <string.h>

#ifndef SIDE
#error "compile with -DSIDE=<n>"
#endif
#define PASTE2(a, b) a##b
#define PASTE(a, b) PASTE2(a, b)
#define NAME(x) PASTE(PASTE(ref_mm, SIDE), x)

typedef uint32_t mm_t;
#include "COAST.h" /* the reference's tests/COAST.h (-I) */

#define side SIDE
#define first_matrix NAME(_first)
#define second_matrix NAME(_second)
#define results_matrix NAME(_results)
#define xor_golden NAME(_golden)
#define matrix_multiply NAME(_matrix_multiply)
#define checkGolden NAME(_checkGolden)
#define mm_run_test NAME(_mm_run_test)

mm_t first_matrix[side][side];
mm_t second_matrix[side][side];
uint32_t xor_golden;

#include "mm_common/mm_common_tmr.c"

/* returns checkGolden(): 0 = XOR of results equals `golden` */
int NAME(_run)(const uint32_t *f, const uint32_t *s, uint32_t *r, uint32_t golden)
{
    memcpy(first_matrix, f, sizeof(first_matrix));
    memcpy(second_matrix, s, sizeof(second_matrix));
    xor_golden = golden;
    mm_run_test();
    memcpy(r, results_matrix, sizeof(results_matrix));
    return checkGolden();
}
