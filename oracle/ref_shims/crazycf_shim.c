/*
 * crazycf_shim.c -- builds the REFERENCE tests/crazyCF/crazyCF.c (the CFCSS test program) from the source where it lies
 * under /root/reference, as a callable: its main() renamed, its printf captured, its srand(42) fed the caller's seed, its
 * global `size` set before the call.  `timesThroughWhile = 10` is a literal inside main() and stays what it is.
 * Test infrastructure only.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

static unsigned ref_ccf_seed = 42;
static int ref_ccf_printed, ref_ccf_nprints, ref_ccf_total;

static int ref_ccf_printf(const char *fmt, ...)
{
    va_list ap;
    int v;
    va_start(ap, fmt);
    v = va_arg(ap, int);
    va_end(ap);
    if (fmt[0] == 't') { /* "total so far: %d\n" (crazyCF.c:54) */
        ref_ccf_printed = v;
        ref_ccf_nprints += 1;
    } else /* "Total = %d\n" (:69) */
        ref_ccf_total = v;
    return 0;
}

#define size ref_ccf_size
#define golden ref_ccf_golden
#define generateGolden ref_ccf_generateGolden
#define fillArray ref_ccf_fillArray
#define main ref_ccf_main
#define printf ref_ccf_printf
#define srand(x) srand(ref_ccf_seed)

#include "crazyCF/crazyCF.c"

#undef size
#undef main
#undef printf
#undef srand

/* one run with (seed, size); out = {Total, the "total so far" value, how many such lines} */
void ref_crazycf(unsigned seed, int size, int out[3])
{
    ref_ccf_seed = seed;
    ref_ccf_size = size;
    ref_ccf_printed = 0;
    ref_ccf_nprints = 0;
    ref_ccf_total = 0;
    (void)ref_ccf_main();
    out[0] = ref_ccf_total;
    out[1] = ref_ccf_printed;
    out[2] = ref_ccf_nprints;
}
