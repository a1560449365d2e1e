/*
 * qs_shim.c -- builds the REFERENCE quick_sort (tests/quicksort/quicksort.c:109-129) from the source where it lies under
 * /root/reference.  The file is a whole benchmark (globals, an endless driver, main): everything is renamed into a ref_qs_
 * namespace and only the sort is exported.  Test infrastructure only.
 */
#define ind ref_qs_ind
#define local_errors ref_qs_local_errors
#define in_block ref_qs_in_block
#define seed_value ref_qs_seed_value
#define array ref_qs_array
#define golden_array ref_qs_golden_array
#define golden_array_rev ref_qs_golden_array_rev
#define init_array ref_qs_init_array
#define quick_sort ref_qs_quick_sort
#define quick_sort_rev ref_qs_quick_sort_rev
#define checker ref_qs_checker
#define qsort_test ref_qs_qsort_test
#define main ref_qs_main

#include "quicksort/quicksort.c"

void ref_quicksort(int *a, int n) { ref_qs_quick_sort(a, n); }
