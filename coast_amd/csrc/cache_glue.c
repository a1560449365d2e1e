/*
 * cache_glue.c -- the strong calc_sum for ONE benchmark's array size.  In the reference `data_array_elements` is a macro
 * (tests/cache_test/cacheTest.c:78) and calc_sum(int *array) does not carry it, so the drop-in needs this one-line TU
 * compiled with the same value:  gcc -Ddata_array_elements=600 -c cache_glue.c
 */
#ifndef data_array_elements
#error "compile with -Ddata_array_elements=<the benchmark's array size>"
#endif

int coast_dropin_calc_sum(int *array, int n);

int calc_sum(int *array)
{
    return coast_dropin_calc_sum(array, data_array_elements);
}
