/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See vlg_oracle_impl.h for the header comment.
 * Builds libvlg_oracle.so: fp32 (`_f32`) and fp64 (`_f64`) CPU restatements of the hot path.
 * OpenMP parallelises over sentences only (the DP itself is the scalar loop of the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL float
#define SUF _f32
#define EXP expf
#define LOG logf
#define SQRT sqrtf
#include "vlg_oracle_impl.h"
#undef REAL
#undef SUF
#undef EXP
#undef LOG
#undef SQRT

#define REAL double
#define SUF _f64
#define EXP exp
#define LOG log
#define SQRT sqrt
#include "vlg_oracle_impl.h"
#undef REAL
#undef SUF
#undef EXP
#undef LOG
#undef SQRT

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
