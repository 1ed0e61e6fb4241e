/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see trx_oracle_body.h for the full header).
 * Build: make -C oracle        (gcc -O2 -fopenmp -shared -fPIC -> oracle/liboracle.so)
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

#define REAL float
#define SUF _f32
#define FLOOR floorf
#define SIN sinf
#define COS cosf
#define TANH tanhf
#define FMA fmaf
#include "trx_oracle_body.h"
#undef REAL
#undef SUF
#undef FLOOR
#undef SIN
#undef COS
#undef TANH
#undef FMA

#define REAL double
#define SUF _f64
#define FLOOR floor
#define SIN sin
#define COS cos
#define TANH tanh
#define FMA fma
#include "trx_oracle_body.h"

int orc_version(void) { return 1; }
