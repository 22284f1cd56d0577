// One translation unit of the one-wave-per-instance kernels: hipcc -DALTRO_WIDE_TU=k compiles the instantiations that
// ALTRO_WIDE_KERNELS (solve_wide.h) assigns to unit k.  _lib.build() compiles the units side by side and links them with
// altro_batch.hip (which declares every instantiation `extern`).
#ifndef ALTRO_WIDE_TU
#error "compile with -DALTRO_WIDE_TU=<unit>"
#endif
#include "solve_wide.h"
