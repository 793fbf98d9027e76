// conv_igemm.inc instantiated for TEDM_PREC_* = 3 (one translation unit per arithmetic mode: they compile in parallel)
#define CONV_PREC 3
#include "conv_igemm.inc"
