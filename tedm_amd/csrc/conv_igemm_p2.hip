// conv_igemm.inc instantiated for TEDM_PREC_* = 2 (one translation unit per arithmetic mode: they compile in parallel)
#define CONV_PREC 2
#include "conv_igemm.inc"
