// conv_igemm.inc instantiated for TEDM_PREC_* = 1 (one translation unit per arithmetic mode: they compile in parallel)
#define CONV_PREC 1
#include "conv_igemm.inc"
