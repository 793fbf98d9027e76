// conv_igemm.inc instantiated for TEDM_PREC_* = 4 (one translation unit per arithmetic mode: they compile in parallel)
#define CONV_PREC 4
#include "conv_igemm.inc"
