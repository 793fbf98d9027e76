// conv_igemm.inc instantiated for TEDM_PREC_* = 0 (one translation unit per arithmetic mode: they compile in parallel)
#define CONV_PREC 0
#include "conv_igemm.inc"
