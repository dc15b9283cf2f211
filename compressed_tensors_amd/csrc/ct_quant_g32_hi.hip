// bits 5-8 of the any-bit-width pack-group kernels
#include "ct_quant_g32.inc"
