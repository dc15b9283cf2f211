// bits 1-4 of the any-bit-width pack-group kernels
#define CT_G32_LO 1
#include "ct_quant_g32.inc"
