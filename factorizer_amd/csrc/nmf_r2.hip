#define FZ_R 2
#define FZ_AT float
#include "nmf_kernels.inc"
