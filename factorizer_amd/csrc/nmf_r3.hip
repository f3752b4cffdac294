#define FZ_R 3
#define FZ_AT float
#include "nmf_kernels.inc"
