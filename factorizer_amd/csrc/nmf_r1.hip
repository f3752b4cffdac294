#define FZ_R 1
#define FZ_AT float
#include "nmf_kernels.inc"
