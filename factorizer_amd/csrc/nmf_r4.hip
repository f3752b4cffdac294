#define FZ_R 4
#define FZ_AT float
#include "nmf_kernels.inc"
