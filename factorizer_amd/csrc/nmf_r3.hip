#define FZ_R 3
#include "nmf_kernels.inc"
