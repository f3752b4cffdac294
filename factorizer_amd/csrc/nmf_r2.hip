#define FZ_R 2
#include "nmf_kernels.inc"
