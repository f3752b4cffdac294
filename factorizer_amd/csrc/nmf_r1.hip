#define FZ_R 1
#include "nmf_kernels.inc"
