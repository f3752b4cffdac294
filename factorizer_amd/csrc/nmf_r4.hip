#define FZ_R 4
#include "nmf_kernels.inc"
