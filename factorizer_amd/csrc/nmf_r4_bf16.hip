#define FZ_R 4
#define FZ_AT fz::bf16
#define FZ_AT_TAG _bf16
#include "nmf_kernels.inc"
