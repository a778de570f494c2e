#define TAA_OPTION_GAUSSIAN_WEIGHTING 1
#define TAA_OPTION_BICUBIC_FILTER 1
#define TAA_OPTION_YCOCG_COLOR_SPACE 0
#define T1_NS t1_f3
#define T1_ENTRY ref_taa_flags3
#include "ref_t1_body.inc"
