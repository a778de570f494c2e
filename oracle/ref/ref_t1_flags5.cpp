#define TAA_OPTION_GAUSSIAN_WEIGHTING 1
#define TAA_OPTION_BICUBIC_FILTER 0
#define TAA_OPTION_YCOCG_COLOR_SPACE 1
#define T1_NS t1_f5
#define T1_ENTRY ref_taa_flags5
#include "ref_t1_body.inc"
