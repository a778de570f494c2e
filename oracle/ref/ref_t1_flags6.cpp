#define TAA_OPTION_GAUSSIAN_WEIGHTING 0
#define TAA_OPTION_BICUBIC_FILTER 1
#define TAA_OPTION_YCOCG_COLOR_SPACE 1
#define T1_NS t1_f6
#define T1_ENTRY ref_taa_flags6
#include "ref_t1_body.inc"
