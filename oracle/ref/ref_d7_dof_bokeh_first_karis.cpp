// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D7 with FEATURE_FLAG_ENABLE_KARIS_INVERSE (Macros: DOF_OPTION_KARIS_INVERSE = 1, DepthOfField.cpp:635)
#define DOF_OPTION_KARIS_INVERSE 1
#define D7NS d7k
#define D7FN ref_dof_bokeh_first_karis
#include "ref_d7_body.inc"
