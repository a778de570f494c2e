// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D7 without FEATURE_FLAG_ENABLE_KARIS_INVERSE (Macros: DOF_OPTION_KARIS_INVERSE = 0, DepthOfField.cpp:635)
#define DOF_OPTION_KARIS_INVERSE 0
#define D7NS d7
#define D7FN ref_dof_bokeh_first
#include "ref_d7_body.inc"
