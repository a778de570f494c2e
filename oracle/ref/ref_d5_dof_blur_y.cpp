// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D5 vertical (Macros: DOF_CIRCLE_OF_CONFUSION_BLUR_TYPE = DOF_CIRCLE_OF_CONFUSION_BLUR_Y, DepthOfField.cpp:564)
#define DOF_CIRCLE_OF_CONFUSION_BLUR_TYPE 1
#define D5NS d5y
#define D5FN ref_dof_blur_y
#include "ref_d5_body.inc"
