// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D5 horizontal (Macros: DOF_CIRCLE_OF_CONFUSION_BLUR_TYPE = DOF_CIRCLE_OF_CONFUSION_BLUR_X, DepthOfField.cpp:529)
#define DOF_CIRCLE_OF_CONFUSION_BLUR_TYPE 0
#define D5NS d5x
#define D5FN ref_dof_blur_x
#include "ref_d5_body.inc"
