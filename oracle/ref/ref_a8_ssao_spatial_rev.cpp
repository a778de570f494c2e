// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A8 with reversed depth.
#define SSAO_OPTION_INVERTED_DEPTH 1
#define a8 a8_rev
#define ref_ssao_spatial_reconstruction ref_ssao_spatial_reconstruction_rev
#include "ref_a8_ssao_spatial.cpp"
