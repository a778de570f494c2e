// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A7 with reversed depth.
#define SSAO_OPTION_INVERTED_DEPTH 1
#define a7 a7_rev
#define ref_ssao_resampled_history ref_ssao_resampled_history_rev
#include "ref_a7_resampled_history.cpp"
