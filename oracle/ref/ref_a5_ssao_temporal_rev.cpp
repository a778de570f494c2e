// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A5 with reversed depth.
#define SSAO_OPTION_INVERTED_DEPTH 1
#define a5 a5_rev
#define ref_ssao_temporal_accumulation ref_ssao_temporal_accumulation_rev
#include "ref_a5_ssao_temporal.cpp"
