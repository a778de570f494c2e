#define SSAO_ALGORITHM 1
#define A3_NS a3_hbao
#define A3_ENTRY ref_ssao_compute_ao_hbao
#include "ref_a3_body.inc"
