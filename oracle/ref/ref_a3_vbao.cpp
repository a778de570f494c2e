#define SSAO_ALGORITHM 2
#define A3_NS a3_vbao
#define A3_ENTRY ref_ssao_compute_ao_vbao
#include "ref_a3_body.inc"
