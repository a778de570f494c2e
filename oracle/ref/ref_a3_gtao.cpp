#define SSAO_ALGORITHM 0
#define A3_NS a3_gtao
#define A3_ENTRY ref_ssao_compute_ao_gtao
#include "ref_a3_body.inc"
