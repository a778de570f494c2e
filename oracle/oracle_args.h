/* oracle_args.h -- TEST INFRASTRUCTURE ONLY.  The C argument block shared by the two checker libraries:
 * oracle/_ref/libmifx_ref.so (reference shader source compiled for the CPU) and oracle/libmifx_oracle.so
 * (hand-written restatement).  Host images are row-major float arrays, `c` floats per texel. */
#ifndef MIFX_ORACLE_ARGS_H
#define MIFX_ORACLE_ARGS_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct
{
    float* data;
    int    w, h, c;
} ref_img;

#define REF_MAX_IN 14
#define REF_MAX_MIPS 12
typedef struct
{
    ref_img     in[REF_MAX_IN][REF_MAX_MIPS];
    int         in_mips[REF_MAX_IN];
    ref_img     out[4];
    const void* cam0;    /* CameraAttribs, current frame (576 bytes)  */
    const void* cam1;    /* CameraAttribs, previous frame             */
    const void* attribs; /* the pass' attribs struct, byte-identical to the reference struct */
    int         ival[8];
    float       fval[8];
} ref_args;
#ifdef __cplusplus
}
#endif
#endif
