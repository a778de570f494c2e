// api_stubs.cpp -- entry points declared in include/mifx.h whose implementation has not landed yet.
// Each returns MIFX_ERR_NOT_IMPLEMENTED loudly (never a silent fallback); the file shrinks as passes land.
#include "mifx_objects.h"

using namespace mifx;

#define MIFX_STUB(name)                                  \
    do                                                   \
    {                                                    \
        set_error(name ": not implemented in this build"); \
        return MIFX_ERR_NOT_IMPLEMENTED;                 \
    } while (0)

mifx_chain::~mifx_chain() {}

extern "C" {
mifx_status mifx_chain_create(const mifx_device_desc*, const mifx_postfx_create_info*, mifx_chain**) { MIFX_STUB("mifx_chain_create"); }
void        mifx_chain_destroy(mifx_chain*) {}
mifx_status mifx_chain_execute(mifx_chain*, const mifx_chain_frame*, const mifx_image2d*) { MIFX_STUB("mifx_chain_execute"); }
mifx_status mifx_chain_get_postfx(mifx_chain*, mifx_postfx**) { MIFX_STUB("mifx_chain_get_postfx"); }
mifx_status mifx_chain_reset_history(mifx_chain*) { MIFX_STUB("mifx_chain_reset_history"); }
}
