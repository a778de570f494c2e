// api_stubs.cpp -- every entry point declared in include/mifx.h is implemented; this file is intentionally empty and kept only so
// that a future not-yet-implemented entry point has an obvious home (it must return MIFX_ERR_NOT_IMPLEMENTED loudly, never fall back).
#include "mifx_objects.h"
