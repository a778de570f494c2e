// TEST INFRASTRUCTURE ONLY: a DiligentCore header name the reference includes; the recording stand-in lives in dg_mock.hpp / dg_helpers.hpp
#pragma once
#include "dg_helpers.hpp"
