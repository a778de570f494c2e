// TEST INFRASTRUCTURE ONLY: the reference reaches this path as ../../../../DiligentCore/... ; the recording stand-in lives in flat/dg_mock.hpp / dg_helpers.hpp
#pragma once
#include "dg_helpers.hpp"
