// TEST INFRASTRUCTURE ONLY -- empty stand-in (view_culling.cpp includes it without using it)
#pragma once
#include <opencv2/core.hpp>
