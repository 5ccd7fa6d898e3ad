// TEST INFRASTRUCTURE ONLY -- stand-in (everything is in ceres/ceres.h of this directory)
#pragma once
#include <ceres/ceres.h>
