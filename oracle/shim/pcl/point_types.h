// TEST INFRASTRUCTURE ONLY -- stand-in (everything is in pcl/point_cloud.h of this directory)
#pragma once
#include <pcl/point_cloud.h>
