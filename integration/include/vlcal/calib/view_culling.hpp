// see integration/include/vlcal/costs/nid_cost.hpp: forwards the reference's include to the drop-in
#pragma once
#include <vlcal_amd/view_culling.hpp>
