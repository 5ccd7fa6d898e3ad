// see integration/include/vlcal/costs/nid_cost.hpp: forwards the reference's include to the drop-in
#pragma once
#include <vlcal_amd/cost_calculator_nid.hpp>
