// see integration/include/vlcal/costs/nid_cost.hpp: vlcal::CostCalculator lives in the drop-in header too
#pragma once
#include <vlcal_amd/cost_calculator_nid.hpp>
