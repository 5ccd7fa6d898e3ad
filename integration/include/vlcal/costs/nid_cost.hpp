// integration/include/ goes FIRST on the include path of a reference checkout (together with
// -DNIDREG_WITH_REFERENCE_DEPS): the reference's own `#include <vlcal/costs/nid_cost.hpp>` then resolves to the
// MI355X drop-in ("replace by include/vlcal_amd/nid_cost.hpp").
#pragma once
#include <vlcal_amd/nid_cost.hpp>
