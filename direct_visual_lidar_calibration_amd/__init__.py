"""MI355X-native NID direct LiDAR-camera registration core (drop-in for the vlcal calibrate hot path).

Host-side mirror of the reference interface lives in ``nid.py`` (``NIDCost``, ``CostCalculatorNID``,
``MultiNIDCost``, ``create_camera``); the compute is hand-written HIP in ``csrc/`` behind the C ABI
declared in ``include/nidreg.h``.  Importing this package does not load the HIP library; the first
use of a cost object does, and fails loudly if it was not built (``__graft_entry__.build()``).
"""
__version__ = "0.1.0"
