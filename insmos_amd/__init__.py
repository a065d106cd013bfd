"""insmos_amd -- the InsMOS sparse-voxel inference hot path, native on MI355X."""
import os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  InsMOS_Model keeps a few launch sets in flight on
# their own worker streams next to the caller's stream; streams that share a hardware queue serialise (measured in round 1 with
# four windows in flight: 447 -> 485 windows/s with 8 queues).  Read by the runtime when it initialises, i.e. effective if this
# package is imported before the first GPU call of the process; a value already set by the user wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
