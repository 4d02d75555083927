#!/usr/bin/env python3
"""AUTO's cost table (us per round of every RK4 STM kernel family at 64 steps): the defaults of a new context against two
calibrations on the device (lto_calibrate_kernels)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lowthrustopt_amd as lto

c = lto.Context(0)
print("defaults  ", c.kernel_round_costs(12), c.kernel_round_costs(14), "whole-segment lanes", c.kernel_lane_round_us())
print("calibrated", c.calibrate_kernels(), "whole-segment lanes", c.kernel_lane_round_us())
print("again     ", c.calibrate_kernels(), "whole-segment lanes", c.kernel_lane_round_us())
c.close()
