"""Reference path patch_samplers/region_samplers.py -> the MI355X implementation."""
from deephisto_amd.patch_samplers.region_samplers import *  # noqa: F401,F403
from deephisto_amd.patch_samplers.region_samplers import (AnnoRegionDenseSampler, AnnoRegionRndSampler, RegionAnnotation,  # noqa: F401
                                                          extract_and_save_subset)
