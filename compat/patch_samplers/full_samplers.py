"""Reference path patch_samplers/full_samplers.py -> the MI355X implementation."""
from deephisto_amd.patch_samplers.full_samplers import *  # noqa: F401,F403
from deephisto_amd.patch_samplers.full_samplers import FullImageDenseSampler, FullImageRndSampler, SamplerExecutionMode  # noqa: F401
