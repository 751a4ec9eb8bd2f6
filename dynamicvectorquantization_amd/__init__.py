"""MI355X-native DQ-VAE / DQ-Transformer hot path."""
import os as _os

# HIP maps streams onto at most GPU_MAX_HW_QUEUES hardware queues (ROCm 7.2 default: 4, of which the sampler's lanes saw TWO: kernels
# of streams that share a queue serialise).  Eight queues let four sampling lanes run four token steps at once -- 10.1 k -> 14.2 k
# token-steps/s at batch 8, 26.3 k -> 33.6 k at batch 50 (profiles/r06_sampler_lanes.txt) -- and are neutral for the two-stream training
# steps (425.8 vs 425.4 img/s, 468.0 vs 467.9).  Read by the HIP runtime when it initialises: set before the first GPU call; an
# explicit setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
