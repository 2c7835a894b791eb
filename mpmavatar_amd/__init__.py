"""MI355X-native hot path of the MPM garment simulator."""
import os

# One hardware queue per HIP stream.  The ROCm runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES = 4 hardware
# queues; the finite-difference training step (fd.MaterialFD) drives four solver contexts on four streams beside torch's own, two of
# them end up on one queue and run one after the other: 36.0 k substeps/s with the default, 45.3 k with 8 queues, nothing more with 16
# (profiles/r04_experiments.md 14).  The runtime reads the variable at its first HIP call, so it still takes effect here -- after
# `import torch`, before anything touched the device; a value set by the user wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
